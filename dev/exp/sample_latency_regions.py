"""model(..., mode='sample') one batch of 64 at a time (the reference's eval loop) by regions per image: python dev/exp/sample_latency_regions.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import captioning.models as models
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL as cfg
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch = torch.bfloat16, 64
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
for R in (36, 64, 100):
    att = torch.from_numpy(W.synthetic_att_feats(64, R, cfg.att_feat_size, seed=3)).cuda()
    fc = torch.zeros(64, 0, device="cuda")
    for knob in ("1", "0"):
        os.environ["BOFI_BOUND_LOOP"] = knob; H.lib().bofi_reload_env()
        with torch.no_grad():
            for _ in range(3):
                model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
            t = [model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")[5] for _ in range(10)]
        print(f"R {R:3d} BOFI_BOUND_LOOP={knob}: {sorted(t)[5] * 1e3:.3f} ms per batch of 64 (loop kernel active: {model.engine().bound_loop_active(R)})", flush=True)
