"""Latency of model(..., mode='sample') (NAIC greedy, batch 64, bf16 engine) with the bounding loop enqueued in full and under the adaptive cap:
python dev/exp/sample_latency.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg

sd = W.make_state_dict(cfg, seed=0, preset="full") if "preset" in W.make_state_dict.__code__.co_varnames else W.make_state_dict(cfg, seed=0)
opt = cfg.to_opt()
opt.bofi_compute_dtype = torch.bfloat16
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.cuda().eval()
att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda()
fc = torch.zeros(64, 0, device="cuda")
for cap in (0, None):
    model.opt.bofi_naic_iter_cap = cap
    model.__dict__.pop("_naic_recent", None)
    with torch.no_grad():
        for _ in range(8):
            r = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    ecap = getattr(model.engine(), "_iter_cap", 0)
    print(f"bofi_naic_iter_cap {cap}: {ms:.3f} ms per call of 64 images ({64 / ms * 1e3:.0f} img/s), live iterations {model._naic_recent[-1]}, engine cap {ecap}")
