"""One region count, a few launches one at a time (for rocprofv3 --kernel-trace --stats): python dev/exp/regions_one.py R"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
R = int(sys.argv[1]); B = 320
eng = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
eng.load_state_dict(W.make_state_dict(cfg, seed=0))
eng.set_decodes_in_flight(4)
att = torch.from_numpy(W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=10)).cuda().to(torch.bfloat16)
out = eng.decode_naic(att, graph=True, q1_group=64)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.decode_naic(att, graph=True, q1_group=64, out=out)
torch.cuda.synchronize()
print(f"R {R}: {(time.perf_counter() - t0) * 100:.3f} ms per 320-image launch alone; bound_iters {int(out['bound_iters'])}")
