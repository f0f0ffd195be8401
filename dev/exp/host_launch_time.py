"""Host time of one graphed engine launch call (BofiEngine.decode_naic(..., graph=True, out=...)): what the fourth stream's launch waits for in a
four-launch region.  python dev/exp/host_launch_time.py"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
sd = W.make_state_dict(cfg, seed=0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=320, max_regions=36); eng.load_state_dict(sd)
att = torch.from_numpy(W.synthetic_att_feats(320, 36, cfg.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
engines = [eng] + [eng.fork() for _ in range(3)]
streams = [torch.cuda.Stream() for _ in engines]
outs = []
for e, st in zip(engines, streams):
    with torch.cuda.stream(st):
        outs.append(e.decode_naic(att, graph=True, q1_group=64, iter_cap=12))
        e.decode_naic(att, graph=True, out=outs[-1], q1_group=64, iter_cap=12)
torch.cuda.synchronize()
for rep in range(5):
    t = [time.perf_counter()]
    for e, st, o in zip(engines, streams, outs):
        with torch.cuda.stream(st):
            e.decode_naic(att, graph=True, out=o, q1_group=64, iter_cap=12)
        t.append(time.perf_counter())
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    print("host us per launch call:", [round((b - a) * 1e6, 1) for a, b in zip(t[:-2], t[1:-1])], " all four enqueued after", round((t[-2] - t[0]) * 1e6, 1), "us; region", round((t[-1] - t[0]) * 1e3, 3), "ms")

# (measured, round 4: 45-60 us per call, all four enqueued after 210-290 us of a 5.9 ms region; enqueuing from four host threads is slower -- 400-870 us)
