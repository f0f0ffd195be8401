"""Host time of one replayed decode_naic call (Python wrapper + hipGraphLaunch) and of a region of four of them on four streams, against the device time of the region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine, pick_concurrent_streams
dev = torch.device("cuda:0")
eng = BofiEngine(cfg, torch.bfloat16, max_batch=320, max_regions=36, device=dev)
eng.load_state_dict(W.make_state_dict(cfg, seed=0))
streams = pick_concurrent_streams(4, dev)
engines = [eng] + [eng.fork() for _ in range(3)]
for e in engines:
    e.set_decodes_in_flight(4)
att = torch.from_numpy(W.synthetic_att_feats(320, 36, cfg.att_feat_size, seed=1235)).to(dev).to(torch.bfloat16)
outs = []
for e, s in zip(engines, streams):
    with torch.cuda.stream(s):
        outs.append(e.decode_naic(att, graph=True, q1_group=64))
        e.decode_naic(att, graph=True, q1_group=64, out=outs[-1])
torch.cuda.synchronize()
def region():
    for e, s, o in zip(engines, streams, outs):
        with torch.cuda.stream(s):
            e.decode_naic(att, graph=True, q1_group=64, out=o)
for _ in range(3):
    region(); torch.cuda.synchronize()
host, total = [], []
for _ in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    region()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"region of 4 replayed decodes: host time to issue {host[7]:.3f} ms (min {host[0]:.3f}), region {total[7]:.3f} ms (min {total[0]:.3f})")
