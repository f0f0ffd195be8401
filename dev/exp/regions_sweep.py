"""Decode throughput against the number of regions per image (real bottom-up features have 10-100; every BASELINE config has 36):
    python dev/exp/regions_sweep.py
    python dev/exp/regions_sweep.py [images per launch, default 320]
images/s and rows/s of 320-image (or larger) launches, 4 in flight (engine forks, graphs), by R; which kernels serve the attention sublayers and the bounding loop."""
import sys, time
import torch
sys.path.insert(0, ".")
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine, pick_concurrent_streams
sd = W.make_state_dict(cfg, seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
for R in (36, 48, 50, 64, 100):
    root = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
    root.load_state_dict(sd)
    engs = [root] + [root.fork() for _ in range(3)]
    streams = pick_concurrent_streams(4)
    for e in engs:
        e.set_decodes_in_flight(4)
    atts = [torch.from_numpy(W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=10 + k)).cuda().to(torch.bfloat16) for k in range(4)]
    outs = []
    for e, s, a in zip(engs, streams, atts):
        with torch.cuda.stream(s):
            outs.append(e.decode_naic(a, graph=True, q1_group=64))
            e.decode_naic(a, graph=True, q1_group=64, out=outs[-1])
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for j in range(n):
        k = j % 4
        with torch.cuda.stream(streams[k]):
            engs[k].decode_naic(atts[k], graph=True, q1_group=64, out=outs[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"R {R:3d}: {n * B / dt:10.1f} img/s  {n * B * R / dt / 1e6:7.2f} M region rows/s  ({dt / n * 1e3:.3f} ms per {B}-image launch; loop kernel: {root.bound_loop_active(R)})", flush=True)
    del engs, root, outs, atts
    torch.cuda.empty_cache()
