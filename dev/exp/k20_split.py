"""A timed region of EXACTLY 20 batches of 64 (the driver's --steps 20) split over engine launches in different ways: equal launches start in the same phase
(four encoders, then four bounding loops -- 80 workgroups on the chip --, then four filling passes); unequal ones drift apart.
    python dev/exp/k20_split.py "5,5,5,5" "6,6,4,4" "7,5,5,3" "4,4,4,4,4" ...     (a split = batches per launch; launch i runs on stream i % 4; "a+b" = two launches one
    after the other on one stream)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine, pick_concurrent_streams
dev = torch.device("cuda:0")
splits = sys.argv[1:] or ["5,5,5,5", "6,6,4,4", "7,5,5,3", "8,6,4,2", "3+2,3+2,5,5", "2+3,2+3,5,5"]
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64 * 10, max_regions=36, device=dev)
eng.load_state_dict(W.make_state_dict(cfg, seed=0))
streams = pick_concurrent_streams(4, dev)
engines = [eng] + [eng.fork() for _ in range(3)]
for e in engines:
    e.set_decodes_in_flight(4)
pool = torch.from_numpy(W.synthetic_att_feats(64 * 20, 36, cfg.att_feat_size, seed=1235)).to(dev).to(torch.bfloat16)
pn = eng.decode_naic(pool[:640], graph=False, q1_group=64)["phrase_num"]
def region(plan):
    """plan: per stream a list of (features, out) launches"""
    for k, launches in enumerate(plan):
        with torch.cuda.stream(streams[k]):
            for f, o in launches:
                engines[k].decode_naic(f, graph=True, out=o, q1_group=64)
for spec in splits:
    per_stream = [[int(x) for x in s.split("+")] for s in spec.split(",")]
    assert sum(sum(s) for s in per_stream) == 20 and len(per_stream) <= 4, spec
    plan, at = [], 0
    for k, sizes in enumerate(per_stream):
        launches = []
        for c in sizes:
            f = pool[at * 64:(at + c) * 64].contiguous(); at += c
            with torch.cuda.stream(streams[k]):
                o = engines[k].decode_naic(f, graph=True, q1_group=64)
            launches.append((f, o))
        plan.append(launches)
    torch.cuda.synchronize()
    for _ in range(3):
        region(plan)
    torch.cuda.synchronize()
    import time
    t = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        region(plan)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    print(f"{spec:24s} region {t[len(t) // 2]:.3f} ms (min {t[0]:.3f})  -> {1280 / t[len(t) // 2] * 1e3:9.0f} img/s", flush=True)
