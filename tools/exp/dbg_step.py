import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
sd = W.make_state_dict(cfg, seed=0, gen_scale=4.0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
forks = [eng, eng.fork(), eng.fork()]
atts = [torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=500 + i)).cuda().to(torch.bfloat16) for i in range(3)]
L = cfg.seq_length + 2
g = torch.Generator().manual_seed(0)
exts, lasts = [], []
for k in range(3):
    last = torch.randint(1, 15, (64,), generator=g).int()
    ext = torch.zeros(64, L, dtype=torch.int32); ext[:, 0] = cfg.len_idx
    for b in range(64):
        ext[b, 1:int(last[b])] = torch.randint(4, 7, (int(last[b]) - 1,), generator=g).int()
    exts.append(ext.cuda()); lasts.append(last.cuda())
for e, a in zip(forks, atts):
    e.encode(a)
torch.cuda.synchronize()
ref = [tuple(t.clone() for t in e.bound_step(x, l, 36)) for e, x, l in zip(forks, exts, lasts)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in forks]
for mode in ("sequential", "concurrent"):
    bad = [0, 0, 0]
    for rep in range(300):
        outs = []
        for k, (e, st) in enumerate(zip(forks, streams)):
            with torch.cuda.stream(st):
                outs.append(e.bound_step(exts[k], lasts[k], 36))
            if mode == "sequential": torch.cuda.synchronize()
        torch.cuda.synchronize()
        for k in range(3):
            if not (torch.equal(outs[k][0], ref[k][0]) and torch.equal(outs[k][1], ref[k][1])):
                bad[k] += 1
                if bad[k] == 1:
                    rows = ((outs[k][0] != ref[k][0]).any(1) | (outs[k][1] != ref[k][1]).any(1)).nonzero().flatten().tolist()
                    print(mode, "engine", k, "rows differing:", rows[:10], "max diff", float((outs[k][0] - ref[k][0]).abs().max()))
    print(mode, "mismatching calls of 300:", bad)
