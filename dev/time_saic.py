#!/usr/bin/env python3
"""Time the SAIC greedy decode at the benchmark shape (B=64, bf16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
sd = W.make_state_dict(cfg, 0)
if "--multi" in sys.argv:                   # weights with which the mode lays out several phrases (tests/golden/tiny_saic_multi)
    sd = W.with_len_row_shared(sd, cfg)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
if "--multi" in sys.argv:                   # keep images that open a phrase (one without NaN-halts the batch, as in the reference)
    pool = torch.from_numpy(W.synthetic_att_feats(256, 36, cfg.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
    pn = torch.cat([eng.decode_naic(c)["phrase_num"].clone() for c in pool.split(64)])
    att = pool[pn > 0][:64].contiguous()
graph = "--no-graph" not in sys.argv
r = eng.decode_saic(att, graph=graph); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): r = eng.decode_saic(att, graph=graph, out=r)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"SAIC greedy B=64 bf16: {dt*1e3:.2f} ms/batch = {64/dt:.0f} images/s, iterations {int(r['bound_iters'])}, NaN {bool(r['seq_logprob'].isnan().any())}, "
      f"tokens/image {float((r['seq'] > 0).sum(1).float().mean()):.1f}, phrases/image {float(r['phrase_num'].float().mean()):.1f}")
# the same decode with the loop enqueued for (live iterations + 2, in steps of 4) iterations, and its continuation when needed
live = int(r["bound_iters"])
cap = min(cfg.seq_length, -(-(live + 2) // 4) * 4)
if cap < cfg.seq_length:
    r2 = eng.decode_saic(att, graph=graph, it_range=(1, cap)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r2 = eng.decode_saic(att, graph=graph, out=r2, it_range=(1, cap))
        if int(r2["bound_iters"]) >= cap:
            r2 = eng.decode_saic(att, graph=graph, out=r2, it_range=(cap + 1, cfg.seq_length))
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / 10
    same = all(torch.equal(r[k], r2[k]) for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"))
    print(f"  with {cap} of {cfg.seq_length} iterations enqueued (the count of live iterations read back after every decode): {dt2*1e3:.2f} ms/batch, same results: {same}")
