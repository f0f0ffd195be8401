#!/usr/bin/env python3
"""In-graph cost of the bounding iteration's per-image tail kernel: BOFI_DBG_TAIL_ONLY=1 reduces bofi_engine_bound_step to
tail(ATTN) + tail(HEADS); 100 steps are captured in a graph and replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BOFI_DBG_TAIL_ONLY", "1")
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
sd = W.make_state_dict(cfg, seed=0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
eng.encode(att)
L = cfg.seq_length + 2
NL = 100
for nkeys in (1, 8, 16):
    ext = torch.zeros(64, L, dtype=torch.int32, device="cuda"); ext[:, 0] = cfg.len_idx; ext[:, 1:nkeys] = 5
    last = torch.full((64,), nkeys, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.bound_step(ext, last, 36); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(NL):
                o = eng.bound_step(ext, last, 36)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); torch.cuda.synchronize()
    print(f"keys={nkeys}: {e0.elapsed_time(e1) * 1e3 / 10 / NL:.2f} us per step (mode {os.environ['BOFI_DBG_TAIL_ONLY']})")
