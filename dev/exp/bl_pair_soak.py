"""Soak of the loop kernel's pair exchange (round 6): many decodes on several forks / streams at once, launch sizes 16 ... 384 images, pair mode against the one-workgroup
results computed beforehand -- slot layouts, ids, iteration counts and status words must be equal every time (the exchange relies on write-through stores + agent-scope loads
being coherent across XCDs without fences).   python dev/exp/bl_pair_soak.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_BOUND_LOOP"] = "2"
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sd = W.make_state_dict(cfg, seed=0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=384, max_regions=36)
eng.load_state_dict(sd)
sizes = [16, 48, 150, 320, 384, 64, 257]
feats = {b: torch.from_numpy(W.synthetic_att_feats(b, 36, cfg.att_feat_size, seed=100 + b)).cuda().to(torch.bfloat16) for b in sizes}
keys = ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters", "bound_saturated")
os.environ["BOFI_BL_PAIR"] = "0"; H.lib().bofi_reload_env()
ref = {}
for b in sizes:
    r = eng.decode_naic(feats[b], strict_q1=False)
    torch.cuda.synchronize()
    ref[b] = {k: r[k].clone() for k in keys}
    ref[b]["lp"] = r["seq_logprob"].nan_to_num().clone()
os.environ["BOFI_BL_PAIR"] = "2"; H.lib().bofi_reload_env()
forks = [eng.fork() for _ in range(4)]
streams = [torch.cuda.Stream() for _ in forks]
bad = pairs = 0
cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
for it in range(rounds):
    outs = []
    for k, (f, s) in enumerate(zip(forks, streams)):
        b = sizes[(it + k) % len(sizes)]
        with torch.cuda.stream(s):
            outs.append((b, f, f.decode_naic(feats[b], strict_q1=False)))
    torch.cuda.synchronize()
    for b, f, r in outs:
        ok = all(torch.equal(r[k], ref[b][k]) for k in keys) and torch.equal(r["seq_logprob"].nan_to_num(), ref[b]["lp"])
        bad += 0 if ok else 1
        H.check(H.lib().bofi_engine_debug_copy(f._h, b"counters", H.ptr(cnt), 32, H.stream_ptr()), "debug_copy")
        torch.cuda.synchronize()
        pairs += int(cnt[5])
    if it % 50 == 0:
        print(f"round {it}: {bad} decodes differ so far, {pairs} groups ran as pairs", flush=True)
print(f"{rounds * len(forks)} decodes in flight on {len(forks)} streams, sizes {sizes}: {bad} differ from the one-workgroup results; {pairs} groups ran as a pair of workgroups")
sys.exit(1 if bad else 0)
