"""Where the bf16 engine's bound-head log-prob error comes from: the first bounding step of the FULL preset on 64 images,
the engine's own intermediate values fed into the float32 oracle at three seams:
    (a) heads alone      : the oracle's float32 row-0 vector through heads whose hidden-layer weights are rounded to bf16 (what the engine stores)
    (b) before the heads : the ENGINE's normed row-0 vector (the tail kernel's `xs`, BOFI_DBG_PART) through the oracle's float32 heads
    (c) encoder only     : the ENGINE's memory (float32 copy of its encoder output) through the oracle's float32 bounding layer and heads
(b) - (c) is the bounding layer's own chain (cross K|V projection, query projection, attention, output projection, feed-forward).
    BOFI_DBG_PART=1 python dev/exp/bound_head_attribution.py [row-block]"""
import os, sys
sys.path.insert(0, ".")
os.environ["BOFI_DBG_PART"] = "1"
os.environ["BOFI_RB_MIN_ROWS"] = "0" if "row-block" in sys.argv else "1000000000"
import torch
import torch.nn.functional as F
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
from oracle import boficap_oracle as O

B = 64
sd = W.make_state_dict(cfg, seed=0, gen_scale=1.0)          # the preset of test_bf16_logits_within_tolerance_on_every_image
w = O.as_torch(sd)
att_np = W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=99)
L = cfg.seq_length + 2
ext0 = torch.zeros(B, L, dtype=torch.long); ext0[:, 0] = cfg.len_idx
tm = torch.zeros(B, L, L, dtype=torch.bool); tm[:, :, 0] = True
with torch.no_grad():
    memory, src_mask = O.memory_of(w, cfg, torch.from_numpy(att_np))
    emb = O.add_pe(w, O.embed(w, "model.syn_embed", ext0, cfg.d_model))
    row0 = O.bound_row0(w, cfg, emb, memory, src_mask, tm)
    _, o_llp, _, o_slp = O.bound_heads(w, row0)

eng = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=36)
eng.load_state_dict(sd)
att = torch.from_numpy(att_np).cuda().to(torch.bfloat16)
mem_e = eng.encode(att).cpu()
llp, slp = eng.bound_step(ext0.to(torch.int32).cuda(), torch.ones(B, dtype=torch.int32, device="cuda"), 36)
nh = 2 * cfg.head_hidden
t = torch.empty(B, 8 * nh + cfg.d_model, device="cuda")
H.check(H.lib().bofi_engine_debug_copy(eng._h, b"dbg_part", H.ptr(t), t.numel() * 4, H.stream_ptr()))
torch.cuda.synchronize()
xs_e = t[:, 8 * nh:].cpu()

live_len, live_syn = [0, 1, 2, 3, 4, 9], [1, 4, 5, 6]
def err(l, s):
    return float((l - o_llp)[:, live_len].abs().max()), float((s - o_slp)[:, live_syn].abs().max()), float((l - o_llp).abs().max()), float((s - o_slp).abs().max())
def show(name, l, s):
    a, b, c, d = err(l, s)
    print(f"{name:58s} live classes: len {a:.4f} syn {b:.4f}   all classes: len {c:.4f} syn {d:.4f}")

bf = lambda x: x.to(torch.bfloat16).float()
lp = "model.length_predictor"
def heads(out, round_w1):
    res = []
    for nm in ("Length", "Syntactic"):
        w1, b1 = w[f"{lp}.{nm}_classifier1.weight"], w[f"{lp}.{nm}_classifier1.bias"]
        w2, b2 = w[f"{lp}.{nm}_classifier2.weight"], w[f"{lp}.{nm}_classifier2.bias"]
        h = F.relu(out @ (bf(w1) if round_w1 else w1).T + b1)
        res.append(F.log_softmax(h @ w2.T + b2, -1))
    return res
with torch.no_grad():
    print(f"row-0 vector in front of the heads: |oracle| max {float(row0.abs().max()):.2f}; engine - oracle max {float((xs_e - row0).abs().max()):.4f}, rms {float((xs_e - row0).pow(2).mean().sqrt()):.5f}")
    print(f"memory: |oracle| max {float(memory.abs().max()):.2f}; engine - oracle max {float((mem_e - memory).abs().max()):.4f}, rms {float((mem_e - memory).pow(2).mean().sqrt()):.5f}")
    show("engine (bf16), as the test measures it", llp.cpu(), slp.cpu())
    show("(a) heads alone: float32 row 0, hidden weights rounded to bf16", *heads(row0, True))
    show("(b) engine's row 0 through float32 heads", *heads(xs_e, False))
    show("    engine's row 0 through heads with bf16 hidden weights", *heads(xs_e, True))
    row0_c = O.bound_row0(w, cfg, emb, mem_e, src_mask, tm)
    lc, sc = heads(row0_c, False)
    show("(c) engine's memory through the float32 bounding layer + heads", lc, sc)
    d_l, d_s = (llp.cpu() - lc), (slp.cpu() - sc)
    print(f"(d) the bounding kernels' OWN error -- engine against the float32 bounding layer + heads on the engine's memory: live classes len "
          f"{float(d_l[:, live_len].abs().max()):.4f} syn {float(d_s[:, live_syn].abs().max()):.4f}   all classes: len {float(d_l.abs().max()):.4f} syn {float(d_s.abs().max()):.4f}")
    lb, sb = heads(xs_e, False)
    d_l, d_s = (lb - lc), (sb - sc)
    print(f"    the same with float32 head weights (engine's row 0 through float32 heads): live classes len {float(d_l[:, live_len].abs().max()):.4f} syn "
          f"{float(d_s[:, live_syn].abs().max()):.4f}   all classes: len {float(d_l.abs().max()):.4f} syn {float(d_s.abs().max()):.4f}")
    # the heads' sensitivity: d logit / d row0 (largest singular direction) -- how much a unit error of the row-0 vector can move a log-prob
    w1, w2 = w[f"{lp}.Length_classifier1.weight"], w[f"{lp}.Length_classifier2.weight"]
    print(f"length head: ||W2|| {float(torch.linalg.matrix_norm(w2, 2)):.1f}, ||W1|| {float(torch.linalg.matrix_norm(w1, 2)):.2f}, logit span {float(o_llp.max() - o_llp.min()):.1f}")
