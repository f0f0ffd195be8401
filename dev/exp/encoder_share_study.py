"""Where does the ENCODER's share of the bound heads' bf16 error come from?  A CPU study on the float32 oracle (test infrastructure, not product): only the
encoder's WEIGHTS are rounded to bf16 -- att_embed, the six layers, both -- the activations stay float32, and the first bounding step's live-class log-probs are compared with the
all-float32 ones.  python dev/exp/encoder_share_study.py"""
import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import boficap_oracle as O
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
torch.set_num_threads(8)
sd = W.make_state_dict(cfg, seed=0)
w = O.as_torch(sd)
B = 32
att = torch.from_numpy(W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=99)).bfloat16().float()
L = cfg.seq_length + 2
ext0 = torch.zeros(B, L, dtype=torch.long); ext0[:, 0] = cfg.len_idx
tm = torch.zeros(B, L, L, dtype=torch.bool); tm[:, :, 0] = True
live_len, live_syn = [0, 1, 2, 3, 4, 9], [1, 4, 5, 6]
def heads(wd):
    with torch.no_grad():
        mem, sm = O.memory_of(wd, cfg, att)
        _, llp, _, slp = O.bound_step_na(w, cfg, ext0, mem, sm, tm)      # bounding layer always with the float32 weights: the ENCODER's share
    return llp, slp, mem
l0, s0, m0 = heads(w)
def rounded(sel):
    wd = dict(w)
    for k, v in w.items():
        if sel(k) and v.dim() == 2:
            wd[k] = v.bfloat16().float()
    return wd
for name, sel in (("att_embed weight only", lambda k: k.startswith("att_embed")),
                  ("encoder layers' weights only", lambda k: k.startswith("model.encoder")),
                  ("att_embed + encoder weights", lambda k: k.startswith("att_embed") or k.startswith("model.encoder"))):
    l1, s1, m1 = heads(rounded(sel))
    print(f"{name:34s}: memory |d| {float((m1 - m0).abs().max()):.4f}  len-head live |dlogp| {float((l1 - l0)[:, live_len].abs().max()):.4f}  syn-head {float((s1 - s0)[:, live_syn].abs().max()):.4f}")
