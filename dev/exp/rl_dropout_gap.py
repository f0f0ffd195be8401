#!/usr/bin/env python3
"""How far dropout moves the log-probs of sampled captions: the self-critical step here samples on the inference engine (no dropout)
and differentiates a pass WITH dropout, the reference does both in one stochastic pass.  The per-caption log-ratio
log pi_dropout(s) - log pi_0(s) is the importance weight between the two estimators; this reports its size on the seeded model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import captioning.models as models
from boficap_amd import weights as W, xe
from boficap_amd.config import FULL as cfg
sd = W.with_len_row_shared(W.make_state_dict(cfg, seed=0, gen_scale=1.0), cfg)
opt = cfg.to_opt(); opt.seed = 7
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.cuda().eval()
pool = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda()
fc = torch.zeros(64, 0, device="cuda")
keep = model(fc, pool, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")[2] > 0
att = pool[keep][:10].contiguous()
n = 5
with torch.no_grad():
    saic, naic = model.sample_pair(att, None, n, 1.0)
    P = xe.Params(model)
    base = xe.sampled_logprobs(P, cfg, att, None, saic, naic, sample_n=n, training=False)
    rows = {"SAIC": [], "NAIC": []}
    for trial in range(8):
        lp = xe.sampled_logprobs(P, cfg, att, None, saic, naic, sample_n=n, training=True, seed=1000 + trial)
        for name, s, a, b in (("SAIC", saic, lp[0], base[0]), ("NAIC", naic, lp[1], base[1])):
            seq = s["seq"].long()
            m = (seq > 0).float()
            ta = a.gather(2, seq.unsqueeze(2)).squeeze(2) * m
            tb = b.gather(2, seq.unsqueeze(2)).squeeze(2) * m
            rows[name].append(((ta - tb).abs().sum() / m.sum(), (ta.sum(1) - tb.sum(1))))
for name, r in rows.items():
    tok = torch.stack([x[0] for x in r]).mean()
    ratio = torch.cat([x[1] for x in r])
    print(f"{name}: mean |log p_dropout - log p_0| per sampled token {float(tok):.4f}; per-caption log-ratio mean {float(ratio.mean()):+.3f}, std {float(ratio.std()):.3f}"
          f" (importance weight exp(.) between the two estimators; {ratio.numel()} caption x mask draws)")
