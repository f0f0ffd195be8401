#!/usr/bin/env python3
"""Which launches of an XE step are not this library's, and which line of boficap_amd asks for them (torch.profiler, one eager step)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.collate import synthetic_training_batch
from boficap_amd.config import FULL as cfg
from boficap_amd.trainer import XETrainer
opt = cfg.to_opt(); opt.seed = 42; opt.bofi_train_dtype = torch.bfloat16
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, seed=0).items()}, strict=True)
model.cuda().train()
tr = XETrainer(model, opt, graph=False)
hb = synthetic_training_batch(cfg, 64, 5, seed=100)
batch = {k: torch.from_numpy(v).cuda() for k, v in hb.items()}
batch["max_phrase_num"] = int(hb["phrase_num"].max()); batch["max_tokens"] = int((hb["phrase_length"].sum(-1) - 1).max())
batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda(); batch["att_masks"] = None
batch = tr.add_token_rows(batch, hb)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr._forward_backward_eager(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 0 and e.count >= 1:
        own = [f for f in (e.stack or []) if "boficap_amd" in f]
        rows.append((e.count, e.key, (own[0].split("boficap_amd/")[-1] if own else "(autograd / other)")[:110], e.device_time_total))
rows.sort(key=lambda r: -r[0])
tot = collections.Counter()
for c, k, site, t in rows:
    tot[k] += c
print("launching aten ops per step:", dict(tot.most_common(14)))
for c, k, site, t in rows[:40]:
    print(f"{c:4d}  {k:24s} {t:9.1f} us  {site}")
