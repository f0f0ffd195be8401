#!/usr/bin/env python3
"""Which Python lines of the XE step launch kernels that are not this library's (fills, adds, copies, index ops)?
One eager step under torch.profiler with stacks; aten ops with device time, grouped by the innermost boficap_amd frame."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity

import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.collate import synthetic_training_batch
from boficap_amd.config import FULL as cfg
from boficap_amd.trainer import XETrainer

mode = sys.argv[1] if len(sys.argv) > 1 else "xe"
dev = torch.device("cuda:0")
sd = W.make_state_dict(cfg, seed=0)
opt = cfg.to_opt()
opt.seed = 42
opt.bofi_train_dtype = torch.bfloat16
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev).train()
tr = XETrainer(model, opt, graph=False)
hb = synthetic_training_batch(cfg, 64, 5, seed=100)
batch = {k: torch.from_numpy(v).to(dev) for k, v in hb.items()}
batch["max_phrase_num"] = int(hb["phrase_num"].max())
batch["max_tokens"] = int((hb["phrase_length"].sum(-1) - 1).max())
batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=7)).to(dev)
batch["att_masks"] = None
batch = tr.add_token_rows(batch, hb)
for _ in range(3):
    tr._forward_backward_eager(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr._forward_backward_eager(batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    own = sum(k.duration for k in ev.kernels)
    if own <= 0:
        continue
    frame = "?"
    node = ev
    while node is not None and frame == "?":                     # the op's own stack, else its callers' (autograd nodes run under backward())
        for fr in node.stack or []:
            if "boficap_amd" in fr or "captioning/" in fr:
                frame = fr.split("/")[-1]
                break
        if frame == "?" and not (node.stack or []):
            nm = getattr(node.cpu_parent, "name", None)
            if nm and not nm.startswith("aten::"):
                frame = "under " + nm[:60]
        node = node.cpu_parent
    key = (ev.name, tuple(k.name[:40] for k in ev.kernels)[:1], frame)
    agg[key][0] += 1
    agg[key][1] += own
tot = 0.0
for (name, kern, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    tot += us
    print(f"{us:8.1f} us {n:4d}x  {name:28s} {kern[0] if kern else '':40s} {frame}")
print(f"listed total {tot:.1f} us")
