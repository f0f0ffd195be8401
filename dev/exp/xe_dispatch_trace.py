#!/usr/bin/env python3
"""Which lines of boficap_amd ask torch for a device kernel during one eager XE step (forward + criterion + backward)?  A TorchDispatchMode
records every aten op that launches work, with the innermost boficap_amd frame of the Python stack (autograd runs single-threaded here so that
backward's custom Functions are seen too).  python dev/exp/xe_dispatch_trace.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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

VIEW = ("view", "reshape", "expand", "permute", "transpose", "select", "slice", "unsqueeze", "squeeze", "detach", "alias", "as_strided", "t.default", "_unsafe_view",
        "unbind", "split", "narrow", "size", "stride", "is_", "sym_", "_local_scalar", "lift_fresh", "empty", "_to_copy")
sites = collections.Counter()

class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            own = [f for f in traceback.extract_stack() if "boficap_amd" in f.filename and "xe_dispatch" not in f.filename]
            site = f"{os.path.basename(own[-1].filename)}:{own[-1].lineno} {own[-1].name}" if own else "(outside boficap_amd)"
            shp = ",".join("x".join(map(str, a.shape)) for a in args if torch.is_tensor(a)) if os.environ.get("XE_TRACE_SHAPES") else ""
            sites[(name.replace("aten.", "") + (" [" + shp + "]" if shp else ""), site)] += 1
        return func(*args, **(kwargs or {}))

with torch.autograd.set_multithreading_enabled(False):
    with Trace():
        tr._forward_backward_eager(batch)
torch.cuda.synchronize()
print(f"{sum(sites.values())} aten ops that are not views, by (op, innermost boficap_amd line):")
for (op, site), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {op:60s} {site}")
