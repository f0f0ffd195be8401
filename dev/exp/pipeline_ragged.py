"""decode_many on ragged loader batches (region counts 10 .. 100 per image, every batch clipped to its longest image, masks given):
    python dev/exp/pipeline_ragged.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 64, 100
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
rng = np.random.default_rng(0)
pool = torch.from_numpy(W.synthetic_att_feats(64, 100, cfg.att_feat_size, seed=5)).to(torch.bfloat16)
items, n_rows = [], 0
for _ in range(160):
    lens = np.clip(rng.normal(45, 20, 64).round().astype(np.int64), 10, 100)          # bottom-up adaptive features: 10 .. 100 boxes
    rmax = int(lens.max())
    m = (np.arange(rmax)[None] < lens[:, None]).astype(np.float32)
    items.append((pool[:, :rmax].contiguous().pin_memory(), torch.from_numpy(m)))
    n_rows += int(lens.sum())
for _ in model.decode_many(items[:80]):
    pass
torch.cuda.synchronize()
t0 = time.time(); n = 0
for r in model.decode_many(items):
    n += r["seq"].size(0)
dt = time.time() - t0
print(f"{n} images in {len(items)} ragged batches (mean {n_rows / n:.1f} regions per image, batches clipped to {min(i[0].shape[1] for i in items)} .. {max(i[0].shape[1] for i in items)}): "
      f"{n / dt:.1f} img/s, {n_rows / dt / 1e6:.2f} M real region rows/s")
