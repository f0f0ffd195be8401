"""DecodePipeline at 16-20 batches per launch: images/s with the features on the device, in pinned host memory, with and without the host-side results
(stats / clones), 3 and 4 launches in flight (GPU_MAX_HW_QUEUES=8 set here: launch streams + the copy stream on queues of their own)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
N = 81920
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 64, 36
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
u = torch.from_numpy(W.synthetic_att_feats(2048, 36, cfg.att_feat_size, seed=1235)).to(torch.bfloat16)
host = torch.cat([u] * (N // 2048)).pin_memory()
devt = host[:20480].cuda()
def run(batches, **kw):
    for _ in model.decode_many(batches[:128], **kw):
        pass
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    for r in model.decode_many(batches, **kw):
        n += r["seq"].size(0)
    return n / (time.time() - t0)
hb = [host[i:i + 64] for i in range(0, host.size(0), 64)]
db = [devt[i % 20480:i % 20480 + 64] for i in range(0, N, 64)]
for nf, bpl in ((3, 16), (4, 16), (3, 20), (4, 20)):
    for name, b, kw in (("device-resident", db, {}), ("pinned host", hb, {}), ("pinned host, no stats", hb, {"stats": False})):
        print(f"{nf} in flight x {bpl} batches, {name:22s}: {run(b, in_flight=nf, batches_per_launch=bpl, **kw):10.1f} img/s", flush=True)
