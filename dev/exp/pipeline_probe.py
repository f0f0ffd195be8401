"""Where the time of TransformerModel.decode_many goes: python dev/exp/pipeline_probe.py [images]"""
import sys, time
import torch
sys.path.insert(0, ".")
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 64, 36
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
u = torch.from_numpy(W.synthetic_att_feats(2048, 36, cfg.att_feat_size, seed=1235)).to(torch.bfloat16)
host = torch.cat([u] * (N // 2048)).pin_memory()
dev = host[:20480].cuda()
def run(batches, **kw):
    for _ in model.decode_many(batches[:40], **kw):
        pass
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    for r in model.decode_many(batches, **kw):
        n += r["seq"].size(0)
    dt = time.time() - t0
    return n / dt
hb = [host[i:i + 64] for i in range(0, host.size(0), 64)]
db = [dev[i:i + 64] for i in range(0, dev.size(0), 64)]
print(f"{N} images from pinned host memory, stats on : {run(hb):10.1f} img/s")
print(f"{N} images from pinned host memory, stats off: {run(hb, stats=False):10.1f} img/s")
print(f"20480 images from pinned host memory         : {run(hb[:320]):10.1f} img/s")
print(f"20480 images resident on the device          : {run(db):10.1f} img/s")
print(f"20480 images resident, stats off             : {run(db, stats=False):10.1f} img/s")
for nf, bpl in ((4, 10), (5, 5), (3, 8)):
    print(f"{N} images from host, {nf} in flight x {bpl} batches: {run(hb, in_flight=nf, batches_per_launch=bpl):10.1f} img/s")
