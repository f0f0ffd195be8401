import sys, time
import torch
sys.path.insert(0, ".")
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
N = 40960
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 64, 36
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
u = torch.from_numpy(W.synthetic_att_feats(2048, 36, cfg.att_feat_size, seed=1235)).to(torch.bfloat16)
host = torch.cat([u] * (N // 2048)).pin_memory()
def run(batches, **kw):
    for _ in model.decode_many(batches[:60], **kw):
        pass
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    for r in model.decode_many(batches, **kw):
        n += r["seq"].size(0)
    return n / (time.time() - t0)
hb = [host[i:i + 64] for i in range(0, host.size(0), 64)]
import os
for nf, bpl in ((4, 5), (3, 8), (4, 8), (3, 10)):
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: {N} images from host, {nf} in flight x {bpl} batches: {run(hb, in_flight=nf, batches_per_launch=bpl):10.1f} img/s", flush=True)
