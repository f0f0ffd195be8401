"""Host time per launch inside DecodePipeline and the end-to-end rate at 20 480 images: python dev/exp/pipeline_probe3.py"""
import sys, time
import torch
sys.path.insert(0, ".")
import captioning.models as models
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd import engine as E
opt = cfg.to_opt(); opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 64, 36
model = models.setup(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}, strict=True)
model.cuda().eval()
u = torch.from_numpy(W.synthetic_att_feats(2048, 36, cfg.att_feat_size, seed=1235)).to(torch.bfloat16)
host = torch.cat([u] * 10).pin_memory()
hb = [host[i:i + 64] for i in range(0, host.size(0), 64)]
tl = {"launch": 0.0, "finish": 0.0, "n": 0}
ol, of = E.DecodePipeline._launch, E.DecodePipeline._finish
def _launch(self, *a):
    t = time.perf_counter(); r = ol(self, *a); tl["launch"] += time.perf_counter() - t; tl["n"] += 1; return r
def _finish(self, launch):
    t = time.perf_counter(); r = list(of(self, launch)); tl["finish"] += time.perf_counter() - t; return iter(r)
E.DecodePipeline._launch, E.DecodePipeline._finish = _launch, _finish
for nf, bpl in ((3, 8), (3, 5), (3, 16), (2, 16), (3, 10)):
    kw = dict(in_flight=nf, batches_per_launch=bpl)
    for _ in model.decode_many(hb[:4 * nf * bpl], **kw):
        pass
    torch.cuda.synchronize()
    tl.update(launch=0.0, finish=0.0, n=0)
    t0 = time.perf_counter(); n = 0
    for r in model.decode_many(hb, **kw):
        n += r["seq"].size(0)
    dt = time.perf_counter() - t0
    print(f"{nf} in flight x {bpl} batches: {n} images {n / dt:9.1f} img/s; {tl['n']} launches, host {tl['launch'] / tl['n'] * 1e3:.3f} ms per _launch, {tl['finish'] / tl['n'] * 1e3:.3f} ms per _finish (incl. waiting), wall {dt / tl['n'] * 1e3:.3f} ms per launch", flush=True)
