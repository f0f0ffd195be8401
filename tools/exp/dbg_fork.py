import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
sd = W.make_state_dict(cfg, seed=0, gen_scale=4.0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
atts = [torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=500 + i)).cuda().to(torch.bfloat16) for i in range(3)]
def snap(r): return {k: v.clone() for k, v in r.items() if torch.is_tensor(v)}
def same(a, b): return all(torch.equal(a[k], b[k]) for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"))
ref = [snap(eng.decode_naic(a)) for a in atts]; torch.cuda.synchronize()
again = [snap(eng.decode_naic(a)) for a in atts]; torch.cuda.synchronize()
print("eager repeat equal:", [same(a, b) for a, b in zip(ref, again)])
g = [snap(eng.decode_naic(a, graph=True)) for a in atts]; torch.cuda.synchronize()
print("graph (same engine) equal:", [same(a, b) for a, b in zip(ref, g)])
f1 = eng.fork()
fe = [snap(f1.decode_naic(a)) for a in atts]; torch.cuda.synchronize()
print("fork eager sequential equal:", [same(a, b) for a, b in zip(ref, fe)])
forks = [eng, f1, eng.fork()]
streams = [torch.cuda.Stream() for _ in forks]
for mode in ("seq-streams", "concurrent", "concurrent", "concurrent", "concurrent"):
    outs = [None] * 3
    for rep in range(3):
        for k, (e, st) in enumerate(zip(forks, streams)):
            with torch.cuda.stream(st):
                outs[k] = e.decode_naic(atts[k], graph=True, out=outs[k])
            if mode == "seq-streams": torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, [same(r, o) for r, o in zip(ref, outs)], [int((r["phrase_length"] != o["phrase_length"]).any(1).sum()) for r, o in zip(ref, outs)])
for r, o in zip(ref, outs):
    bad = (r["phrase_length"] != o["phrase_length"]).any(1).nonzero().flatten().tolist()
    for i in bad[:3]:
        print("image", i, "ref pl", r["phrase_length"][i].tolist(), "syn", r["phrase_syn"][i].tolist())
        print("       ", "got pl", o["phrase_length"][i].tolist(), "syn", o["phrase_syn"][i].tolist(), "iters", int(r["bound_iters"]), int(o["bound_iters"]))
