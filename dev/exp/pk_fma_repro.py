"""Reproducer of the round-2 wrong-sum finding: the bounding tail's hidden-layer partial sums (dumped through
BOFI_DBG_PART) of three engines stepping CONCURRENTLY on three streams against their own sequential results and a float64 expectation.

    dev/exp/build_slp_variant.sh                       # naic.hip compiled WITH the SLP vectoriser -> boficap_amd/libboficap_hip_slp.so
    BOFI_DBG_PART=1 BOFI_LIB_PATH=boficap_amd/libboficap_hip_slp.so python dev/exp/pk_fma_repro.py     # the suspect build
    BOFI_DBG_PART=1 python dev/exp/pk_fma_repro.py                                                    # the shipped build

(restored from the round-2 history, dev/exp/dbg_step4.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
lib = H.lib()
sd = W.make_state_dict(cfg, seed=0, gen_scale=4.0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
forks = [eng, eng.fork(), eng.fork()]
atts = [torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=500 + i)).cuda().to(torch.bfloat16) for i in range(3)]
L, d, nh = cfg.seq_length + 2, cfg.d_model, 200
ext = torch.zeros(64, L, dtype=torch.int32, device="cuda"); ext[:, 0] = cfg.len_idx
last = torch.ones(64, dtype=torch.int32, device="cuda")
NB = 64 * (8 * nh + d) * 4
def part(e):
    t = torch.empty(64, 8 * nh + d, dtype=torch.float32, device="cuda")
    H.check(lib.bofi_engine_debug_copy(e._h, b"dbg_part", H.ptr(t), NB, H.stream_ptr()))
    return t
for e, a in zip(forks, atts): e.encode(a)
torch.cuda.synchronize()
ref = []
for e in forks:
    o = e.bound_step(ext, last, 36); p = part(e); torch.cuda.synchronize(); ref.append((o[0].clone(), p.clone()))
# expected partial sums from the dumped normalised row and the bf16-rounded hidden weights (float64 on the host)
lp = "model.length_predictor"
w1 = torch.cat([torch.from_numpy(sd[lp + ".Length_classifier1.weight"]), torch.from_numpy(sd[lp + ".Syntactic_classifier1.weight"])], 0)   # [200, 512]
w1 = w1.to(torch.bfloat16).double()
def expected(prow):
    xs = prow[8 * nh:].double().cpu()
    return torch.stack([(w1[:, sl * 64:(sl + 1) * 64] * xs[sl * 64:(sl + 1) * 64]).sum(1) for sl in range(8)]).reshape(-1)
e0 = expected(ref[0][1][5])
print("sequential reference vs float64 expectation, image 5: max abs diff", float((ref[0][1][5, :8 * nh].double().cpu() - e0).abs().max()))
worst = max(float((ref[k][1][r, :8 * nh].double().cpu() - expected(ref[k][1][r])).abs().max()) for k in range(3) for r in range(0, 64, 7))
print("sequential reference, worst over sampled images:", worst)
streams = [torch.cuda.Stream() for _ in forks]
shown = 0
reps = int(os.environ.get('REPS', '1500'))
nbad = [0, 0, 0]
for rep in range(reps):
    outs = []
    for k, (e, st) in enumerate(zip(forks, streams)):
        with torch.cuda.stream(st):
            outs.append(e.bound_step(ext, last, 36))
    torch.cuda.synchronize()
    for k in range(3):
        if not torch.equal(outs[k][0], ref[k][0]):
            nbad[k] += 1
        if not torch.equal(outs[k][0], ref[k][0]) and shown < 6:
            shown += 1
            p = part(forks[k]); torch.cuda.synchronize()
            diff = (p != ref[k][1])
            rows = diff.any(1).nonzero().flatten().tolist()
            for r in rows[:2]:
                idx = diff[r].nonzero().flatten().tolist()
                where = [("xs", i - 8 * nh) if i >= 8 * nh else ("part slice %d unit %d" % (i // nh, i % nh)) for i in idx[:12]]
                ex = expected(p[r])
                print("   errors vs float64: got", [round(float(p[r, i].double().cpu() - ex[i]), 5) for i in idx[:6]], "ref", [round(float(ref[k][1][r, i].double().cpu() - ex[i]), 5) for i in idx[:6]])
                print("engine", k, "row", r, "n diff", len(idx), where, "vals", [(round(float(p[r, i]), 4), round(float(ref[k][1][r, i]), 4)) for i in idx[:6]])
print("library", H.LIB_PATH, "mismatching concurrent steps per engine of", reps, ":", nbad, "failures shown", shown)
