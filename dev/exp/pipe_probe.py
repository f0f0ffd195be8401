"""Software-pipelined decodes (round 5 experiment): the bounding loop of decode j runs on a side stream while the launch stream already encodes
decode j + NS on the stream's second engine fork.
    python dev/exp/pipe_probe.py [launch streams] [loop streams] [launches] [batches per launch]
Prints images/s of (a) whole decodes, NS in flight (the bench's form) and (b) the pipelined form, same inputs, same engines' kernels."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")               # (launch + side streams on hardware queues of their own: the runtime's default is 4)
import torch
sys.path.insert(0, ".")
from boficap_amd import weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine, pick_concurrent_streams
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 80
C = int(sys.argv[4]) if len(sys.argv) > 4 else 5
B = 64 * C
dev = torch.device("cuda:0")
sd = W.make_state_dict(cfg, seed=0)
root = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=36, device=dev)
root.load_state_dict(sd)
streams = pick_concurrent_streams(NS + NL, dev, candidates=32)
assert len(streams) >= NS + NL, len(streams)
heavy, side = streams[:NS], streams[NS:NS + NL]
eng = [[root if (k == 0 and p == 0) else root.fork() for p in range(2)] for k in range(NS)]
for row in eng:
    for e in row:
        e.set_decodes_in_flight(NS)
atts = [[torch.from_numpy(W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=1235 + 7 * k + 101 * p)).to(dev).to(torch.bfloat16).contiguous() for p in range(2)] for k in range(NS)]
# (quirk Q1: a batch whose last image lays out nothing decodes to NaN -- irrelevant for timing)
outs = [[None, None] for _ in range(NS)]
for k in range(NS):
    for p in range(2):
        with torch.cuda.stream(heavy[k]):
            o = eng[k][p].decode_naic(atts[k][p], graph=True, q1_group=64)
            for ph in ("e", "b", "f"):
                eng[k][p].decode_naic(atts[k][p], graph=True, q1_group=64, out=o, phases=ph)
            outs[k][p] = o
torch.cuda.synchronize()
whole = {(k, p): {n: v.clone() for n, v in outs[k][p].items() if torch.is_tensor(v)} for k in range(NS) for p in range(2)}

def run_whole(n):
    for j in range(n):
        k, p = j % NS, (j // NS) % 2
        with torch.cuda.stream(heavy[k]):
            eng[k][p].decode_naic(atts[k][p], graph=True, q1_group=64, out=outs[k][p])

def run_piped(n):
    ev_e = [[torch.cuda.Event() for _ in range(2)] for _ in range(NS)]
    ev_l = [[torch.cuda.Event() for _ in range(2)] for _ in range(NS)]
    pend = [None] * NS                                           # the decode whose filling pass this stream still owes
    for j in range(n):
        k, p = j % NS, (j // NS) % 2
        hs, ls = heavy[k], side[j % NL]
        with torch.cuda.stream(hs):
            eng[k][p].decode_naic(atts[k][p], graph=True, q1_group=64, out=outs[k][p], phases="e")
            ev_e[k][p].record(hs)
        with torch.cuda.stream(ls):
            ls.wait_event(ev_e[k][p])
            eng[k][p].decode_naic(atts[k][p], graph=True, q1_group=64, out=outs[k][p], phases="b")
            ev_l[k][p].record(ls)
        if pend[k] is not None:
            q = pend[k]
            with torch.cuda.stream(hs):
                hs.wait_event(ev_l[k][q])
                eng[k][q].decode_naic(atts[k][q], graph=True, q1_group=64, out=outs[k][q], phases="f")
        pend[k] = p
    for k in range(NS):
        if pend[k] is not None:
            q = pend[k]
            with torch.cuda.stream(heavy[k]):
                heavy[k].wait_event(ev_l[k][q])
                eng[k][q].decode_naic(atts[k][q], graph=True, q1_group=64, out=outs[k][q], phases="f")

for name, fn in (("whole decodes", run_whole), ("pipelined (loop on side streams)", run_piped), ("whole decodes", run_whole), ("pipelined (loop on side streams)", run_piped)):
    fn(2 * NS * 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(N)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(outs[k][p]["seq"], whole[(k, p)]["seq"]) and torch.equal(outs[k][p]["phrase_length"], whole[(k, p)]["phrase_length"]) for k in range(NS) for p in range(2))
    print(f"{name:36s}: {N * B / dt:10.1f} img/s   {dt / N * 1e3:.4f} ms per launch of {B} images   (results equal the first decodes: {ok})")
