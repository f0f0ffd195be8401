"""In-kernel timeline of the persistent bounding-loop kernel (bound_loop.hip; workgroup 0, wavefront 0; s_memtime stamps):
    python dev/exp/bl_stamps.py [iteration] [images]
Prints the stage ends of one iteration, the kernel's span and the whole decode's time with the loop kernel and with the
five-launch iterations (BOFI_BOUND_LOOP=0)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 320
os.environ["BOFI_BL_DBG"] = str(IT + 1)
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL
from boficap_amd.engine import BofiEngine
L = H.lib()
L.bofi_bl_stamps.restype = C.c_int; L.bofi_bl_stamps.argtypes = [C.c_void_p]
sd = W.make_state_dict(FULL, seed=0)
eng = BofiEngine(FULL, torch.bfloat16, max_batch=B, max_regions=36)
eng.load_state_dict(sd)
att = torch.from_numpy(W.synthetic_att_feats(B, 36, FULL.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
NAMES = ["top", "S1 softmax", "S1 ctx", "S2 Wo_self", "norm", "S3 Wq", "S4 cross-attn", "S5 Wo_src", "norm", "S6 W1", "S7 W2", "norm", "S8 hidden", "out layers",
         "log-softmax", "bookkeeping"]
def timed(n=20):
    out = eng.decode_naic(att, graph=True, q1_group=64)
    for _ in range(3):
        eng.decode_naic(att, graph=True, q1_group=64, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng.decode_naic(att, graph=True, q1_group=64, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
ms, out = timed()
buf = (C.c_ulonglong * 32)()
H.check(L.bofi_bl_stamps(buf))
print(f"{B} images, decode alone {ms:.3f} ms; bound_iters {int(out['bound_iters'])}; workgroup 0 ran {buf[31]} iterations in {buf[1] - buf[0]} ticks (entry -> exit)")
print(f"iteration {IT} of workgroup 0 (ticks since the iteration's top; s_memtime):")
for i in range(1, 16):
    print(f"  {NAMES[i]:14s} {buf[2 + i] - buf[2]:8d}  (+{buf[2 + i] - buf[2 + i - 1]})")
os.environ["BOFI_BOUND_LOOP"] = "0"
L.bofi_reload_env()
ms0, _ = timed()
print(f"the same decode with the five-launch iterations (BOFI_BOUND_LOOP=0): {ms0:.3f} ms")
