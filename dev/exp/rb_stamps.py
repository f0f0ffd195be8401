"""In-kernel timeline of the attention sublayer kernel (workgroup 0): BOFI_RB_DBG=16 python dev/exp/rb_stamps.py [B Lq Lk]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = str(int(os.environ.get("BOFI_RB_DBG", "0")) | 16)
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
B, Lq, Lk = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (320, 36, 36)
d, dev = 512, "cuda"
M = B * Lq
qkv = torch.randn(B * max(Lq, Lk), 3 * d, device=dev).to(torch.bfloat16)      # (keys / values: B * Lk rows -- the cross shapes read past B * Lq)
x = torch.randn(M, d, device=dev)
wop = torch.empty(d * d, dtype=torch.bfloat16, device=dev)
w = (torch.randn(d, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(wop), d, d, H.stream_ptr()))
bo = torch.randn(d, device=dev)
klen = torch.full((B,), Lk, dtype=torch.int32, device=dev)
for it in range(5):
    H.check(L.bofi_attn_block(H.ptr(qkv), 3 * d, H.ptr(qkv[:, d:]), 3 * d, H.ptr(qkv[:, 2 * d:]), 3 * d, B, Lq, Lk, H.ptr(klen), 1, 0, 0, 0, H.ptr(wop), H.ptr(bo),
                              H.ptr(x), d, H.ptr(x), d, None, None, H.stream_ptr()))
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
H.check(L.bofi_rb_stamps(buf))
names = ["entry", "consts + barrier", "attention done", "barrier (V dead)", "block written + barrier", "output projection done", "barrier (block dead)", "rows stored"]
nw = 16 if (int(os.environ.get("BOFI_RB_ATTN_W", "0")) or (8 if Lq <= 20 else 16)) == 16 else 8
t0 = min(buf[w * 16] for w in range(nw))
print(f"B {B} Lq {Lq} Lk {Lk}: s_memtime ticks (shader cycles) after the first wavefront's entry; wavefronts 0, {nw // 2 - 1}, {nw // 2}, {nw - 1} of {nw}")
for i, n in enumerate(names):
    print(f"  {n:26s}" + "".join(f"{buf[w * 16 + i] - t0:8d}" for w in (0, nw // 2 - 1, nw // 2, nw - 1)))
