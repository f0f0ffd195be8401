"""In-kernel timeline of the feed-forward sublayer kernel (workgroup 0): python tools/exp/rb_ffn_stamps.py [M]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = "16"
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 11520
d, dff, dev = 512, 2048, "cuda"
x = torch.randn(M, d, device=dev)
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16); w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
w1p, w2p = pack(w1), pack(w2)
c1, cs1, b2 = torch.randn(dff, device=dev), w1.float().sum(1), torch.randn(d, device=dev)
for _ in range(5):
    H.check(L.bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
H.check(L.bofi_rb_stamps(buf))
t0 = min(buf[w * 16 + 15] for w in range(8))
print(f"M {M}: cycles after the first wavefront's entry; producer wavefront 0 (segment done | chunk handed over), consumer wavefront 4 (chunk arrived | chunk consumed)")
for c in range(8):
    print(f"  chunk {c}: producer {buf[0 * 16 + 2 * c] - t0:7d} {buf[0 * 16 + 2 * c + 1] - t0:7d}   consumer {buf[4 * 16 + 2 * c] - t0:7d} {buf[4 * 16 + 2 * c + 1] - t0:7d}")
