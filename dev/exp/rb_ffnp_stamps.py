"""In-kernel timeline of rb_ffn2_kernel<true> (feed-forward sublayer + the next projection in one launch), workgroup 0: python dev/exp/rb_ffnp_stamps.py [M]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = "16"
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
d, dff, dev, N = 512, 2048, "cuda", 1536
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).bfloat16(); w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).bfloat16(); wq = (torch.randn(N, d, device=dev) / math.sqrt(d)).bfloat16()
w1p, w2p, wqp = pack(w1), pack(w2), pack(wq)
c1, cs1, b2, cq, csq = torch.randn(dff, device=dev), w1.float().sum(1), torch.randn(d, device=dev), torch.randn(N, device=dev), wq.float().sum(1)
for M in [int(v) for v in sys.argv[1:]] or [64, 11520]:
    x, q = torch.randn(M, d, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run = lambda: H.check(L.bofi_ffn_proj_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, H.ptr(wqp), H.ptr(cq), H.ptr(csq),
                                                  H.ptr(q), N, N, M, dff, H.stream_ptr()))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    H.check(L.bofi_rb_stamps(buf))
    t0 = min(buf[(8 + w) * 16] for w in range(8))
    rel = lambda w, k: buf[(8 + w) * 16 + k] - t0
    print(f"M {M}: main loop done {max(rel(w, 4) for w in range(8))}, rows stored {max(rel(w, 2) for w in range(8))}, projection starts {max(rel(w, 3) for w in range(8))}, exit {max(rel(w, 1) for w in range(8))} ticks")
    print("  epilogue per wavefront (tiles staged | first pass stored | tiles staged | rows stored): " + "  ".join(f"w{w}: {rel(w, 5)}|{rel(w, 6)}|{rel(w, 7)}|{rel(w, 2)}" for w in (0, 3, 4, 7)))
    # the same sublayer without the projection
    run0 = lambda: H.check(L.bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
    for _ in range(3):
        run0()
    torch.cuda.synchronize()
    buf0 = (C.c_ulonglong * 256)()
    H.check(L.bofi_rb_stamps(buf0))
    t00 = min(buf0[(8 + w) * 16] for w in range(8))
    r0 = lambda w, k: buf0[(8 + w) * 16 + k] - t00
    print(f"  without the projection: main loop done {max(r0(w, 4) for w in range(8))}, exit {max(r0(w, 1) for w in range(8))}; epilogue " + "  ".join(f"w{w}: {r0(w, 5)}|{r0(w, 6)}|{r0(w, 7)}|{r0(w, 2)}" for w in (0, 3, 4, 7)))
    for w in (0, 4, 7):
        print(f"  wavefront {w}: projection chunks (MFMAs issued | stored): " + "  ".join(f"{buf[w * 16 + 2 * k] - t0}|{buf[w * 16 + 2 * k + 1] - t0}" for k in range(3)))
