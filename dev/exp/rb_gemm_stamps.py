"""In-kernel timeline of rb_gemm_kernel (workgroup 0; s_memtime stamps = shader clock, BOFI_RB_DBG=16): python dev/exp/rb_gemm_stamps.py
Per 64-column chunk a wavefront issues 256 MFMA (4 096 cycles alone on its SIMD; two wavefronts share a SIMD)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = "16"
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
d, dev = 512, "cuda"
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
for M, N, f32 in ((11520, 1536, 0), (6400, 1536, 0), (6400, 512, 0), (11520, 7168, 0), (6400, 9536, 1)):
    x = torch.randn(M, d, device=dev)
    w = (torch.randn(N, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
    wp, c, cs = pack(w), torch.randn(N, device=dev), w.float().sum(1)
    y = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    run = lambda: H.check(L.bofi_linear_block(H.ptr(x), d, H.ptr(wp), H.ptr(c), H.ptr(cs), H.ptr(y), N, f32, M, N, 0, H.stream_ptr()))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    buf = (C.c_ulonglong * 256)()
    H.check(L.bofi_rb_stamps(buf))
    t0 = min(buf[(8 + w_) * 16] for w_ in range(8))
    staged = max(buf[(8 + w_) * 16 + 1] for w_ in range(8)) - t0
    t1 = max(buf[(8 + w_) * 16 + 2] for w_ in range(8)) - t0
    nch = N // 64
    print(f"M {M} N {N} {'f32' if f32 else 'bf16'} out: {us:.1f} us per launch; workgroup 0: block staged at {staged}, exit at {t1} ticks ({t1 / 2300:.1f} us at 2.3 GHz); {nch} chunks")
    for w_ in (0, 7):
        row = []
        for k in range(8):
            a_, b_ = buf[w_ * 16 + 2 * k], buf[w_ * 16 + 2 * k + 1]
            if a_ >= t0 and b_ >= a_ and b_ - t0 <= t1:
                row.append(f"{a_ - t0}|{b_ - t0}")
        print(f"  wavefront {w_}: chunk MFMAs issued | chunk stored: " + "  ".join(row))
