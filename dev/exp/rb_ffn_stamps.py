"""In-kernel timeline of the feed-forward sublayer kernel rb_ffn2_kernel (workgroup 0; s_memtime stamps, BOFI_RB_DBG=16):
    python dev/exp/rb_ffn_stamps.py [M ...]
Per hidden chunk (256 columns) a SIMD runs one producer wavefront (64 rows x 64 hidden columns x K 512 = 256 MFMA 16x16x32) and one
consumer wavefront (64 rows x 128 output columns x K 256 = 256 MFMA): 512 MFMA x 16 cycles = 8 192 cycles if the pipe never waits."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = "16"
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
d, dff, dev = 512, 2048, "cuda"
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16); w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
w1p, w2p = pack(w1), pack(w2)
c1, cs1, b2 = torch.randn(dff, device=dev), w1.float().sum(1), torch.randn(d, device=dev)
for M in [int(v) for v in sys.argv[1:]] or [64, 11520]:
    x = torch.randn(M, d, device=dev)
    run = lambda: H.check(L.bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
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
    t0 = min(buf[(8 + w) * 16] for w in range(8))
    t1 = max(buf[(8 + w) * 16 + 1] for w in range(8))
    print(f"M {M} ({(M + 63) // 64} workgroups): {us:.1f} us per launch (events, back to back); workgroup 0 lives {t1 - t0} s_memtime ticks")
    print("  cycles after the first wavefront's entry; producer wavefront 0 (segment done | chunk handed over), consumer wavefront 4 (chunk arrived | chunk consumed)")
    for c in range(8):
        print(f"  chunk {c}: producer {buf[0 * 16 + 2 * c] - t0:7d} {buf[0 * 16 + 2 * c + 1] - t0:7d}   consumer {buf[4 * 16 + 2 * c] - t0:7d} {buf[4 * 16 + 2 * c + 1] - t0:7d}")
    per = (buf[4 * 16 + 15] - buf[4 * 16 + 1]) / 7.0
    print(f"  steady state: {per:.0f} ticks per chunk at the consumer against 8 192 MFMA cycles per SIMD per chunk")
