"""In-kernel timeline of the persistent feed-forward sublayer kernel rb_ffn3_kernel (workgroup 0; s_memtime stamps, BOFI_RB_DBG=16):
    BOFI_RB_FFN_BPW=3 python dev/exp/rb_ffn_stamps.py [M ...]
Per hidden chunk (256 columns) a SIMD runs one producer wavefront (64 rows x 64 hidden columns x K 512 = 256 MFMA 16x16x32) and one
consumer wavefront (64 rows x 128 output columns x K 256 = 256 MFMA): 512 MFMA x 16 cycles = 8 192 cycles if the pipe never waits;
a 64-row block is 8 chunks = 65.5 k cycles of MFMA issue per SIMD.
Stamps: producer wavefront 0: entry, then per block [staging starts, block staged, (segment done, chunk handed over) x 8];
consumer wavefront 4: entry, then per block [(chunk arrived, chunk consumed) x 8, block closed]."""
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
    def row(base):
        out = []
        for i in range(128):
            if buf[base + i] == 0:
                break
            out.append(buf[base + i])
        return out
    P, Cn = row(0), row(128)
    t0 = min(P[0], Cn[0])
    print(f"M {M} ({(M + 63) // 64} row blocks, BOFI_RB_FFN_BPW={os.environ.get('BOFI_RB_FFN_BPW', 'default')}): {us:.1f} us per launch (events, back to back); "
          f"workgroup 0: {max(P[-1], Cn[-1]) - t0} ticks from entry to its last stamp")
    nb = (len(P) - 1) // 18
    for j in range(nb):
        p = P[1 + 18 * j: 1 + 18 * (j + 1)]
        c = Cn[1 + 17 * j: 1 + 17 * (j + 1)]
        print(f"  block {j}: staging {p[0] - t0:7d} .. {p[1] - t0:7d}   closed by the consumer at {c[16] - t0:7d}" if len(c) == 17 else f"  block {j}: (stamps ran out)")
        for k in range(8):
            if len(c) == 17:
                print(f"    chunk {k}: producer {p[2 + 2 * k] - t0:7d} {p[3 + 2 * k] - t0:7d}   consumer {c[2 * k] - t0:7d} {c[2 * k + 1] - t0:7d}")
    if nb >= 2 and len(Cn) >= 1 + 17 * nb:
        per = (Cn[17 * nb] - Cn[17]) / (nb - 1)
        print(f"  block to block (closed -> closed): {per:.0f} ticks against 65 536 MFMA cycles per SIMD per block")
