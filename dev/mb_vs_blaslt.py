#!/usr/bin/env python3
"""The path's GEMM shapes on this build's kernel against the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS),
bf16 operands, bias, bf16 output, each timed over graph-free back-to-back launches with HIP events.
A calibration of what the shapes allow on this hardware -- the product never calls the library.  usage: mb_vs_blaslt.py [iters]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
BIG = len(sys.argv) > 2 and sys.argv[2] == "big"      # compute-bound shapes: what the inner loop sustains without the latency of short K
SHAPES = [("cross K|V of all layers", 2304, 7168, 512), ("encoder FFN w_1", 2304, 2048, 512), ("encoder FFN w_2", 2304, 512, 2048),
          ("generator.proj", 1280, 9491, 512), ("att_embed", 2304, 512, 2048), ("q|k|v self", 2304, 1536, 512),
          ("XE decoder rows w_1", 2560, 2048, 512), ("XE vocabulary rows", 2560, 9491, 512)]


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


if BIG:
    SHAPES = [("square", 4096, 4096, 4096), ("wide, short K", 8192, 8192, 512), ("path rows, long K", 2304, 7168, 4096), ("square 8k", 8192, 8192, 8192)]
out = []
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    bb = b.bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ours = lambda: H.check(H.lib().bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), None, N, H.ptr(y),
                                               H.dtype_code(y), N, M, N, K, 0, None, 0, H.stream_ptr()))
    lib = lambda: torch.nn.functional.linear(x, w, bb)
    ours()
    ref = lib()
    err = float((y.float() - ref.float()).abs().max())
    t_ours, t_lib = timed(ours), timed(lib)
    fl = 2.0 * M * N * K
    rec = dict(shape=f"{name}: M={M} N={N} K={K}", ours_us=round(t_ours, 2), library_us=round(t_lib, 2),
               ours_tflops=round(fl / t_ours / 1e6, 1), library_tflops=round(fl / t_lib / 1e6, 1), max_abs_diff=round(err, 4))
    out.append(rec)
    print(json.dumps(rec), flush=True)
