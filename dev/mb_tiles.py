#!/usr/bin/env python3
"""Sweep the GEMM tile configurations (BOFI_GEMM_TILE override) over the shapes of the decode and training paths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H

lib = H.lib()
CONFIGS = ["64x64x2x8", "64x64x3x8", "128x64x2x8", "128x64x3x8", "128x128x2x8", "128x128x3x8", "128x128x2x16", "256x128x2x8", "128x256x2x8",
           "128x64x2x4", "128x128x2x4", "64x64x2x4"]
SHAPES = [(6400, 512, 512), (6400, 1536, 512), (6400, 2048, 512), (6400, 512, 2048), (6400, 9491, 512), (6400, 512, 9536), (2304, 1024, 512),
          (2304, 2048, 512), (2304, 512, 2048), (1280, 9491, 512), (2304, 7168, 512)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.zeros(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    res = []
    for cfg in [None] + CONFIGS:
        if cfg is None:
            os.environ.pop("BOFI_GEMM_TILE", None)
        else:
            os.environ["BOFI_GEMM_TILE"] = cfg
        run = lambda: lib.bofi_linear(H.ptr(x), H.DT_BF16, K, H.ptr(w), H.DT_BF16, H.ptr(b), None, N, H.ptr(y), H.DT_F32, N, M, N, K, 0, None, 0, H.stream_ptr())
        if run() != 0:
            continue
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        res.append((us, cfg or "heuristic"))
    base = [r for r in res if r[1] == "heuristic"][0][0]
    res.sort()
    print(f"M={M} N={N} K={K}: heuristic {base:.1f} us ({2.0*M*N*K/base/1e6:.0f} TF) | best " +
          ", ".join(f"{c} {u:.1f}" for u, c in res[:4]), flush=True)
