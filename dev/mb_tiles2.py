#!/usr/bin/env python3
"""Sweep GEMM tile / ring configurations (BOFI_GEMM_TILE override) over the encoder and fill shapes, inside a hipGraph, rotating over
several operand sets so that a launch finds its operands where the decode finds them (MALL / far L2, not the local L2).
kinds: c = consumer (bf16 out), p = producer (float32 residual in/out + bf16 copy).  usage: mb_tiles2.py [MxNxKxkind ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H

lib = H.lib()
CONFIGS = ["64x64x2x8", "64x64x3x8", "128x64x3x8", "64x64x4x8", "64x64x6x8", "64x64x9x8", "128x64x2x8", "128x64x4x8", "128x64x6x8", "128x128x3x8", "128x128x4x8",
           "64x128x4x8", "64x128x6x8", "256x128x3x8"]
SHAPES = ["2304x1536x512xc", "2304x512x512xp", "2304x2048x512xc", "2304x512x2048xp", "1280x1536x512xc", "1280x512x512xp", "1280x512x512xc",
          "1280x2048x512xc", "1280x512x2048xp", "2304x7168x512xc"]
if len(sys.argv) > 1:
    SHAPES = sys.argv[1:]
NSET, NL = 6, 24
for sh in SHAPES:
    M, N, K, kind = sh.split("x"); M, N, K = int(M), int(N), int(K)
    xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(NSET)]
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(NSET)]
    b = torch.zeros(N, device="cuda")
    if kind == "c":
        ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NSET)]
        def run(i):
            return lib.bofi_linear(H.ptr(xs[i]), H.DT_BF16, K, H.ptr(ws[i]), H.DT_BF16, H.ptr(b), None, N, H.ptr(ys[i]), H.DT_BF16, N, M, N, K, 1, None, 0, H.stream_ptr())
    else:
        rs = [torch.randn(M, N, device="cuda") for _ in range(NSET)]
        y2 = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NSET)]
        def run(i):
            return lib.bofi_linear_ex(H.ptr(xs[i]), H.DT_BF16, K, H.ptr(ws[i]), H.DT_BF16, H.ptr(b), H.ptr(rs[i]), N, H.ptr(rs[i]), H.DT_F32, N, M, N, K, 0,
                                      None, 0, 0.0, 0, None, H.ptr(y2[i]), N, H.stream_ptr())
    res = []
    st = torch.cuda.Stream()
    for cfg in [None] + CONFIGS:
        if cfg is None:
            os.environ.pop("BOFI_GEMM_TILE", None)
        else:
            os.environ["BOFI_GEMM_TILE"] = cfg
        with torch.cuda.stream(st):
            if run(0) != 0:
                continue
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for j in range(NL):
                    run(j % NSET)
            for _ in range(2):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                gr.replay()
            e1.record(st)
            torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) * 1e3 / 10 / NL, cfg or "heuristic"))
    base = [r for r in res if r[1] == "heuristic"][0][0]
    print(f"{sh}: heuristic {base:.1f} us ({2.0*M*N*K/base/1e6:.0f} TF) | " + ", ".join(f"{c} {u:.1f}" for u, c in sorted(res)), flush=True)
