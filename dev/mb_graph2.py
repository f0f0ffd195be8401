#!/usr/bin/env python3
"""In-graph per-kernel cost of one bofi_linear shape (dependent chain). usage: mb_graph2.py M N K [residual]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H
lib = H.lib()
M, N, K = (int(a) for a in sys.argv[1:4])
NL = 100
x = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16); bias = torch.zeros(N, device="cuda")
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def gemm(): H.check(lib.bofi_linear(H.ptr(x), 1, K, H.ptr(w), 1, H.ptr(bias), None, N, H.ptr(o), 1, N, M, N, K, 0, None, 0, H.stream_ptr()))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gemm(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(NL): gemm()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): gr.replay()
    e1.record(s); torch.cuda.synchronize()
    print(f"M={M} N={N} K={K} dbg={os.environ.get('BOFI_GEMM_DBG','0')} tile={os.environ.get('BOFI_GEMM_TILE','auto')}: {e0.elapsed_time(e1) * 1e3 / 20 / NL:.2f} us per kernel in graph")
