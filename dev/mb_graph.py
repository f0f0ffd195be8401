#!/usr/bin/env python3
"""Per-kernel cost inside a captured graph: N back-to-back dependent launches of one small op."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H
lib = H.lib()
N = 200
x = torch.randn(64, 512, device="cuda"); g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
y = torch.empty(64, 512, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(512, 512, device="cuda") / 22).to(torch.bfloat16); bias = torch.zeros(512, device="cuda"); o = torch.empty(64, 512, device="cuda")
def ln(): H.check(lib.bofi_layernorm(H.ptr(x), H.ptr(g), H.ptr(b), H.ptr(y), 1, 64, 512, H.stream_ptr()))
def gemm(): H.check(lib.bofi_linear(H.ptr(y), 1, 512, H.ptr(w), 1, H.ptr(bias), None, 512, H.ptr(o), 0, 512, 64, 512, 512, 0, None, 0, H.stream_ptr()))
def both():
    ln(); gemm()
for name, fn, per in (("ln64", ln, 1), ("gemm64x512x512", gemm, 1), ("ln+gemm", both, 2)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(N): fn()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): gr.replay()
        e1.record(s); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) * 1e3 / 20 / (N * per):.2f} us per kernel inside a graph of {N * per} kernels")
