#!/usr/bin/env python3
"""Launch one bofi_linear shape a few times (for rocprofv3 --pmc runs).  usage: mb_one.py M N K [iters] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H
M, N, K = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
tdt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "f32") else torch.bfloat16
x = torch.randn(M, K, device="cuda").to(tdt); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(tdt)
b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=tdt)
for _ in range(iters):
    H.check(H.lib().bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), None, N, H.ptr(y), H.dtype_code(y), N, M, N, K, 0, None, 0, H.stream_ptr()))
torch.cuda.synchronize()
