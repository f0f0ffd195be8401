#!/usr/bin/env python3
"""Time one bofi_linear shape with HIP events inside a hipGraph-free tight loop. usage: mb_time.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H
M, N, K = (int(a) for a in sys.argv[1:4])
tdt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(tdt); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(tdt)
b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=tdt)
def run():
    H.check(H.lib().bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), None, N, H.ptr(y), H.dtype_code(y), N, M, N, K, 0, None, 0, H.stream_ptr()))
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} dbg={os.environ.get('BOFI_GEMM_DBG','0')}: {e0.elapsed_time(e1)*1e3/200:.2f} us/launch (eager, includes launch gaps)")
