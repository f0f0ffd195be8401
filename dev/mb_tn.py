#!/usr/bin/env python3
"""Microbenchmark of the weight-gradient GEMM (bofi_gemm_tn_acc) on the shapes of the XE step: TFLOP/s per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip

lib = hip.lib()
for M, NI, NJ in ((6400, 512, 512), (6400, 1536, 512), (6400, 2048, 512), (6400, 512, 2048), (2304, 1024, 512), (6400, 9536, 512), (2304, 512, 2048)):
    a = torch.randn(M, NI, device="cuda").to(torch.bfloat16)
    b = torch.randn(M, NJ, device="cuda").to(torch.bfloat16)
    c = torch.zeros(NI, NJ, device="cuda")
    run = lambda: hip.check(lib.bofi_gemm_tn_acc(hip.ptr(a), NI, NI, hip.ptr(b), NJ, NJ, hip.ptr(c), NJ, M, NI, NJ, None, hip.stream_ptr()))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"M={M:5d} NI={NI:5d} NJ={NJ:5d}  {us:8.1f} us  {2.0 * M * NI * NJ / us / 1e6:7.1f} TFLOP/s", flush=True)

# the step's weight gradients as ONE grouped call (bofi_gemm_tn_grouped): 24 decoder-sized problems
import ctypes as C
MBM = int(os.environ.get("MB_M", "5120"))                    # rows of the grouped problems (half the rows = half the loop, the same epilogue)
probs = [(MBM, 512, 512)] * 12 + [(MBM, 1536, 512)] * 4 + [(MBM, 2048, 512)] * 4 + [(MBM, 512, 2048)] * 4
ts = [(torch.randn(M, NI, device="cuda").bfloat16(), torch.randn(M, NJ, device="cuda").bfloat16(), torch.zeros(NI, NJ, device="cuda")) for M, NI, NJ in probs]
n = len(probs)
vp, ci = C.c_void_p * n, C.c_int * n
args = (n, vp(*[hip.ptr(t[0]) for t in ts]), ci(*[p_[1] for p_ in probs]), ci(*[p_[1] for p_ in probs]), vp(*[hip.ptr(t[1]) for t in ts]),
        ci(*[p_[2] for p_ in probs]), ci(*[p_[2] for p_ in probs]), vp(*[hip.ptr(t[2]) for t in ts]), ci(*[p_[2] for p_ in probs]),
        ci(*[p_[0] for p_ in probs]), ci(*[p_[1] for p_ in probs]), ci(*[p_[2] for p_ in probs]), None)
run = lambda: hip.check(lib.bofi_gemm_tn_grouped(*args, hip.stream_ptr()))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
fl = sum(2.0 * M * NI * NJ for M, NI, NJ in probs)
print(f"grouped x{n} M={MBM}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  (BOFI_TN_DBG={os.environ.get('BOFI_TN_DBG', '0')}, BOFI_TN_WT={os.environ.get('BOFI_TN_WT', 'auto')})", flush=True)
