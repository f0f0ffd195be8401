import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H
shapes = [(5120, 512, 512), (5120, 1536, 512), (5120, 2048, 512), (5120, 512, 2048), (2304, 512, 512), (2304, 1536, 512)]
def timed(fn, iters=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
out = []
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    fn = lambda: H.check(H.lib().bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), None, N, H.ptr(y), 0, N, M, N, K, 0, None, 0, H.stream_ptr()))
    out.append("%dx%dx%d %.2f" % (M, N, K, timed(fn)))
print(os.environ.get("BOFI_GEMM_TILE", "default"), " | ".join(out))
