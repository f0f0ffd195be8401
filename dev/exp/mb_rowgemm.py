"""The bounding loop's row GEMMs alone (in-graph time per launch): python dev/exp/mb_rowgemm.py
shapes of one iteration at 320 images: Wo_src (N 512, K 512, residual, copy, stats_out), W1 (N 2048, K 512, LayerNorm fold, ReLU, bf16 out),
W2 (N 512, K 2048 as 4 split-K slabs, residual)."""
import math, sys
import torch
sys.path.insert(0, ".")
from boficap_amd import hip as H
L = H.lib()
dev = "cuda"


def timed(fn, iters=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def case(M, N, K, splitk=1, stats=False, relu=0, residual=False, yb=False, y=True, stats_out=False, skip=False):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    st = torch.rand(M, 32, 2, device=dev) + 1.0 if stats else None
    cs = w.float().sum(1) if stats else None
    res = torch.randn(M, N, device=dev) if residual else None
    yo = torch.empty(splitk, M, N, device=dev) if y else None
    ybo = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if yb else None
    so = torch.empty(M, N // 16, 2, device=dev) if stats_out else None
    sk = torch.zeros(1, dtype=torch.int32, device=dev) if skip else None

    def run():
        H.check(L.bofi_rowgemm(H.ptr(x), K, H.ptr(w), H.ptr(bias), H.ptr(st), 32 if stats else 0, H.ptr(cs), H.ptr(res), N, H.ptr(yo), N, H.ptr(ybo), N,
                               H.ptr(so), M, N, K, splitk, relu, H.ptr(sk), 1 << 30, None, None, H.stream_ptr()))
    t = timed(run)
    print(f"M {M:4d} N {N:5d} K {K:5d} splitk {splitk} stats {int(stats)} relu {relu} res {int(residual)} yb {int(yb)} y {int(y)} so {int(stats_out)} skip {int(skip)}:"
          f" {t:6.2f} us   ({N // 16 * splitk * ((M + 63) // 64)} workgroups)", flush=True)


for M in (320, 64):
    case(M, 512, 512, residual=True, yb=True, stats_out=True, skip=True)
    case(M, 2048, 512, stats=True, relu=1, yb=True, y=False, skip=True)
    case(M, 2048, 512, stats=False, relu=1, yb=True, y=False)
    case(M, 512, 2048, splitk=4, residual=True, skip=True)
    case(M, 512, 512)
    case(M, 1024, 512)
    case(M, 2048, 512)
    case(M, 4096, 512)
