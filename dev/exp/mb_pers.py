"""Persistent (gemm_pers.hip) against one-tile-per-workgroup (gemm_glds.hip) GEMM on the decode's large shapes: bofi_linear_fused,
LayerNorm folded in / bf16 out (the qkv, w_1, kv_all form) and residual + statistics out (the output-projection / w_2 form).
Operands rotate over 4 sets (Infinity Cache, not L2, as in the decode); time = HIP events over 40 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import hip as H
lib = H.lib()
SHAPES = [(11520, 1536, 512, "ln"), (11520, 2048, 512, "ln"), (11520, 6144, 512, "ln"), (11520, 512, 512, "res"), (11520, 512, 2048, "res"),
          (9216, 2048, 512, "ln"), (5760, 1536, 512, "ln"), (5760, 2048, 512, "ln"), (5760, 512, 2048, "res"), (2304, 2048, 512, "ln"), (2304, 512, 2048, "res")]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) if v.isdigit() else v for v in a.split("x")) for a in sys.argv[1:]]
NSET, REP = 4, 40
for M, N, K, mode in SHAPES:
    sets = []
    for i in range(NSET):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        if mode == "ln":
            st = torch.rand(M, K // 32, 2, device="cuda") + 1.0
            cs = w.float().sum(1)
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            args = (H.ptr(x), K, H.ptr(w), H.ptr(bias), None, N, H.ptr(y), H.DT_BF16, N, None, N, H.ptr(st), H.ptr(cs), 0, None, M, N, K, 1)
            keep = (x, w, bias, st, cs, y)
        else:
            res = torch.randn(M, N, device="cuda")
            y2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            so = torch.empty(M, N // 32, 2, device="cuda")
            args = (H.ptr(x), K, H.ptr(w), H.ptr(bias), H.ptr(res), N, H.ptr(res), H.DT_F32, N, H.ptr(y2), N, None, None, 0, H.ptr(so), M, N, K, 0)
            keep = (x, w, bias, res, y2, so)
        sets.append((args, keep))
    out = []
    for pers in ("0", "1"):
        os.environ["BOFI_GEMM_PERS"] = pers
        os.environ["BOFI_GEMM_PERS_MIN"] = "1"
        lib.bofi_reload_env()
        s = H.stream_ptr()
        for i in range(8):
            H.check(lib.bofi_linear_fused(*sets[i % NSET][0], s))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(REP):
            H.check(lib.bofi_linear_fused(*sets[i % NSET][0], s))
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / REP)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K} {mode}: one-tile {out[0]:7.1f} us ({fl / out[0] / 1e6:5.0f} TF)   persistent {out[1]:7.1f} us ({fl / out[1] / 1e6:5.0f} TF)   x{out[0] / out[1]:.2f}", flush=True)
