"""In-kernel timeline of the persistent GEMM (BOFI_GEMM_DBG bit 64): wavefront 0 of the consumers and of the loaders in workgroups 0..7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import hip as H
lib = H.lib()
M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "11520x2048x512").split("x"))
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
plain = len(sys.argv) > 3 and sys.argv[3] == "plain"          # no folded LayerNorm (no statistics loads in the loaders)
buf = torch.zeros(8 * 2 * 256, dtype=torch.int64, device="cuda")
os.environ["BOFI_GEMM_DBG"] = str(64 | dbg); os.environ["BOFI_GEMM_DBG_BUF"] = str(buf.data_ptr()); os.environ["BOFI_GEMM_PERS_MIN"] = "1"
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); st = torch.rand(M, K // 32, 2, device="cuda") + 1.0; cs = w.float().sum(1)
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for rep in range(3):
    buf.zero_()
    H.check(lib.bofi_linear_fused(H.ptr(x), K, H.ptr(w), H.ptr(bias), None, N, H.ptr(y), H.DT_BF16, N, None, N, None if plain else H.ptr(st), None if plain else H.ptr(cs), 0, None, M, N, K, 1, H.stream_ptr()))
    torch.cuda.synchronize()
b = buf.cpu().view(8, 2, 256)
nk = K // 64
for wg in (0, 5):
    c = b[wg, 0]; l = b[wg, 1]
    t0 = int(min(c[0], l[0]))
    cs_ = [(int(v) - t0) / 100 for v in c if v > 0]; ls = [(int(v) - t0) / 100 for v in l if v > 0]
    print(f"workgroup {wg}: consumer stamps {len(cs_)}, loader stamps {len(ls)} (us from the first stamp)")
    # consumer: per step (before barrier, after barrier), per tile + (before tile barrier, after epilogue)
    i = 0; tile = 0
    while i + 2 * nk + 2 <= len(cs_):
        steps = [(cs_[i + 2 * k], cs_[i + 2 * k + 1]) for k in range(nk)]
        tb, te = cs_[i + 2 * nk], cs_[i + 2 * nk + 1]
        print(f"  consumer tile {tile}: " + " ".join(f"[{a:6.2f} wait {b_ - a:4.2f}]" for a, b_ in steps) + f"  k-loop end {tb:6.2f}  epilogue end {te:6.2f}")
        i += 2 * nk + 2; tile += 1
    i = 0; srow = []
    while i + 3 <= len(ls):
        srow.append((ls[i], ls[i + 1], ls[i + 2])); i += 3
    for t in range(0, len(srow), nk):
        print(f"  loader tile {t // nk}: " + " ".join(f"[landed {a:6.2f} bar +{b_ - a:4.2f} issue +{c_ - b_:4.2f}]" for a, b_, c_ in srow[t:t + nk]))
