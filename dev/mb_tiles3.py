#!/usr/bin/env python3
"""Throughput form of mb_tiles2.py: four streams replay a graph of the same GEMM (own operand sets) at once, as four decodes in
flight do; reports the aggregate time per GEMM.  usage: mb_tiles3.py [MxNxKxkind ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boficap_amd import hip as H

lib = H.lib()
CONFIGS = ["64x64x2x8", "64x64x4x8", "128x64x2x8", "128x64x4x8", "128x128x3x8", "128x128x4x8", "64x128x4x8", "256x128x3x8", "128x128x2x8", "256x128x2x8"]
SHAPES = ["2304x1536x512xc", "2304x512x512xp", "2304x2048x512xc", "2304x512x2048xp", "1280x1536x512xc", "1280x512x512xp",
          "1280x2048x512xc", "1280x512x2048xp", "2304x7168x512xc", "1280x9491x512xf"]
if len(sys.argv) > 1:
    SHAPES = sys.argv[1:]
NSTR, NSET, NL = 4, 2, 16
for sh in SHAPES:
    M, N, K, kind = sh.split("x"); M, N, K = int(M), int(N), int(K)
    xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(NSET * NSTR)]
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(NSET)]      # the decodes share the weights
    b = torch.zeros(N, device="cuda")
    if kind in "cf":
        odt, oc = (torch.bfloat16, H.DT_BF16) if kind == "c" else (torch.float32, H.DT_F32)
        ys = [torch.empty(M, N, device="cuda", dtype=odt) for _ in range(NSET * NSTR)]
        def run(i, wi):
            return lib.bofi_linear(H.ptr(xs[i]), H.DT_BF16, K, H.ptr(ws[wi]), H.DT_BF16, H.ptr(b), None, N, H.ptr(ys[i]), oc, N, M, N, K, 0, None, 0, H.stream_ptr())
    else:
        rs = [torch.randn(M, N, device="cuda") for _ in range(NSET * NSTR)]
        y2 = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NSET * NSTR)]
        def run(i, wi):
            return lib.bofi_linear_ex(H.ptr(xs[i]), H.DT_BF16, K, H.ptr(ws[wi]), H.DT_BF16, H.ptr(b), H.ptr(rs[i]), N, H.ptr(rs[i]), H.DT_F32, N, M, N, K, 0,
                                      None, 0, 0.0, 0, None, H.ptr(y2[i]), N, H.stream_ptr())
    res = []
    streams = [torch.cuda.Stream() for _ in range(NSTR)]
    for cfg in [None] + CONFIGS:
        if cfg is None:
            os.environ.pop("BOFI_GEMM_TILE", None)
        else:
            os.environ["BOFI_GEMM_TILE"] = cfg
        graphs, ok = [], True
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                if run(si * NSET, 0) != 0:
                    ok = False
                    break
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for j in range(NL):
                        run(si * NSET + j % NSET, j % NSET)
                graphs.append(gr)
        if not ok:
            continue
        def replay_all(n):
            for _ in range(n):
                for st, gr in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        gr.replay()
        replay_all(2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_event(e0)
        replay_all(10)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) * 1e3 / 10 / NL / NSTR, cfg or "heuristic"))
    base = [r for r in res if r[1] == "heuristic"][0][0]
    print(f"{sh}: heuristic {base:.2f} us/GEMM aggregate ({2.0*M*N*K/base/1e6:.0f} TF) | " + ", ".join(f"{c} {u:.2f}" for u, c in sorted(res)), flush=True)
