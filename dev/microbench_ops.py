#!/usr/bin/env python3
"""Per-shape timing of the C-ABI operators on the MI355X (developer tool, HIP-event timing).

    python dev/microbench_ops.py [--dtype bf16|f32] [--iters 200]
Prints one line per (op, shape): average microseconds per launch and achieved TFLOP/s / GB/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from boficap_amd import hip as H  # noqa: E402


def timeit(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters       # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lib = H.lib()
    st = H.stream_ptr()
    shapes = [("att_embed", 2304, 512, 2048), ("enc_qkv", 2304, 1536, 512), ("enc_o", 2304, 512, 512), ("enc_ffn1", 2304, 2048, 512),
              ("enc_ffn2", 2304, 512, 2048), ("kv_all", 2304, 7168, 512), ("fill_qkv", 1280, 1536, 512), ("fill_o", 1280, 512, 512),
              ("fill_ffn1", 1280, 2048, 512), ("fill_ffn2", 1280, 512, 2048), ("vocab", 1280, 9491, 512),
              ("bound_o", 64, 512, 512), ("bound_ffn1", 64, 2048, 512), ("bound_ffn2", 64, 512, 2048)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(tdt)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(tdt)
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        y32 = torch.empty(M, N, device="cuda")
        yt = torch.empty(M, N, device="cuda", dtype=tdt)

        def run(y, res):
            H.check(lib.bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), H.ptr(res), N, H.ptr(y),
                                    H.dtype_code(y), N, M, N, K, 0, None, 0, st))
        for label, y, res in (("out=T", yt, None), ("out=f32+res", y32, r)):
            us = timeit(lambda: run(y, res), args.iters)
            print(f"linear {name:10s} M={M:5d} N={N:5d} K={K:5d} {label:12s} {us:8.2f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
    # layernorm
    for rows in (2304, 1280, 64):
        x = torch.randn(rows, 512, device="cuda")
        g, b = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
        y = torch.empty(rows, 512, device="cuda", dtype=tdt)
        us = timeit(lambda: H.check(lib.bofi_layernorm(H.ptr(x), H.ptr(g), H.ptr(b), H.ptr(y), H.dtype_code(y), rows, 512, st)), args.iters)
        print(f"layernorm rows={rows:5d} d=512 {us:8.2f} us  {rows * 512 * (4 + y.element_size()) / us / 1e3:8.1f} GB/s", flush=True)
    # attention
    for name, B, h, Lq, Lk in (("enc_self", 64, 8, 36, 36), ("fill_self", 64, 8, 20, 20), ("fill_cross", 64, 8, 20, 36), ("bound_cross", 64, 8, 1, 36)):
        d = h * 64
        q = torch.randn(B, Lq, d, device="cuda").to(tdt)
        k = torch.randn(B, Lk, d, device="cuda").to(tdt)
        v = torch.randn(B, Lk, d, device="cuda").to(tdt)
        o = torch.empty(B, Lq, d, device="cuda", dtype=tdt)
        us = timeit(lambda: H.check(lib.bofi_attention(H.ptr(q), d, H.ptr(k), d, H.ptr(v), d, H.ptr(o), d, H.dtype_code(o), B, h, Lq, Lk, None, 0, 0, st)), args.iters)
        print(f"attention {name:11s} B={B} h={h} Lq={Lq} Lk={Lk} {us:8.2f} us", flush=True)
    # vocab finalize
    lg = torch.randn(1280, 9491, device="cuda")
    seq = torch.empty(1280, dtype=torch.int64, device="cuda")
    us = timeit(lambda: H.check(lib.bofi_vocab_finalize(H.ptr(lg), 1280, 9491, 20, 1, None, 0, H.ptr(seq), st)), args.iters)
    print(f"vocab_finalize rows=1280 V=9491 {us:8.2f} us  {1280 * 9491 * 4 * 3 / us / 1e3:8.1f} GB/s (2 reads + 1 write)", flush=True)


if __name__ == "__main__":
    main()
