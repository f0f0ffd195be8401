"""Time the row-block sublayer kernels alone on the chip (BOFI_RB_ATTN_W=16|8: attention workgroup shape) (in-graph time per launch, buffers rotated through the Infinity Cache).
python dev/exp/mb_rowblock.py [ffn|attn|gemm|train]   (BOFI_RB_FFN_V / BOFI_RB_FFN_BPW / BOFI_RB_GEMM_BPW select the kernel forms)"""
import math, os, sys
import torch
sys.path.insert(0, ".")
from boficap_amd import hip as H

H.lib()
d, dff = 512, 2048
dev = "cuda"


def timed(fn, iters=50, rot=1):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i % rot)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                fn(i % rot)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def pack(w):
    N, K = w.shape
    out = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().bofi_pack_frag(H.ptr(w), H.ptr(out), N, K, H.stream_ptr()))
    return out


def ffn(M):
    rot = 4
    xs = [torch.randn(M, d, device=dev) for _ in range(rot)]
    ys = [torch.empty(M, d, device=dev) for _ in range(rot)]
    ybs = [torch.empty(M, d, dtype=torch.bfloat16, device=dev) for _ in range(rot)]
    sts = [torch.empty(M, 16, 2, device=dev) for _ in range(rot)]
    w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
    w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
    w1p, w2p = pack(w1), pack(w2)
    c1, cs1, b2 = torch.randn(dff, device=dev), w1.float().sum(1), torch.randn(d, device=dev)

    import os
    extra = os.environ.get("MB_EXTRA", "0") == "1"                  # the optional bf16 copy + partial sums (the engine writes neither at 320 images)

    def run(i):
        H.check(H.lib().bofi_ffn_block(H.ptr(xs[i]), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(ys[i]), d, H.ptr(ybs[i]) if extra else None,
                                       H.ptr(sts[i]) if extra else None, M, dff, H.stream_ptr()))
    t = timed(run, rot=rot)
    print(f"ffn_block M {M:6d}: {t:7.2f} us  {4.0 * M * d * dff / t * 1e-6:7.1f} TFLOP/s  weight stream {4 * 1048576 / t * 1e-3:6.1f} GB/s per workgroup", flush=True)


def ffn_streams(M, nstr=4, nl=12):
    """nstr streams each replay a graph of nl feed-forward launches (two weight sets alternating, own activations) at once, as decodes in
    flight do: aggregate time per launch."""
    xs = [[torch.randn(M, d, device=dev) for _ in range(2)] for _ in range(nstr)]
    ws = []
    for _ in range(2):
        w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
        ws.append((pack(w1), torch.randn(dff, device=dev), w1.float().sum(1), pack(w2), torch.randn(d, device=dev)))
    streams = [torch.cuda.Stream() for _ in range(nstr)]
    graphs = []
    for si, st in enumerate(streams):
        def run(j):
            w1p, c1, cs1, w2p, b2 = ws[j % 2]
            x = xs[si][j % 2]
            H.check(H.lib().bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
        with torch.cuda.stream(st):
            run(0); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for j in range(nl):
                    run(j)
            graphs.append(g)
    def replay_all(n):
        for _ in range(n):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
    replay_all(2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in streams:
        st.wait_event(e0)
    replay_all(10)
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 10 / nl / nstr
    clk = ""
    if os.environ.get("BOFI_RB_DBG") == "16":                      # the stamped kernel: shader clock of workgroup 0's last launch (s_memtime against the 100 MHz s_memrealtime)
        import ctypes as C
        buf = (C.c_ulonglong * 256)()
        H.lib().bofi_rb_stamps.restype = C.c_int; H.lib().bofi_rb_stamps.argtypes = [C.c_void_p]
        H.check(H.lib().bofi_rb_stamps(buf))
        for base in (0, 128):
            if buf[base + 126]:
                clk += f"  [{'producer' if base == 0 else 'consumer'} wavefront: {(buf[base + 127] - buf[base]) / (buf[base + 126] * 10.0):.3f} GHz over {buf[base + 126] / 100.0:.1f} us]"
    print(f"ffn_block M {M:6d} x {nstr} streams: {t:7.2f} us per launch aggregate  {4.0 * M * d * dff / t * 1e-6:7.1f} TFLOP/s{clk}", flush=True)


def attn(B, Lq, Lk, cross):
    rot = 4
    M = B * Lq
    if cross:
        qs = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(rot)]
        kvs = [torch.randn(B * Lk, 7168, device=dev).to(torch.bfloat16) for _ in range(rot)]
    else:
        qkvs = [torch.randn(M, 3 * d, device=dev).to(torch.bfloat16) for _ in range(rot)]
    xs = [torch.randn(M, d, device=dev) for _ in range(rot)]
    ybs = [torch.empty(M, d, dtype=torch.bfloat16, device=dev) for _ in range(rot)]
    ys = [torch.empty(M, d, device=dev) for _ in range(rot)]
    sts = [torch.empty(M, 16, 2, device=dev) for _ in range(rot)]
    wop = pack((torch.randn(d, d, device=dev) / math.sqrt(d)).to(torch.bfloat16))
    bo = torch.randn(d, device=dev)
    klen = torch.full((B,), Lk, dtype=torch.int32, device=dev)

    def run(i):
        if cross:
            q, k, v, ldq, ldk = qs[i], kvs[i][:, 1024:], kvs[i][:, 1536:], d, 7168
        else:
            q, k, v, ldq, ldk = qkvs[i], qkvs[i][:, d:], qkvs[i][:, 2 * d:], 3 * d, 3 * d
        import os
        opt = int(os.environ.get("MB_OPT", "0"))
        H.check(H.lib().bofi_attn_block(H.ptr(q), ldq, H.ptr(k), ldk, H.ptr(v), ldk, B, Lq, Lk, H.ptr(klen), 1, 0, 0, 0, H.ptr(wop), H.ptr(bo),
                                        H.ptr(xs[i]), d, H.ptr(ys[i] if opt & 1 else xs[i]), d, None if opt & 2 else H.ptr(ybs[i]), None if opt & 4 else H.ptr(sts[i]), H.stream_ptr()))
    t = timed(run, rot=rot)
    print(f"attn_block B {B:4d} Lq {Lq} Lk {Lk} {'cross' if cross else 'self '}: {t:7.2f} us", flush=True)


def gemm(M, N, f32out):
    rot = 4
    xs = [torch.randn(M, d, device=dev) for _ in range(rot)]
    ys = [torch.empty(M, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev) for _ in range(rot)]
    w = (torch.randn(N, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
    wp, c, cs = pack(w), torch.randn(N, device=dev), w.float().sum(1)

    def run(i):
        H.check(H.lib().bofi_linear_block(H.ptr(xs[i]), d, H.ptr(wp), H.ptr(c), H.ptr(cs), H.ptr(ys[i]), N, 1 if f32out else 0, M, N, 0, H.stream_ptr()))
    t = timed(run, rot=rot)
    print(f"linear_block M {M:6d} N {N:5d} {'f32 ' if f32out else 'bf16'}: {t:7.2f} us  {2.0 * M * d * N / t * 1e-6:7.1f} TFLOP/s", flush=True)


def train_shapes():
    """The XE step's forward / dX shapes (K = 512): the tiled GEMM (bf16 operand in, float32 out) against the row-block projection
    kernel (float32 stream in with the LayerNorm fold, float32 / bf16 out)."""
    rot = 4
    for M in (5120, 2304):
        for N in (512, 1536, 2048):
            xs = [torch.randn(M, d, device=dev) for _ in range(rot)]
            xbs = [x.bfloat16() for x in xs]
            w = (torch.randn(N, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
            wp, c, cs = pack(w), torch.randn(N, device=dev), w.float().sum(1)
            y32 = [torch.empty(M, N, device=dev) for _ in range(rot)]
            y16 = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(rot)]
            t_tiled = timed(lambda i: H.check(H.lib().bofi_linear(H.ptr(xbs[i]), H.dtype_code(xbs[i]), d, H.ptr(w), H.dtype_code(w), H.ptr(c), None, N, H.ptr(y32[i]), 0, N,
                                                                  M, N, d, 0, None, 0, H.stream_ptr())), rot=rot)
            t_rb32 = timed(lambda i: H.check(H.lib().bofi_linear_block(H.ptr(xs[i]), d, H.ptr(wp), H.ptr(c), H.ptr(cs), H.ptr(y32[i]), N, 1, M, N, 0, H.stream_ptr())), rot=rot)
            t_rb16 = timed(lambda i: H.check(H.lib().bofi_linear_block(H.ptr(xs[i]), d, H.ptr(wp), H.ptr(c), H.ptr(cs), H.ptr(y16[i]), N, 0, M, N, 0, H.stream_ptr())), rot=rot)
            print(f"M {M:5d} N {N:5d} K 512: tiled bf16->f32 {t_tiled:6.2f} us | row-block f32->f32 {t_rb32:6.2f} us | row-block f32->bf16 {t_rb16:6.2f} us", flush=True)


def ffn_proj(M, N, nstr=1, nl=12):
    """The feed-forward sublayer + the projection that reads its output: two launches (ffn_block, linear_block) against the one fused launch
    (ffn_linear_block); nstr streams replaying graphs of nl sublayers at once (1 = alone)."""
    res = {}
    ws = []
    for _ in range(2):
        w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
        wj = (torch.randn(N, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        ws.append((pack(w1), torch.randn(dff, device=dev), w1.float().sum(1), pack(w2), torch.randn(d, device=dev), pack(wj), torch.randn(N, device=dev), wj.float().sum(1)))
    xs = [[torch.randn(M, d, device=dev) for _ in range(2)] for _ in range(nstr)]
    ps = [[torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2)] for _ in range(nstr)]
    streams = [torch.cuda.Stream() for _ in range(nstr)]
    for fused in (0, 1):
        graphs = []
        for si, st in enumerate(streams):
            def run(j):
                w1p, c1, cs1, w2p, b2, wjp, cj, csj = ws[j % 2]
                x, pj = xs[si][j % 2], ps[si][j % 2]
                if fused:
                    H.check(H.lib().bofi_ffn_linear_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, M, dff,
                                                          H.ptr(wjp), H.ptr(cj), H.ptr(csj), H.ptr(pj), N, N, H.stream_ptr()))
                else:
                    H.check(H.lib().bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
                    H.check(H.lib().bofi_linear_block(H.ptr(x), d, H.ptr(wjp), H.ptr(cj), H.ptr(csj), H.ptr(pj), N, 0, M, N, 0, H.stream_ptr()))
            with torch.cuda.stream(st):
                run(0); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for j in range(nl):
                        run(j)
                graphs.append(g)
        def replay_all(n):
            for _ in range(n):
                for st, g in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        g.replay()
        replay_all(2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_event(e0)
        replay_all(10)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record(); torch.cuda.synchronize()
        res[fused] = e0.elapsed_time(e1) * 1e3 / 10 / nl / nstr
    fl = (4.0 * M * d * dff + 2.0 * M * d * N)
    print(f"ffn + projection M {M:6d} N {N:5d} x {nstr} streams: two launches {res[0]:7.2f} us ({fl / res[0] * 1e-6:6.1f} TFLOP/s) | one launch {res[1]:7.2f} us "
          f"({fl / res[1] * 1e-6:6.1f} TFLOP/s)  per sublayer, aggregate", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "ffn"
    if what == "ffnproj":
        for M, N in ((11520, 1536), (11520, 7168), (6400, 1536)):
            for n in (1, 2, 4):
                ffn_proj(M, N, n)
    elif what == "ffn":
        for M in (11520, 6400, 2304, 1280, 64):
            ffn(M)
    elif what == "ffn4":
        for M in (11520, 6400):
            for n in (1, 2, 4):
                ffn_streams(M, n)
    elif what == "train":
        train_shapes()
    elif what == "gemm":
        for M, N, f in ((11520, 1536, False), (6400, 1536, False), (6400, 512, False), (11520, 7168, False), (6400, 9600, True), (64, 1536, False), (64, 9600, True)):
            gemm(M, N, f)
    else:
        for B in (320, 64, 2):
            attn(B, 36, 36, False)
            attn(B, 20, 20, False)
            attn(B, 20, 36, True)
