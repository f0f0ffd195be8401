"""Gate for VERDICT r5 item 1 ("take W_o out of the attention workgroups"): before any kernel is written, what would the two pieces cost?
  (a) the attention CORE as a light kernel (context rows to memory as bf16, no W_o, no residual): the tiled family's attn_bf16_kernel IS that kernel
      (register-resident S^T = K Q^T, softmax, O^T = V^T P^T, wavefront = (image, head)) -- timed against the fused sublayer kernel (bofi_attn_block);
  (b) W_o as a head segment of the feed-forward kernel: 0.5 MB more weight stream per 80-row block -- proxy: bofi_ffn_block at d_ff 2560 against 2048
      (+1 MB of weights per block = two such segments: half the difference).
Each alone on the chip and as four concurrent streams (graphs of 12 launches, as four decodes in flight).   python dev/exp/gate_attn_split.py"""
import math, sys
import torch
sys.path.insert(0, ".")
from boficap_amd import hip as H

H.lib()
d, dev = 512, "cuda"


def pack(w):
    N, K = w.shape
    out = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().bofi_pack_frag(H.ptr(w), H.ptr(out), N, K, H.stream_ptr()))
    return out


def streams_time(make_run, nstr, nl=12, reps=10):
    streams = [torch.cuda.Stream() for _ in range(nstr)]
    graphs = []
    runs = [make_run(si) for si in range(nstr)]                   # (kept alive until the timing is over: a captured graph holds no reference to the tensors its launches
    torch.cuda.synchronize()                                      # use, and torch.cuda.graph() empties the allocator's cache when a capture starts -- freed operands would be unmapped)
    for si, st in enumerate(streams):
        run = runs[si]
        with torch.cuda.stream(st):
            run(0); run(1); torch.cuda.synchronize()
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
    replay_all(reps)
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / reps / nl / nstr
    del graphs, runs
    return t


def attn_case(B, Lq, Lk, cross, nstr):
    M = B * Lq
    wop = pack((torch.randn(d, d, device=dev) / math.sqrt(d)).to(torch.bfloat16))
    bo = torch.randn(d, device=dev)
    klen = torch.full((B,), Lk, dtype=torch.int32, device=dev)
    res = {}
    for form in ("fused", "core"):
        def make_run(si):
            rot = 2
            if cross:
                qs = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(rot)]
                kvs = [torch.randn(B * Lk, 7168, device=dev).to(torch.bfloat16) for _ in range(rot)]
            else:
                qkvs = [torch.randn(M, 3 * d, device=dev).to(torch.bfloat16) for _ in range(rot)]
            xs = [torch.randn(M, d, device=dev) for _ in range(rot)]
            ctx = [torch.empty(M, d, dtype=torch.bfloat16, device=dev) for _ in range(rot)]

            def run(j):
                i = j % rot
                if cross:
                    q, k, v, ldq, ldk = qs[i], kvs[i][:, 1024:], kvs[i][:, 1536:], d, 7168
                else:
                    q, k, v, ldq, ldk = qkvs[i], qkvs[i][:, d:], qkvs[i][:, 2 * d:], 3 * d, 3 * d
                if form == "fused":
                    H.check(H.lib().bofi_attn_block(H.ptr(q), ldq, H.ptr(k), ldk, H.ptr(v), ldk, B, Lq, Lk, H.ptr(klen), 1, 0, 0, 0, H.ptr(wop), H.ptr(bo),
                                                    H.ptr(xs[i]), d, H.ptr(xs[i]), d, None, None, H.stream_ptr()))
                else:
                    H.check(H.lib().bofi_attention(H.ptr(q), ldq, H.ptr(k), ldk, H.ptr(v), ldk, H.ptr(ctx[i]), d, 1, B, 8, Lq, Lk, H.ptr(klen), 1, 0, H.stream_ptr()))
            return run
        res[form] = streams_time(make_run, nstr)
    print(f"attention B {B} Lq {Lq} Lk {Lk} {'cross' if cross else 'self '} x {nstr} stream(s): fused sublayer {res['fused']:6.2f} us | core only {res['core']:6.2f} us  (per launch, aggregate)", flush=True)
    return res


def ffn_case(M, nstr):
    res = {}
    for dff in (2048, 2560):
        ws = []
        for _ in range(2):
            w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
            w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
            ws.append((pack(w1), torch.randn(dff, device=dev), w1.float().sum(1), pack(w2), torch.randn(d, device=dev)))

        def make_run(si):
            xs = [torch.randn(M, d, device=dev) for _ in range(2)]

            def run(j):
                w1p, c1, cs1, w2p, b2 = ws[j % 2]
                x = xs[j % 2]
                H.check(H.lib().bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff, H.stream_ptr()))
            return run
        res[dff] = streams_time(make_run, nstr)
    print(f"feed-forward M {M} x {nstr} stream(s): d_ff 2048 {res[2048]:6.2f} us | d_ff 2560 {res[2560]:6.2f} us -> a 0.5 MB head segment ~ {(res[2560] - res[2048]) / 2:5.2f} us", flush=True)
    return res


def head_case(M, nstr, N=1536):
    """The BUILT head segment (rb_ffn5_kernel<.., HEAD>, round 6): feed-forward + projection tail without / with W_o + residual in front (bofi_ffn_linear_block against
    bofi_attn_out_ffn_block), same streams."""
    res = {}
    dff = 2048
    ws = []
    for _ in range(2):
        w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).to(torch.bfloat16)
        wj = (torch.randn(N, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        wo = (torch.randn(d, d, device=dev) / math.sqrt(d)).to(torch.bfloat16)
        ws.append((pack(w1), torch.randn(dff, device=dev), w1.float().sum(1), pack(w2), torch.randn(d, device=dev), pack(wj), torch.randn(N, device=dev), wj.float().sum(1), pack(wo), torch.randn(d, device=dev)))
    for form in ("plain", "head"):
        def make_run(si):
            xs = [torch.randn(M, d, device=dev) for _ in range(2)]
            cs = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(2)]
            ps = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2)]

            def run(j):
                w1p, c1, cs1, w2p, b2, wjp, cj, csj, wop, bo = ws[j % 2]
                x, c, pj = xs[j % 2], cs[j % 2], ps[j % 2]
                if form == "plain":
                    H.check(H.lib().bofi_ffn_linear_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, M, dff, H.ptr(wjp), H.ptr(cj), H.ptr(csj),
                                                          H.ptr(pj), N, N, H.stream_ptr()))
                else:
                    H.check(H.lib().bofi_attn_out_ffn_block(H.ptr(x), d, H.ptr(c), d, H.ptr(wop), H.ptr(bo), H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, M, dff,
                                                            H.ptr(wjp), H.ptr(cj), H.ptr(csj), H.ptr(pj), N, N, H.stream_ptr()))
            return run
        res[form] = streams_time(make_run, nstr)
    print(f"feed-forward + q|k|v tail M {M} x {nstr} stream(s): plain {res['plain']:6.2f} us | with the W_o head segment {res['head']:6.2f} us  (+{res['head'] - res['plain']:5.2f})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "head":
        for nstr in (1, 4):
            head_case(320 * 36, nstr)
            head_case(320 * 20, nstr)
        sys.exit(0)
    B = 320
    tot = {}
    for nstr in (1, 4):
        a_enc = attn_case(B, 36, 36, False, nstr)
        a_self = attn_case(B, 20, 20, False, nstr)
        a_cross = attn_case(B, 20, 36, True, nstr)
        f_enc = ffn_case(B * 36, nstr)
        f_fill = ffn_case(B * 20, nstr)
        wo_enc, wo_fill = (f_enc[2560] - f_enc[2048]) / 2, (f_fill[2560] - f_fill[2048]) / 2
        today = 6 * (a_enc["fused"] + a_self["fused"] + a_cross["fused"])
        split = 6 * (a_enc["core"] + wo_enc + a_self["core"] + a_cross["core"] + 2 * wo_fill)      # (the self-attention's W_o priced like a head segment too: its chained launch is not cheaper)
        print(f"== {nstr} stream(s): attention sublayers of a decode today {today:7.1f} us; split form (cores + W_o as head segments) {split:7.1f} us", flush=True)
