#!/usr/bin/env python3
"""Benchmark of the NAIC bound+fill decode hot path on MI355X (BASELINE.json metric).

A step = one greedy bound+fill decode (encoder -> bounding loop -> filling pass -> vocabulary
projection / log-softmax / argmax) of one batch of 64 images x 36 regions x 2048 features, bf16,
features resident in HBM, ids + slot layout + the [B,20,V] log-prob tensor written to HBM exactly as
the reference's ``_sample`` returns them.  One process per GPU; images shard across ranks with no
collective (weak scaling: every rank decodes its own 64-image batch per step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--dtype bf16|f32]
                    [--ids-only] [--no-graph] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task brief): metric/value/unit, roofline (MFMA
bound, algorithmic FLOPs of SURVEY.md §8d over the HIP-event time of the decode launches) and
cpu_baseline (the CPU oracle timed on this host's cores, bounded sample).  At N = 1 the line also
carries ``secondary``: short driver-run measurements of the other BASELINE configurations -- XE step
64 x 5 bf16 (config 3), self-critical step 10 x 5 (config 4), batch 256 with 3 refinement rounds
(config 5) -- each with a roofline on EXECUTED GEMM FLOPs and a cpu_baseline.

``--gpus N`` without a torchrun environment starts its own N ranks (children spawned before this
process touches the GPU) and relays rank 0's line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ATT_SEED = 1235      # seed 1234 puts an image with zero phrases last: quirk Q1 then NaNs the whole batch


def self_launch(argv, n: int) -> int:
    """``python bench.py --gpus N`` outside torchrun: start N ranks as children (torch.distributed.run, one process per GPU) and
    relay their output.  Called before anything in this process touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def dist_setup(args):
    """(rank, local_rank, world, device); initialises RCCL when launched with several ranks."""
    rank, local_rank, world = (int(os.environ.get(k, "0")) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))
    world = max(world, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BOFI_BENCH_REHEARSAL=1 (tests, one-GPU boxes): every rank on device 0 over gloo -- RCCL refuses two ranks on one device.  It walks the
    # whole N > 1 path of this script (self-launch, sharding, barriers, the max over ranks, the XE exchange secondary); its numbers mean nothing
    # and the line says so (config.rehearsal)
    rehearsal = os.environ.get("BOFI_BENCH_REHEARSAL") == "1"
    if os.environ.get("BOFI_BENCH_REHEARSAL") == "dry":      # the launcher / sharding / barrier / reduction plumbing of N ranks WITHOUT a GPU (run_dry): gloo on the CPU
        if world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo")
        return rank, local_rank, world, torch.device("cpu")
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)    # RCCL
    return rank, local_rank, world, dev
MFMA_PEAK = {"bf16": 2500.0, "f32": 157.3}      # dense TFLOP/s, MI355X_MICROARCH.md


def f_alg(T: float, cfg) -> float:
    """Algorithmic FLOPs per image for T bound iterations, from the config."""
    d, dff, F, R, S, L, V = cfg.d_model, cfg.d_ff, cfg.att_feat_size, 36, cfg.seq_length, cfg.seq_length + 2, cfg.tgt_vocab
    att_embed = 2 * R * F * d
    enc_layer = 2 * R * (4 * d * d + 2 * d * dff) + 4 * R * R * d
    dec_layer = 2 * S * (4 * d * d + 2 * d * d + 2 * d * dff) + 2 * R * 2 * d * d + 4 * S * S * d + 4 * S * R * d
    vocab = 2 * S * d * V
    bound_once = 2 * (L + R) * 2 * d * d                      # K/V of the 22 slot rows and the 36 memory rows
    # row-0 work per iteration as SURVEY.md §8d prices it (4 d^2 projections; the build precomputes
    # the constant row-0 query, so it executes slightly less than it is charged for)
    bound_iter = 2 * (4 * d * d + 2 * d * dff + 2 * 100 * d + 100 * 30) + 4 * (L + R) * d
    return att_embed + cfg.N_enc * enc_layer + cfg.N_dec * dec_layer + vocab + bound_once + T * bound_iter


# CPU-oracle legs are put off until every GPU leg of the process is through (main runs them): the oracle's 16 OpenMP threads keep spinning for a
# while after a parallel region, and a GPU leg that starts behind one loses host time to them (the self-critical step's scorer / collate and the
# launch loops measured 10-25 % slower right behind an oracle leg).  (result dict, closure) pairs.
_DEFERRED_CPU: list = []


def _cpu_leg(res, fn):
    _DEFERRED_CPU.append((res, fn))


def cpu_baseline(cfg, sd, batch, seed, budget_s=20.0):
    """The CPU oracle (kind 'port': a restatement pinned to the reference by tests/golden) on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import boficap_oracle as O
    from boficap_amd import weights as W
    # the GPU box gives one GPU's share of the host (16 cores), whatever os.cpu_count() says
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("BOFI_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    w = O.as_torch(sd)
    att = torch.from_numpy(W.synthetic_att_feats(batch, 36, cfg.att_feat_size, seed=seed))
    O.sample_naic(w, cfg, att)                                 # warm-up
    times, t_end = [], time.time() + budget_s
    while len(times) < 5 and (time.time() < t_end or not times):
        t0 = time.time()
        O.sample_naic(w, cfg, att)
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{len(times)} x one batch of {batch} images, fp32 torch-CPU oracle, median, {torch.get_num_threads()} threads"}


def gemm_rooflines(dtype, dev, batches=1):
    """MFMA roofline of the GEMM-shaped kernels THE ENGINE RUNS at the measured launch size, on the path's largest shapes: from 4 096 rows
    on (bf16) the row-block sublayer kernels -- bofi_linear_block (LayerNorm-folded projection reading the float32 stream) and
    bofi_ffn_block (the whole feed-forward sublayer) --, below that the tiled / persistent GEMM (bofi_linear).  100 dependent launches
    of one shape captured in a graph, HIP events around 20 replays on the launch stream (in-graph time per launch, launch boundary
    included)."""
    from boficap_amd import hip as H
    lib = H.lib()
    out = []
    peak = MFMA_PEAK["bf16" if dtype == torch.bfloat16 else "f32"]
    rb_min = int(os.environ.get("BOFI_RB_MIN_ROWS", "4096"))

    def timed(run):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            run(); torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(100):
                    run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                g.replay()
            e1.record(st)
            torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / 2000

    def pack(w):
        o = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
        H.check(lib.bofi_pack_frag(H.ptr(w), H.ptr(o), w.shape[0], w.shape[1], H.stream_ptr()))
        return o

    def entry(kernel, shape, us, flops):
        tf = flops / us / 1e6
        out.append({"kernel": kernel, "shape": shape, "us_per_launch": round(us, 2), "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)})

    m_enc, m_fill = 2304 * batches, 1280 * batches
    row_block = dtype == torch.bfloat16 and m_fill >= rb_min
    if row_block:
        d, dff = 512, 2048
        for name, M, N, f32out in (("encoder q|k|v, first layer", m_enc, 1536, False), ("filling-pass q|k|v, first layer", m_fill, 1536, False),
                                   ("generator.proj (weight rows zero-padded 9491 -> 9600, float32 logits)", m_fill, 9600, True)):
            x = torch.randn(M, d, device=dev)
            w = (torch.randn(N, d, device=dev) / d ** 0.5).to(torch.bfloat16)
            wp, c, cs = pack(w), torch.zeros(N, device=dev), w.float().sum(1)
            y = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
            us = timed(lambda: H.check(lib.bofi_linear_block(H.ptr(x), d, H.ptr(wp), H.ptr(c), H.ptr(cs), H.ptr(y), N, 1 if f32out else 0, M, N, 0, H.stream_ptr())))
            n_alg = 9491 if f32out else N
            entry("rb_gemm_kernel (row-block projection, LayerNorm folded, float32 stream in)", f"{name}: M={M} N={n_alg} K=512", us, 2.0 * M * n_alg * d)
        # the feed-forward sublayers: with launches in flight (the headline) each but the filling pass's last also computes the projection that reads
        # its output next (the next layer's q|k|v; after the last encoder layer the stacked cross K|V) from the closed block -- one launch
        for name, M, N in (("encoder feed-forward sublayer + next layer's q|k|v", m_enc, 1536), ("last encoder feed-forward sublayer + cross K|V of all layers", m_enc, 7168),
                           ("filling-pass feed-forward sublayer + next layer's q|k|v", m_fill, 1536), ("filling-pass feed-forward sublayer, last layer", m_fill, 0)):
            x = torch.randn(M, d, device=dev)
            w1 = (torch.randn(dff, d, device=dev) / d ** 0.5).to(torch.bfloat16)
            w2 = (torch.randn(d, dff, device=dev) / dff ** 0.5).to(torch.bfloat16)
            w1p, w2p, c1, cs1, b2 = pack(w1), pack(w2), torch.zeros(dff, device=dev), w1.float().sum(1), torch.zeros(d, device=dev)
            if N:
                wj = (torch.randn(N, d, device=dev) / d ** 0.5).to(torch.bfloat16)
                wjp, cj, csj, pj = pack(wj), torch.zeros(N, device=dev), wj.float().sum(1), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                us = timed(lambda: H.check(lib.bofi_ffn_linear_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, M, dff,
                                                                      H.ptr(wjp), H.ptr(cj), H.ptr(csj), H.ptr(pj), N, N, H.stream_ptr())))
                entry("rb_ffn5_kernel<projection tail> (row-block feed-forward sublayer on 80-row blocks + the LayerNorm-folded projection of each closed block)",
                      f"{name}: M={M} d=512 d_ff=2048 N={N}", us, 4.0 * M * d * dff + 2.0 * M * d * N)
            else:
                us = timed(lambda: H.check(lib.bofi_ffn_block(H.ptr(x), d, H.ptr(w1p), H.ptr(c1), H.ptr(cs1), H.ptr(w2p), H.ptr(b2), H.ptr(x), d, None, None, M, dff,
                                                               H.stream_ptr())))
                entry("rb_ffn5_kernel (row-block feed-forward sublayer, 80-row blocks)", f"{name}: M={M} d=512 d_ff=2048", us, 4.0 * M * d * dff)
        return out
    for name, M, N, K in (("cross K|V of all layers (kv_all)", m_enc, 7168, 512), ("encoder FFN w_1", m_enc, 2048, 512),
                          ("encoder FFN w_2", m_enc, 512, 2048), ("generator.proj", m_fill, 9491, 512)):
        n_alg = N
        if N % 128 and dtype == torch.bfloat16:                # as the bf16 engine runs it: weight rows zero-padded to whole 128-column tiles
            N = (N + 127) // 128 * 128
            name += f" (weight rows zero-padded {n_alg} -> {N}, as the engine runs it)"
        x = torch.randn(M, K, device=dev).to(dtype)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dtype)
        b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=dtype)
        us = timed(lambda: H.check(lib.bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(b), None, N, H.ptr(y), H.dtype_code(y), N,
                                                   M, N, K, 0, None, 0, H.stream_ptr())))
        pers = dtype == torch.bfloat16 and N % 128 == 0 and K % 64 == 0 and ((M + 255) // 256) * (N // 128) >= 90 and os.environ.get("BOFI_GEMM_PERS", "1") != "0"
        entry("gemm_pers_kernel (persistent 256x128 tiles, loader wavefronts)" if pers else "gemm_glds_kernel (one 128x64 tile per workgroup)",
              f"{name}: M={M} N={n_alg} K={K}", us, 2.0 * M * n_alg * K)
    return out


def xe_traffic(args):
    """HBM bytes per XE step from the committed PMC passes -- for the configuration they were taken on only."""
    if args.batch != 64 or args.seq_per_img != 5 or args.dtype != "bf16":
        return None
    for name in ("r05_xe_hbm_traffic.json", "r04_xe_hbm_traffic.json", "r03_xe_hbm_traffic.json", "r02_xe_hbm_traffic.json", "r01_xe_hbm_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f).get("hbm_bytes_per_step")
    return None


def f_alg_xe(cfg, seq_per_img: int, passes: float) -> float:
    """Algorithmic forward FLOPs per IMAGE of the XE step as the reference computes it: the encoder on
    every caption copy, `passes` full bound-layer passes per branch, two decoder passes and two vocabulary projections
    per caption; the step is priced at 3x forward (forward + two backward GEMMs per forward GEMM)."""
    d, dff, F, R, S, L, V = cfg.d_model, cfg.d_ff, cfg.att_feat_size, 36, cfg.seq_length, cfg.seq_length + 2, cfg.tgt_vocab
    att_embed = 2 * R * F * d
    enc_layer = 2 * R * (4 * d * d + 2 * d * dff) + 4 * R * R * d
    dec_layer = 2 * S * (4 * d * d + 2 * d * d + 2 * d * dff) + 2 * R * 2 * d * d + 4 * S * S * d + 4 * S * R * d
    bound_pass = 2 * L * (4 * d * d + 2 * d * d + 2 * d * dff) + 2 * R * 2 * d * d + 4 * L * L * d + 4 * L * R * d
    vocab = 2 * S * d * V
    per_caption = cfg.N_enc * enc_layer + 2 * passes * bound_pass + 2 * cfg.N_dec * dec_layer + 2 * vocab
    return 3.0 * (att_embed + seq_per_img * per_caption)


def cpu_baseline_xe(cfg, sd, spi, budget_s=25.0):
    """The CPU oracle's XE forward + criterion + backward (torch autograd on the host cores) on a 2-image sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import boficap_oracle as O
    from boficap_amd import weights as W
    from boficap_amd.collate import synthetic_training_batch
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("BOFI_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    n_img = 2
    att = torch.from_numpy(W.synthetic_att_feats(n_img, 36, cfg.att_feat_size, seed=ATT_SEED))
    b = {k: torch.from_numpy(v) for k, v in synthetic_training_batch(cfg, n_img, spi, seed=0).items()}
    w = {k: torch.from_numpy(v).clone().requires_grad_(k != "model.pos_embed.pe") for k, v in sd.items()}

    def once():
        for t in w.values():
            t.grad = None
        outs = O.forward_uic(w, cfg, att, b["labels"], None, b["phrase_num"], b["phrase_length"], b["phrase_syn"],
                             b["extend_phrase_syn_seq"], b["extend_phrase_seq"], b["extend_phrase_seq_mask"])
        O.criterion_uic(outs, b["phrase_num"], b["phrase_length"], b["phrase_syn"], b["labels"])[0].backward()
    once()
    times, t_end = [], time.time() + budget_s
    while len(times) < 5 and (time.time() < t_end or not times):
        t0 = time.time()
        once()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(n_img / med, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{len(times)} x XE forward+backward of {n_img} images x {spi} captions, fp32 torch-CPU oracle (no optimiser step), median"}


def _barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def _executed(flops_fixed, flops_skippable, active_share, ms, dtype):
    """Roofline entries on the GEMM FLOPs the step really executes (the library's own tally of 2 M N K per launch; launches of
    loop iterations that return at once because every image is finished are weighted by the share of active iterations)."""
    f = flops_fixed + flops_skippable * active_share
    tf = f / (ms * 1e-3) / 1e12
    return {"executed_gemm_flops": f, "achieved_executed": round(tf, 2), "frac_executed": round(tf / MFMA_PEAK[dtype], 5)}


def run_xe(args, ctx, log, cpu=True):
    """BASELINE config 3: XE training step, batch 64 images x 5 captions per GPU, data-parallel with one flat-bucket
    RCCL all-reduce per step (weak scaling)."""
    from boficap_amd import dp, hip as H
    rank, local_rank, world, dev = ctx
    import captioning.models as models
    from boficap_amd import weights as W
    from boficap_amd.collate import synthetic_training_batch
    from boficap_amd.config import FULL as cfg
    from boficap_amd.trainer import XETrainer
    spi = args.seq_per_img
    log("xe: building model")
    sd = W.make_state_dict(cfg, seed=0)
    opt = cfg.to_opt()
    opt.seed = 42 + 1000003 * rank                              # dropout streams differ per rank, as DataParallel replicas' do
    if args.dtype == "bf16":
        opt.bofi_train_dtype = torch.bfloat16
    model = models.setup(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev).train()
    group = None
    tr = XETrainer(model, opt, group=group, graph=not args.no_graph, streams=bool(args.streams))
    host_batch = synthetic_training_batch(cfg, args.batch, spi, seed=100 + rank)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in host_batch.items()}
    batch["max_phrase_num"] = int(host_batch["phrase_num"].max())
    batch["max_tokens"] = int((host_batch["phrase_length"].sum(-1) - 1).max())
    batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(args.batch, 36, cfg.att_feat_size, seed=ATT_SEED + 10 * rank)).to(dev)
    batch["att_masks"] = None
    batch = tr.add_token_rows(batch, host_batch)               # the vocabulary projection runs over the real tokens' rows only

    # executed GEMM FLOPs of one step: the library's tally over one eager pass (a graph replay does not pass the launchers)
    H.gemm_flops(reset=True)
    tr._forward_backward_eager(batch)
    fl_fixed, fl_skip = H.gemm_flops(reset=True)
    log("xe: warm-up")
    loss = None
    for _ in range(args.warmup):
        loss, _ = tr.step(batch)
    _barrier(world)
    log("xe: timing")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        loss, _ = tr.step(batch)
    e1.record()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3       # time to ENQUEUE a step (Python + launch overhead)
    _barrier(world)
    elapsed = dp.reduce_scalar(time.perf_counter() - t0, "max", device=dev)
    dev_ms = e0.elapsed_time(e1) / args.steps
    if rank != 0:
        return None
    images = args.batch * world * args.steps
    passes = float(batch["max_phrase_num"])
    flops = f_alg_xe(cfg, spi, passes) * args.batch
    achieved = flops / (dev_ms * 1e-3) / 1e12
    ex = _executed(fl_fixed, fl_skip, 1.0, dev_ms, args.dtype)
    # roofline.achieved / frac of this line are the GEMM FLOPs this build EXECUTES over the step's time -- a kernel-quality figure.  The
    # reference-structured figure (the FLOPs the reference would spend on the same batch: encoder per caption copy, max(phrase_num) full bound passes per
    # branch, padded decoder rows) is kept beside it as achieved_reference_flops / frac_reference_flops: it prices the restructuring, not the kernels.
    roof = {"bound": "mfma", "achieved": ex["achieved_executed"], "peak": MFMA_PEAK[args.dtype], "unit": "TFLOP/s",
            "frac": ex["frac_executed"], "traffic": xe_traffic(args),
            "traffic_note": "HBM-side bytes per step, rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate "
                            "passes (the newest profiles/r0N_xe_hbm_traffic.json; batch 64 x 5 bf16); null for other configurations",
            "kernel": "whole XE step (" + ("one hipGraph launch + all-reduce + Adam kernel" if tr.graph else "eager launches") + ")",
            "flops_per_launch": ex["executed_gemm_flops"], "launch_ms": round(dev_ms, 3),
            "achieved_reference_flops": round(achieved, 2), "frac_reference_flops": round(achieved / MFMA_PEAK[args.dtype], 5), "reference_flops_per_launch": flops,
            "note": "achieved / frac: the GEMM FLOPs this build really launches (encode once per image, row-0 bound queries, unpadded rows; library tally, "
                    "forward + backward) / HIP-event time per step; *_reference_flops: the algorithmic FLOPs of the step AS THE REFERENCE COMPUTES IT (SURVEY.md 8d: "
                    "encoder per caption copy, max(phrase_num) full bound passes per branch, padded decoder rows, x3 for fwd+bwd) / the same time"}
    roof.update(ex)
    res = {
        "metric": "images/sec XE training step (forward + criterion + backward + all-reduce + clip + Adam)",
        "value": round(images / elapsed, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"XE training (configs/uic_sd.yml model) batch={args.batch} images x {spi} captions per GPU, 36x2048 regions, "
                               f"d_model=512 6 enc + 6 dec + 1 bound layer, {args.dtype}, dropout on",
                   "images_per_step_per_gpu": args.batch, "captions_per_image": spi, "vocab": cfg.tgt_vocab,
                   "hip_graph": tr.graph, "longest_caption": batch["max_tokens"],
                   # rows the decoder stack and the vocabulary projection run over, both branches together (list padded to a
                   # multiple of 256), against the reference's 2 x N x seq_length
                   "decoder_rows_computed": int(batch["pair_src"].numel()) if "pair_src" in batch else 2 * int(batch["token_rows"].numel()),
                   "decoder_rows_dense": 2 * args.batch * spi * cfg.seq_length, "branches_share_launches": "pair_src" in batch,
                   "final_loss": round(float(loss), 4), "parameters": tr.bucket.numel, "host_enqueue_ms_per_step": round(host_ms, 3),
                   "sharding": "images by rank; RCCL all-reduce over the live part of the flat fp32 gradient bucket per step"},
        "roofline": roof,
    }
    if cpu and world == 1:
        _cpu_leg(res, lambda: cpu_baseline_xe(cfg, sd, spi, budget_s=args.cpu_budget))
    return res


DP_PHASE = ["not started"]      # where the data-parallel secondary is (read by the watchdog of main(): a hang's last phase goes into the record)


def run_xe_dp(args, ctx, log):
    """--gpus N > 1: BASELINE config 3 with its REAL exchange step (the decode shards with no collective, so the scaling run would never
    exercise RCCL otherwise): XE 64 x 5 per rank, float32 ring all-reduce and the bf16 mesh-direct wire, each against the same step
    WITHOUT the exchange (forward + backward + local optimiser): the difference is the collective time the step does not hide."""
    from boficap_amd import dp
    rank, local_rank, world, dev = ctx
    import captioning.models as models
    from boficap_amd import weights as W
    from boficap_amd.collate import synthetic_training_batch
    from boficap_amd.config import FULL as cfg
    from boficap_amd.trainer import XETrainer
    sd = W.make_state_dict(cfg, seed=0)
    out = {"rccl_ranks": world, "batch_per_rank": 64, "captions_per_image": 5}
    for wire in (None, "bf16"):
        wname = "fp32 ring all-reduce" if wire is None else "bf16 mesh-direct wire"
        DP_PHASE[0] = f"{wname}: building the model and the trainer"
        opt = cfg.to_opt()
        opt.seed = 42 + 1000003 * rank
        opt.bofi_train_dtype = torch.bfloat16
        opt.bofi_dp_wire = wire
        model = models.setup(opt)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        model.to(dev).train()
        tr = XETrainer(model, opt, graph=True)
        hb = synthetic_training_batch(cfg, 64, 5, seed=100 + rank)
        batch = {k: torch.from_numpy(v).to(dev) for k, v in hb.items()}
        batch["max_phrase_num"] = int(hb["phrase_num"].max())
        batch["max_tokens"] = int((hb["phrase_length"].sum(-1) - 1).max())
        batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=ATT_SEED + 10 * rank)).to(dev)
        batch["att_masks"] = None
        batch = tr.add_token_rows(batch, hb)

        def agreed(what, fn):
            """run a phase WITHOUT collectives on every rank, then agree on its outcome: one rank's exception must not leave the others in the next collective"""
            err = None
            try:
                fn()
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
            if dp.reduce_scalar(1.0 if err else 0.0, "max", device=dev) > 0:
                raise RuntimeError(f"{what} failed on rank {rank}: {err}" if err else f"{what} failed on another rank")

        def timed(fn, steps=10, warm=3):
            for _ in range(warm):
                fn()
            _barrier(world)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            _barrier(world)
            return dp.reduce_scalar((time.perf_counter() - t0) / steps * 1e3, "max", device=dev)

        def local_step():
            tr.forward_backward(batch)
            tr.optimizer_step()
        DP_PHASE[0] = f"{wname}: a local step (graph capture, no exchange)"
        agreed("a local step (forward + backward + optimiser, no exchange)", local_step)      # graph capture and kernels work on every rank before any rank enters a collective
        DP_PHASE[0] = f"{wname}: timed steps WITH the gradient exchange (collectives)"
        dp_ms = timed(lambda: tr.step(batch))
        DP_PHASE[0] = f"{wname}: timed steps without the exchange"
        local_ms = timed(local_step)
        key = "fp32_ring_all_reduce" if wire is None else "bf16_mesh_direct"
        out[key] = {"step_ms": round(dp_ms, 3), "step_without_exchange_ms": round(local_ms, 3), "exposed_collective_ms": round(max(0.0, dp_ms - local_ms), 3),
                    "images_per_sec": round(64 * world / dp_ms * 1e3, 1), "exchanged_bytes_per_rank": tr.bucket.live_numel * (4 if wire is None else 2),
                    "chunks": tr.dp_chunks}
        log(f"xe dp ({key}): {dp_ms:.2f} ms per step, {local_ms:.2f} without the exchange")
        del tr, model
        torch.cuda.empty_cache()
    return out


def cpu_baseline_rl(cfg, sd, n_img, n, budget_s=15.0):
    """The CPU oracle's self-critical step on a 2-image sample: greedy SAIC decode of the n copies, greedy NAIC decode, the
    teacher-forced re-forward of both modes' captions with torch autograd, new_self_critical x 2, backward (no optimiser step)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch.nn.functional as F
    import boficap_oracle as O
    from boficap_amd import weights as W
    from boficap_amd.collate import phrase_collate
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("BOFI_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    m_img, S = 2, cfg.seq_length
    pool = torch.from_numpy(W.synthetic_att_feats(12, 36, cfg.att_feat_size, seed=ATT_SEED))
    w = {k: torch.from_numpy(v).clone().requires_grad_(k != "model.pos_embed.pe") for k, v in sd.items()}
    with torch.no_grad():
        pn = O.sample_naic(w, cfg, pool)[2]
    att = pool[pn > 0][:m_img]

    def collate(seq, pl, ps):
        N = seq.shape[0]
        labels = np.zeros((N, S + 2), np.int64)
        labels[:, 0] = cfg.bos_idx
        labels[:, 1:S + 1] = seq.numpy()
        plen = pl.numpy().astype(np.int64)
        return phrase_collate(labels, plen, np.where(plen > 0, ps.numpy(), 0), len_idx=cfg.len_idx)

    def once():
        for t in w.values():
            t.grad = None
        with torch.no_grad():
            s_seq, _, _, s_pl, s_ps, _ = O.sample_saic(w, cfg, att.repeat_interleave(n, 0))
            n_seq, _, _, n_pl, n_ps, _ = O.sample_naic(w, cfg, att)
            n_seq, n_pl, n_ps = (t.repeat_interleave(n, 0) for t in (n_seq, n_pl, n_ps))
        memory, src_mask = O.memory_of(w, cfg, att)
        memory, src_mask = memory.repeat_interleave(n, 0), src_mask.repeat_interleave(n, 0)
        cs, cn = collate(s_seq, s_pl, s_ps), collate(n_seq, n_pl, n_ps)
        lp_s = F.log_softmax(O.logit(w, O.decode_sa(w, cfg, memory, torch.from_numpy(cs["extend_phrase_seq"]),
                                                     torch.from_numpy(cs["extend_phrase_syn_seq"][:, 1:-1].copy()), src_mask,
                                                     torch.from_numpy(cs["extend_phrase_seq_mask"]))), dim=-1)
        last = cn["phrase_length"][:, 1:].sum(1) + 1
        syn_mask = torch.zeros(last.shape[0], S, S, dtype=torch.bool)
        syn_mask[:, :, :max(int(last[-1]) - 1, 1)] = True
        lp_n = F.log_softmax(O.logit(w, O.decode_na(w, cfg, memory, torch.from_numpy(cn["extend_phrase_syn_seq"][:, 1:-1].copy()), src_mask, syn_mask)), dim=-1)
        sc = torch.rand(m_img * n, generator=torch.Generator().manual_seed(0))
        (O.new_self_critical(lp_s, s_seq, sc, n)[0] + O.new_self_critical(lp_n, n_seq, sc, n)[0]).backward()
    once()
    times, t_end = [], time.time() + budget_s
    while len(times) < 3 and (time.time() < t_end or not times):
        t0 = time.time()
        once()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(m_img / med, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{len(times)} x self-critical step of {m_img} images x {n} captions per mode (greedy SAIC + NAIC decodes, re-forward with autograd, "
                      "new_self_critical, backward; no scorer, no optimiser step), fp32 torch-CPU oracle, median"}


def run_rl(args, ctx, log, cpu=True):
    """BASELINE config 4: self-critical step, 10 images x 5 sampled captions per GPU in SAIC and in NAIC mode, re-forward with the
    tape, new_self_critical, backward, all-reduce, Adam.  The caption scorer (CIDEr-D in the reference) is an external CPU
    package: a stand-in that scores token overlap with a fixed pseudo-reference runs on the host in its place, so the measured
    step contains the same device->host->device round trip."""
    from boficap_amd import dp, hip as H
    rank, local_rank, world, dev = ctx
    import captioning.models as models
    from boficap_amd import weights as W
    from boficap_amd.config import FULL as cfg
    from boficap_amd.trainer import XETrainer
    n_img, n = args.batch, args.seq_per_img
    sd = W.with_len_row_shared(W.make_state_dict(cfg, seed=0), cfg)      # weights with which the semi-autoregressive mode emits captions
    opt = cfg.to_opt()
    opt.seed = 42 + 1000003 * rank
    opt.bofi_max_batch = max(64, n_img * n)
    if args.dtype == "bf16":
        opt.bofi_train_dtype = torch.bfloat16
        opt.bofi_compute_dtype = torch.bfloat16
    model = models.setup(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev).train()
    tr = XETrainer(model, opt, graph=not args.no_graph)
    # images whose first bound step opens a phrase: one image without any phrase NaNs the whole semi-autoregressive batch
    # (TransformerModel.py:1956-1958, reproduced), which would turn the SAIC half of the step into a no-op
    pool = torch.from_numpy(W.synthetic_att_feats(6 * n_img, 36, cfg.att_feat_size, seed=ATT_SEED + 10 * rank)).to(dev)
    model.eval()
    with torch.no_grad():
        pn = torch.cat([model(torch.zeros(len(c), 0, device=dev), c, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")[2]
                        for c in pool.split(opt.bofi_max_batch)])
    model.train()
    att = pool[pn > 0][:n_img].contiguous()
    if att.size(0) < n_img:
        raise SystemExit("not enough images with a first phrase in the synthetic pool")
    ref = torch.randint(7, cfg.tgt_vocab, (n_img, cfg.seq_length), generator=torch.Generator().manual_seed(rank))

    def score(seq):                                                        # host-side stand-in for the external scorer
        r = ref.repeat_interleave(n, 0)
        return ((seq == r) & (seq > 0)).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1) + 0.01 * (seq > 0).float().sum(1)

    log("rl: warm-up")
    for _ in range(args.warmup):
        loss, rs, rn = tr.rl_step(att, None, score, sample_n=n)
    # executed GEMM FLOPs of one step: eager pass through the launchers (sampling decodes + gradient pass)
    graph, tr.graph = tr.graph, False
    H.gemm_flops(reset=True)
    tr.rl_step(att, None, score, sample_n=n)
    fl_fixed, fl_skip = H.gemm_flops(reset=True)
    tr.graph = graph
    _barrier(world)
    log("rl: timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, rs, rn = tr.rl_step(att, None, score, sample_n=n)
    _barrier(world)
    elapsed = dp.reduce_scalar(time.perf_counter() - t0, "max", device=dev)
    last_default = dict(tr._last_rl)
    ref_info = {"drawn_rows_vs_gradient_pass_rows_max_abs": last_default.get("reference_gap"), "training_forwards_per_step": last_default.get("training_forwards")}
    # the fast form of rounds 1-4 beside it (opt.bofi_rl_reference_estimator = False: samples from the dropout-free inference engine, ONE gradient pass with
    # dropout, replayed as a hipGraph): a few steps
    try:                                                       # (a failure of this leg must not lose the headline line: it is reported beside it)
        model.opt.bofi_rl_reference_estimator = False
        for _ in range(3):
            tr.rl_step(att, None, score, sample_n=n)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(10):
            tr.rl_step(att, None, score, sample_n=n)
        torch.cuda.synchronize(dev)
        fast_info = {"ms_per_step": round((time.perf_counter() - t1) / 10 * 1e3, 2),
                     "what": "samples from the inference engine (no dropout), one gradient pass with dropout as a hipGraph: an importance-weight log-std of ~0.65 per caption "
                             "against the reference's estimator (dev/exp/rl_dropout_gap.py)"}
    except Exception as e:
        log(f"rl: the fast-estimator leg failed: {type(e).__name__}: {e}")
        fast_info = {"error": f"{type(e).__name__}: {e}"}
    del model.opt.bofi_rl_reference_estimator
    tr._last_rl = last_default
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    # iterations of the two sampling loops in which some caption was still open: tokens / phrase lengths are not known here, so
    # the conservative share: (longest sampled caption's phrases + 1) / seq_length, from the step's own bookkeeping
    share = float(tr._last_rl.get("active_share", 1.0))
    roof = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK[args.dtype], "unit": "TFLOP/s", "frac": None, "traffic": None,
            "launch_ms": round(ms, 3),
            "note": "latency-bound at 50 captions: two sampling decodes (20 enqueued iterations each; the semi-autoregressive one runs a decoder "
                    "pass per phrase) with a host round trip for the scores, then the gradient pass.  No reference-structured FLOP figure "
                    "exists for this step (SURVEY.md 8d prices decode and XE only): achieved_executed / frac_executed = GEMM FLOPs this "
                    "build launches per step (library tally; early-out iterations weighted by the share of active ones; the row-list GEMMs of the "
                    "semi-autoregressive iterations >= 2 are not counted -- their row count lives on the device) / wall time per step"}
    roof.update(_executed(fl_fixed, fl_skip, share, ms, args.dtype))
    res = {"metric": "images/sec self-critical step (SAIC + NAIC sampling, re-forward, new_self_critical, backward, all-reduce, Adam)",
           "value": round(n_img * world * args.steps / elapsed, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"self-critical RL step, {n_img} images x {n} sampled captions per GPU in each of the two modes, "
                                  f"36x2048 regions, d_model=512 6+6(+1) layers, {args.dtype}",
                      "images_per_step_per_gpu": n_img, "samples_per_image": n, "final_loss": round(float(loss), 5),
                      "saic_tokens_per_sample": round(float(tr._last_rl["saic_tokens"]), 2), "naic_tokens_per_sample": round(float(tr._last_rl["naic_tokens"]), 2),
                      "scorer": "host-side stand-in (the reference's CIDEr-D scorer is external)", "hip_graph": bool(graph),
                      "active_iteration_share": round(share, 3),
                      "estimator": "the reference's own (loss_wrapper.py:193-209; the default since round 5): every token of both branches drawn from the rows the gradient "
                                   "pass differentiates (one tape-free training forward per phrase under the step's counter-based dropout masks), then that forward with the tape; "
                                   "fast_estimator = rounds 1-4's form, opt.bofi_rl_reference_estimator = False",
                      "reference_estimator": ref_info, "fast_estimator": fast_info},
           "roofline": roof}
    if cpu and world == 1:
        _cpu_leg(res, lambda: cpu_baseline_rl(cfg, sd, n_img, n, budget_s=args.cpu_budget))
    return res


def cpu_baseline_refine(cfg, sd, rounds, seed, budget_s=15.0, batch=256):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import boficap_oracle as O
    from boficap_amd import weights as W
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("BOFI_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    w = O.as_torch(sd)
    att = torch.from_numpy(W.synthetic_att_feats(batch, 36, cfg.att_feat_size, seed=seed))      # (the SAME workload as the GPU leg: its batch size -- 256 for config 5)
    with torch.no_grad():
        O.sample_naic_refine(w, cfg, att, rounds=rounds)
        times, t_end = [], time.time() + budget_s
        while len(times) < 3 and (time.time() < t_end or not times):
            t0 = time.time()
            O.sample_naic_refine(w, cfg, att, rounds=rounds)
            times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{len(times)} x one batch of {batch} images with {rounds} refinement rounds, fp32 torch-CPU oracle, median"}


def run_naic(args, ctx, log, cpu=True, gemm_roofline=True):
    from boficap_amd import dp, hip as H
    rank, local_rank, world, dev = ctx
    from boficap_amd import weights as W
    from boficap_amd.config import FULL as cfg
    from boficap_amd.engine import BofiEngine

    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    log("building weights")
    sd = W.make_state_dict(cfg, seed=0)
    import math
    C = max(1, args.coalesce)
    if args.steps % C:                                          # K timed steps = K / C launches: the largest batch count that divides K
        C = math.gcd(C, args.steps)
        log(f"--steps is not a multiple of --coalesce {args.coalesce}: {C} batches per launch")
    warm_launches = (args.warmup + C - 1) // C                  # (the untimed warm-up is rounded UP to whole launches)
    eng = BofiEngine(cfg, tdt, max_batch=args.batch * C, max_regions=36, device=dev)
    eng.load_state_dict(sd)
    # every rank decodes its own shard of images (different seed per rank), already resident in HBM; with --coalesce the
    # C batches of a launch are C different batches (same generator, more images)
    att = torch.from_numpy(W.synthetic_att_feats(args.batch * C, 36, cfg.att_feat_size, seed=ATT_SEED + 10 * rank)).to(dev).to(tdt).contiguous()
    qg = args.batch if C > 1 else 0
    graph = not args.no_graph
    # executed GEMM FLOPs of one decode: the library's tally over one eager call
    H.gemm_flops(reset=True)
    probe = eng.decode_naic(att, want_logprob=not args.ids_only, graph=False, refine_rounds=args.refine, q1_group=qg)
    # quirk Q1: a batch whose LAST image lays out no phrase decodes to NaN as a whole (TransformerModel.py:1872-1873).  The slot
    # layout of an image does not depend on its place, so such a batch gets one of its other images moved to the end.
    pn = probe["phrase_num"].cpu()
    moved = 0
    for g0 in range(0, att.size(0), args.batch):
        last = g0 + args.batch - 1
        if int(pn[last]) == 0:
            alive = [i for i in range(g0, last) if int(pn[i]) > 0]
            if alive:
                i = alive[-1]
                att[[i, last]] = att[[last, i]]
                moved += 1
    if moved:
        log(f"{moved} batch(es) ended on an image without phrases: reordered")
        probe = eng.decode_naic(att, want_logprob=not args.ids_only, graph=False, refine_rounds=args.refine, q1_group=qg)
        H.gemm_flops(reset=True)
        probe = eng.decode_naic(att, want_logprob=not args.ids_only, graph=False, refine_rounds=args.refine, q1_group=qg)
    fl_fixed, fl_skip = H.gemm_flops(reset=True)
    T = int(probe["bound_iters"].item())
    del probe
    T_all = T                                                   # ... over every input set a launch will carry (the one-batch form decodes other images per stream)
    log("engine ready; first decode (graph capture)")
    # streams that provably overlap (distinct hardware queues), found by timing a spin kernel on pairs
    from boficap_amd.engine import pick_concurrent_streams
    if args.cu_partitions > 1:
        # experiment: every stream is confined to one of P disjoint sets of CUs (hipExtStreamCreateWithCUMask; a mask must leave every XCD
        # some CUs -- bit i is CU i / 8 of XCD i % 8 -- so a set is the same CU range of every XCD), stream k on set k % P
        import ctypes
        hiprt = ctypes.CDLL("libamdhip64.so")
        hiprt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        by_xcd = os.environ.get("BOFI_BENCH_PARTITION_BY_XCD") == "1"      # experiment: stream k on XCDs {k*8/P ..} only (its weights alone in those L2s)
        P, per = args.cu_partitions, 32 // args.cu_partitions
        streams = []
        for k in range(max(1, args.inflight)):
            part = k % P
            bits = ({i for i in range(256) if part * (8 // P) <= i % 8 < (part + 1) * (8 // P)} if by_xcd
                    else {i for i in range(256) if part * per <= i // 8 < (part + 1) * per})
            words = (ctypes.c_uint32 * 8)(*[sum((1 << b) for b in range(32) if (wd * 32 + b) in bits) for wd in range(8)])
            h = ctypes.c_void_p()
            if hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words) != 0:
                raise RuntimeError("hipExtStreamCreateWithCUMask failed")
            streams.append(torch.cuda.ExternalStream(h.value, device=dev))
    else:
        prio = [int(v) for v in os.environ.get("BOFI_BENCH_STREAM_PRIO", "").split(",") if v.strip()] or None      # experiment: stream priorities (-1 = high), e.g. "-1,0"
        streams = pick_concurrent_streams(args.inflight, dev, priorities=prio) if args.inflight > 1 else [torch.cuda.current_stream(dev)]
        if args.inflight > 1 and len(streams) < args.inflight:                       # an unlucky draw of hardware queues: look among more candidates once
            more = pick_concurrent_streams(args.inflight, dev, candidates=48)
            streams = more if len(more) > len(streams) else streams
    log(f"{len(streams)} concurrent streams")
    engines = [eng] + [eng.fork() for _ in range(len(streams) - 1)]
    for e in engines:
        e.set_decodes_in_flight(args.hint or len(engines))    # (--hint: a profiling run that serialises launches but wants the headline's kernel forms)
    # every launch in flight decodes features of its own (a rotation of the batches by whole batches, so the layouts and T stay
    # those of the probe): none of them finds another's features warm in a cache
    nb = att.size(0) // args.batch
    atts = [att] + [torch.cat([att[(k % nb) * args.batch:], att[:(k % nb) * args.batch]]).contiguous() if nb > 1 else att.clone()
                    for k in range(1, len(engines))]
    if nb == 1:                                                 # one batch per launch: other images (same generator, other seeds), Q1-safe
        for k in range(1, len(engines)):
            cand = torch.from_numpy(W.synthetic_att_feats(args.batch, 36, cfg.att_feat_size, seed=ATT_SEED + 10 * rank + 1000 * k)).to(dev).to(tdt)
            pk = eng.decode_naic(cand, want_logprob=False, graph=False, refine_rounds=args.refine)
            alive = (pk["phrase_num"] > 0).nonzero().flatten()
            if int(pk["phrase_num"][-1]) == 0 and alive.numel():
                i = int(alive[-1])
                cand[[i, args.batch - 1]] = cand[[args.batch - 1, i]]
            T_all = max(T_all, int(pk["bound_iters"].item()))   # (an image's layout, and with it its iteration count, does not depend on its place)
            atts[k] = cand.contiguous()
    # ... and every stream alternates between TWO feature tensors (the second: the same batches in another order), so that consecutive
    # launches of a stream do not replay one Infinity-Cache-resident input (round-2 VERDICT)
    atts2 = [torch.cat([a_k[args.batch:], a_k[:args.batch]]).contiguous() if nb > 1 else a_k.clone() for a_k in atts]
    # iteration budget: T_all + 1 bounding iterations enqueued instead of seq_length (one idle iteration is what the check below needs: a decode whose live count
    # stays BELOW the budget provably ended inside it; T_all is exact -- the probe decodes ran these very inputs); every decode reports into `live_word` (atomic max)
    S_it = cfg.seq_length
    cap = T_all + 1 if args.iter_budget == "auto" and T_all + 1 < S_it else 0
    # round 5: under the hint of several decodes in flight the bounding loop is ONE persistent kernel per 16 images that leaves when its images are
    # finished (bound_loop.hip): nothing is enqueued per iteration, so there is no budget to set (the engine would ignore it)
    loop_kernel = all(e.bound_loop_active(36) for e in engines)
    if loop_kernel:
        cap = 0
    if cap and os.environ.get("BOFI_BENCH_ITER_CAP"):           # (tests: a budget the decodes outrun, to walk the re-run)
        cap = int(os.environ["BOFI_BENCH_ITER_CAP"])
    live_word = torch.zeros(1, dtype=torch.int32, device=dev)
    for e in engines:
        e.watch_live_iterations(live_word)

    def budget_held(leg):
        """after a leg's synchronisation: did every decode since the last check end inside the budget?"""
        mx = int(live_word.item())
        live_word.zero_()
        if cap and mx >= cap:
            log(f"iteration budget {cap} missed in the {leg} leg (a decode had {mx} live iterations)")
            return False
        return True
    outs = []
    for e, st, a_k, b_k in zip(engines, streams, atts, atts2):
        with torch.cuda.stream(st):
            outs.append(e.decode_naic(a_k, want_logprob=not args.ids_only, graph=graph, refine_rounds=args.refine, q1_group=qg, iter_cap=cap))
            e.decode_naic(b_k, graph=graph, out=outs[-1], refine_rounds=args.refine, q1_group=qg, iter_cap=cap)       # (captures the second input's graph)
            e.decode_naic(a_k, graph=graph, out=outs[-1], refine_rounds=args.refine, q1_group=qg, iter_cap=cap)
    torch.cuda.synchronize()
    budget_ok = budget_held("capture")
    out = outs[0]
    log("warm-up + timed steps")

    def step(i):
        k = i % len(engines)
        feats = atts2[k] if (i // len(engines)) % 2 else atts[k]
        with torch.cuda.stream(streams[k]):
            engines[k].decode_naic(feats, graph=graph, out=outs[k], refine_rounds=args.refine, q1_group=qg, iter_cap=cap)

    launches = args.steps // C
    # untimed warm-up: at least --warmup steps, rounded up to whole rounds over the streams (EVERY stream replays its graph before the clock starts)
    warm_launches = (warm_launches + len(engines) - 1) // len(engines) * len(engines)
    for i in range(warm_launches):
        step(i)
    # the timed region = exactly --steps steps between a barrier + synchronisation on both sides.  A short region (the driver's 20 steps are 4
    # launches = 6 ms) is one sample of a noisy quantity: it is then measured `regions` times back to back, each bracketed the same way, and
    # the MEDIAN region is reported (config.timed_regions / region_ms say so)
    regions = 5 if launches <= 4 * len(engines) else 1
    spans = []
    for rep in range(regions):
        _barrier(world)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        t0 = time.perf_counter()
        for st, e in zip(streams, ev0):
            e.record(st)
        for i in range(launches):
            step(i)
        for st, e in zip(streams, ev1):
            e.record(st)
        _barrier(world)
        el = dp.reduce_scalar(time.perf_counter() - t0, "max", device=dev)
        # HIP events on the launch streams: with one stream this is the device time per decode; with several
        # in flight it is the longest stream's span divided by all the decodes (steady-state time per decode)
        spans.append((el, max(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / launches))
    spans.sort()
    elapsed, dev_ms = spans[len(spans) // 2]
    region_ms = [round(e * 1e3, 3) for e, _ in spans]
    budget_ok = budget_held("timed") and budget_ok

    # for the record: the same K steps strictly one at a time (latency view of the same workload), HIP events on the stream
    single_ms = None
    if len(engines) > 1:
        # (a caller that runs one decode at a time says so: the engine then takes the sublayer kernels' latency forms -- a hint, part of the graph key)
        eng.set_decodes_in_flight(1)
        for i in range(max(warm_launches, 3)):
            eng.decode_naic(att, graph=graph, out=outs[0], refine_rounds=args.refine, q1_group=qg, iter_cap=cap)
        _barrier(world)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for i in range(launches):
            eng.decode_naic(att, graph=graph, out=outs[0], refine_rounds=args.refine, q1_group=qg, iter_cap=cap)
        s1.record()
        _barrier(world)
        single_ms = s0.elapsed_time(s1) / args.steps
        budget_ok = budget_held("one-at-a-time") and budget_ok
        eng.set_decodes_in_flight(len(engines))
    # for the record: the same launches with the features starting in pinned HOST memory (a loader's numpy arrays), copied to the
    # device on the launch's own stream before every launch -- the PCIe-inclusive rate (never `value`)
    pcie_ms, pcie_note = None, None
    if args.from_host:
        # through boficap_amd.engine.DecodePipeline -- what tools/eval.py and TransformerModel.decode_many run: 3 launches in flight of 16 batches each, the
        # features copied from pinned host memory on a copy stream ahead of the launches (3 launch streams + the copy stream = the runtime's 4 hardware
        # queues), ids / slot layouts / per-image entropy and perplexity back on the host per batch
        from boficap_amd.engine import DecodePipeline
        pipe = DecodePipeline(eng, in_flight=3, batches_per_launch=16)
        pool = torch.cat(atts).cpu()
        pool = torch.cat([pool] * (-(-64 * 320 // pool.size(0)))).pin_memory()                # >= 320 batches of 64 (20 launches), whatever --steps says
        hb = [pool[i:i + args.batch] for i in range(0, pool.size(0) - args.batch + 1, args.batch)]
        for _ in pipe.run(hb[:96]):
            pass
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        n_img = sum(r["seq"].size(0) for r in pipe.run(hb))
        pcie_ms = (time.perf_counter() - h0) / (n_img / args.batch) * 1e3
        pcie_note = f"{len(hb)} batches of {args.batch} through DecodePipeline (3 launches in flight x 16 batches, copy stream ahead, results on the host)"
        del pipe, pool, hb
        torch.cuda.empty_cache()
    traffic, tnote = None, "no PMC pass committed for this configuration"
    names = {1: ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json"), 4: ("r02_hbm_traffic_coalesce4.json",),
             16: ("r06_hbm_traffic_coalesce16.json", "r05_hbm_traffic_coalesce16.json"),
             5: ("r06_hbm_traffic_coalesce5.json", "r05_hbm_traffic_coalesce5.json", "r04_hbm_traffic_coalesce5.json", "r03_hbm_traffic_coalesce5.json", "r02_hbm_traffic_coalesce5.json")}.get(C, ())
    config5 = args.batch == 256 and args.refine == 3 and C == 1 and args.dtype == "bf16"      # (BASELINE config 5: its own PMC pass, profiles/r06_hbm_traffic_config5.json)
    if config5:
        names = ("r06_hbm_traffic_config5.json",)
    for name in names:                                          # HBM bytes per launch from the newest committed PMC run of this configuration
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath) and config5:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_step")
            tnote = (f"HBM-side bytes per launch (one batch of 256, 3 refinement rounds), rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, "
                     f"one launch at a time (profiles/{name})")
            break
        if os.path.exists(tpath) and args.batch == 64 and args.dtype == "bf16" and not args.refine:
            with open(tpath) as f:
                tj = json.load(f)
                traffic = tj.get("hbm_bytes_per_decode", tj.get("hbm_bytes_per_step"))
            tnote = (f"HBM-side bytes per launch ({C} batch(es) of 64), rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, "
                     f"one launch at a time (profiles/{name}); algorithmic minimum = {9.4 * C:.1f} MB features + 125 MB weights + {48.6 * C:.1f} MB log-probs")
            break
    if dp.reduce_scalar(0.0 if budget_ok else 1.0, "max", device=dev) > 0:      # some rank's decode outran the budget: the whole measurement again, every iteration enqueued
        import copy
        for e in engines:
            e.watch_live_iterations(None)
        del engines, outs
        torch.cuda.empty_cache()
        a = copy.copy(args)
        a.iter_budget = "off"
        return run_naic(a, ctx, log, cpu, gemm_roofline)
    ntok = float(out["phrase_length"].sum(1).float().mean().item())
    nan = bool(out["seq_logprob"].isnan().any().item()) if out["seq_logprob"] is not None else False
    if rank != 0:
        return None
    images = args.batch * world * args.steps
    dl = 2 * cfg.seq_length * (6 * cfg.d_model ** 2 + 2 * cfg.d_model * cfg.d_ff) + 4 * cfg.seq_length * (cfg.seq_length + 36) * cfg.d_model
    flops_launch = (f_alg(T, cfg) + args.refine * (cfg.N_dec * dl + 2 * cfg.seq_length * cfg.d_model * cfg.tgt_vocab)) * args.batch * C
    achieved = flops_launch / (dev_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK[args.dtype], "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK[args.dtype], 5), "traffic": traffic, "traffic_note": tnote,
            "kernel": ("whole decode = one hipGraph launch of the path's kernels" if graph else "whole decode (eager launches)")
                      + (f", {C} batches of {args.batch} per launch (quirk Q1 per batch)" if C > 1 else "")
                      + (f", {len(engines)} launches in flight" if len(engines) > 1 else ""),
            "flops_per_launch": flops_launch, "launch_ms": round(dev_ms, 4),
            "note": "algorithmic FLOPs F_alg(T)*batch (SURVEY.md 8d) / HIP-event time per decode on the launch stream; "
                    "achieved_executed: the GEMM FLOPs the decode launches (library tally, idle bound iterations weighted by T / seq_length) / the same time"}
    roof.update(_executed(fl_fixed, fl_skip, T / cfg.seq_length, dev_ms, args.dtype))
    if single_ms:
        one = flops_launch / C / (single_ms * 1e-3) / 1e12
        roof["one_at_a_time"] = {"launch_ms": round(single_ms * C, 4), "achieved": round(one, 2), "frac": round(one / MFMA_PEAK[args.dtype], 5)}
    res = {
        "metric": "images/sec NAR bound+fill greedy decode (NAIC _sample)"
                  + (f"; dynamic batching: {C} batches of {args.batch} images per engine launch ({C * args.batch} images), " if C > 1 else f"; one batch of {args.batch} per launch, ")
                  + f"{len(engines)} launch(es) in flight",
        "value": round(images / elapsed, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"batch={args.batch} NAR bound+fill inference, 36x2048 regions, d_model=512 6+6 layers(+1 bound layer), {args.dtype}"
                               + (f", {args.refine} refinement rounds" if args.refine else ""),
                   "images_per_step_per_gpu": args.batch, "bound_iterations": T, "bound_iterations_enqueued": None if loop_kernel else (cap if cap else cfg.seq_length),
                   "bound_loop": ("one persistent kernel per 16 images runs every iteration of core_NAIC's loop and leaves when its images are finished (bound_loop.hip; "
                                  "fp16 operands from the float32 parameters; launches of <= 384 images: TWO workgroups per group share the feed-forward's weight stream, bit-identical)" if loop_kernel else "five launches per iteration (bound_ops.hip, naic.hip)"),
                   "knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("BOFI_")},
                   "iteration_budget": ("not applicable: the loop kernel ends by itself" if loop_kernel else f"{cap} of {cfg.seq_length} bounding iterations enqueued per decode (largest live count of the probe decodes + 1; the reference's loop "
                                        "stops when every image is finished, TransformerModel.py:1869); every decode folds its live-iteration count into a device word "
                                        "(atomic max) that was read after each leg: all below the budget, else this line would come from a full re-run without it"
                                        if cap else "off: every decode enqueues all seq_length iterations"),
                   "mean_tokens_per_image": round(ntok, 2),
                   "vocab": cfg.tgt_vocab, "seq_logprob_materialised": not args.ids_only, "hip_graph": graph,
                   "decodes_in_flight": len(engines), "features_per_launch_in_flight": "own tensors, two per stream, alternated", "batches_per_launch": C,
                   "timed_regions": regions, "region_ms": region_ms, "warmup_steps_run": warm_launches * C,
                   "timing": (f"the {args.steps}-step region ({launches} launches) timed {regions} times back to back, each between a barrier + synchronisation; value / "
                              "ms_per_step are the MEDIAN region's" if regions > 1 else f"one region of {args.steps} steps ({launches} launches)"),
                   "images_per_launch": C * args.batch, "refine_rounds": args.refine,
                   "one_at_a_time_ms_per_step": round(single_ms, 4) if single_ms else round(elapsed / args.steps * 1e3, 4),
                   "weights": "seeded Xavier init + calibrated bound heads (boficap_amd.weights, seed 0)",
                   "features_from_pinned_host_ms_per_step": round(pcie_ms, 4) if pcie_ms else None, "features_from_pinned_host_how": pcie_note, "att_feats_seed": ATT_SEED, "batches_reordered_for_q1": moved, "nan_in_output": nan, "sharding": "images by rank, no collective"},
        "roofline": roof,
    }
    if world == 1 and gemm_roofline:
        res["roofline_gemm"] = gemm_rooflines(tdt, dev, C)
    log(f"gpu done: {res['value']} images/sec")
    if cpu and world == 1:
        refine_rounds, batch_n, budget_s = args.refine, args.batch, args.cpu_budget
        _cpu_leg(res, (lambda: cpu_baseline_refine(cfg, sd, refine_rounds, ATT_SEED, budget_s=budget_s, batch=batch_n)) if refine_rounds
                 else (lambda: cpu_baseline(cfg, sd, batch_n, ATT_SEED, budget_s=budget_s)))
    del engines, outs, eng
    return res


def _run_cpu_legs(log):
    while _DEFERRED_CPU:
        res, fn = _DEFERRED_CPU.pop(0)
        if res is not None:
            log("timing the CPU oracle: " + str(res.get("metric", ""))[:60])
            res["cpu_baseline"] = fn()


def _compact(res):
    """A secondary measurement as it rides on the headline line: the contract fields plus the roofline and the CPU baseline."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "cpu_baseline")
    out = {k: res[k] for k in keep if k in res}
    out["workload"] = res["config"]["workload"]
    out["config"] = {k: v for k, v in res["config"].items() if k not in ("workload", "weights", "sharding")}
    return out


def run_dry(args, ctx, log):
    """BOFI_BENCH_REHEARSAL=dry: `python bench.py --gpus N` with the GPU work of every rank replaced by a sleep -- the multi-rank SHAPE of this script and nothing else
    (self-launch through torch.distributed.run, WORLD_SIZE against --gpus, the process group, images sharded by rank, barrier-bracketed region, the MAX over ranks of
    the elapsed time, value = every rank's images over that time, ONE line from rank 0, a rank that fails before the exchange -> a non-zero exit of the job).  What an
    8-GPU scaling run exercises that no one-GPU box can (at most 6 processes may share its card); run on the CPU over gloo by tests/test_dp_gloo.py.  The line is
    marked `rehearsal` and `data: none`: it measures nothing and is no product path (the decode has no CPU form: BofiHipError without the library or a device)."""
    import torch.distributed as dist
    from boficap_amd import dp
    rank, local_rank, world, dev = ctx
    fail = os.environ.get("BOFI_BENCH_DRY_FAIL_RANK")
    lo, hi = dp.shard_range(args.batch * world * args.steps, rank, world)        # the images of the whole job, by rank (the decode shards with no collective)
    if fail is not None and int(fail) == rank:
        raise RuntimeError(f"rank {rank}: injected failure before the exchange (BOFI_BENCH_DRY_FAIL_RANK)")
    per_step_s = 1e-3 * (1.0 + 0.25 * rank)                                        # the slowest rank sets the job's time
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(per_step_s * args.steps)
    if world > 1:
        dist.barrier()
    elapsed = dp.reduce_scalar(time.perf_counter() - t0, "max", device=dev)
    mine = torch.tensor([hi - lo], dtype=torch.int64)
    shards = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(shards, mine)
    else:
        shards = [mine]
    total = dp.reduce_scalar(float(hi - lo), "sum", device=dev)
    if rank != 0:
        return None
    return {"metric": "DRY WALK of the multi-rank path (no GPU work): images/sec of sleeping ranks", "value": round(total / elapsed, 1), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none", "data": "none",
            "config": {"workload": "dry walk: every rank sleeps (1 + rank / 4) ms per step", "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                       "dist_backend": dist.get_backend() if world > 1 else None, "shard_images_per_rank": [int(t.item()) for t in shards],
                       "images_per_step_per_gpu": args.batch, "slowest_rank_ms_per_step": round(1e3 * 1e-3 * (1.0 + 0.25 * (world - 1)), 4),
                       "rehearsal": "dry: no GPU work, the launcher / sharding / barrier / reduction plumbing only"}}


def choose_coalesce(steps: int, inflight: int) -> int:
    """Batches of 64 per engine launch for a timed region of ``steps`` batches on ``inflight`` streams: among 16, 20, 10, 8, 5, 4, 2 the count that divides the steps and
    spreads the launches evenly over the streams (fewest rounds x batches per launch), the first of equals; 1 when none divides."""
    cands = [c for c in (16, 20, 10, 8, 5, 4, 2) if steps % c == 0]
    if not cands:
        return 1
    n_str = max(1, inflight)
    return min(cands, key=lambda c: (-(-(steps // c) // n_str)) * c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="naic", choices=["naic", "xe", "rl"],
                    help="naic: bound+fill decode (headline); xe: XE training step (config 3); rl: self-critical step (config 4)")
    ap.add_argument("--seq-per-img", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=None, help="images per step and GPU (default: 64; 10 in rl mode)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--ids-only", action="store_true", help="do not materialise the [B,20,V] log-prob tensor")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=0, help="xe mode: 1 = the forward's four branches on HIP streams of their own")
    ap.add_argument("--refine", type=int, default=0, help="extra filling rounds (BASELINE config 5 uses 3 rounds at batch 256)")
    ap.add_argument("--inflight", type=int, default=4, help="decodes in flight on separate HIP streams (engine forks sharing the weights); "
                    "1 = strictly one decode at a time")
    ap.add_argument("--hint", type=int, default=0, help="profiling: the decodes-in-flight hint handed to the engines whatever --inflight says (0 = --inflight): "
                    "`--inflight 1 --hint 4` runs the headline's throughput kernel forms one launch at a time (counter passes serialise anyway)")
    ap.add_argument("--cu-partitions", type=int, default=1, help="experiment: confine stream k to CU set k %% P of P disjoint sets (the same CU range of "
                    "every XCD; set BOFI_GEMM_PERS_GRID to 256 / P with it); 1 = off")
    ap.add_argument("--coalesce", type=int, default=None, help="dynamic batching: C consecutive steps (batches of --batch images) share ONE engine "
                    "launch; quirk Q1 stays per batch (q1_group), so every step's outputs equal its own separate decode.  K steps = K/C launches.  "
                    "Default for the plain batch-64 decode: 5 or 4 (whichever divides --steps into evenly spread launches), 1 otherwise")
    ap.add_argument("--from-host", action="store_true", default=None, help="also time the launches with the features copied from pinned host memory "
                    "before each one (PCIe-inclusive rate, reported in config; never the headline value).  Default: on for the plain headline run")
    ap.add_argument("--no-from-host", dest="from_host", action="store_false")
    ap.add_argument("--iter-budget", default="auto", choices=["auto", "off"],
                    help="naic: bounding iterations ENQUEUED per decode.  off = all seq_length of them, as round 2 (an iteration past the last live one is five "
                         "launches that return at once).  auto = the largest count of live iterations among the probe decodes + 2; every decode folds its own "
                         "count into a device word (atomic max) that is read after each timed leg: were a decode to need more, the whole measurement is redone "
                         "without the budget.  The reference's loop stops when every image is finished (TransformerModel.py:1869) -- it never runs the idle ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-roofline", action="store_true", help="skip the per-shape GEMM timing (use under rocprofv3 so that the trace holds decodes only)")
    ap.add_argument("--no-secondary", action="store_true", help="headline measurement only (no XE / RL / refinement lines)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-oracle work per baseline")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.batch is None:
        args.batch = 10 if args.mode == "rl" else 64
    if args.from_host is None:
        args.from_host = args.mode == "naic" and args.batch == 64 and not args.refine and args.coalesce is None and not args.no_secondary
    default_coalesce = args.coalesce is None
    if default_coalesce:
        args.coalesce = 1
        if args.mode == "naic" and args.batch == 64 and not args.refine:
            # batches per launch: among 16, 20, 10, 8, 5, 4, 2 the one that divides the timed steps and spreads the launches evenly over the
            # streams (fewest rounds x batches per launch, the first of equals: the default --steps 320 -> 20 launches of 16 batches, five per stream; the
            # driver's --steps 20 -> 4 launches of 5 batches, one per stream).  Round 5: the chip wants ~3 000 images in flight -- 254 k img/s at 10 batches
            # per launch, 260 k at 12, 262 k at 16-20 against 239-243 k at 5 (profiles/r05_launch_shape_sweep.txt)
            args.coalesce = choose_coalesce(args.steps, args.inflight)
    ctx = dist_setup(args)
    rank, world = ctx[0], ctx[2]

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    cpu = not args.no_cpu_baseline
    if os.environ.get("BOFI_BENCH_REHEARSAL") == "dry":
        res = run_dry(args, ctx, log)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == "xe":
        res = run_xe(args, ctx, log, cpu)
    elif args.mode == "rl":
        res = run_rl(args, ctx, log, cpu)
    else:
        res = run_naic(args, ctx, log, cpu, not args.no_gemm_roofline)
        plain = (args.batch == 64 and not args.refine and default_coalesce and args.dtype == "bf16" and not args.ids_only and not args.no_graph)
        if world == 1 and plain and not args.no_secondary:
            import copy
            raw = {}
            torch.cuda.empty_cache()
            # steady state: the headline's timed region is what the caller asked for -- the driver's `--steps 20` holds 4 launches of 5 batches,
            # a quarter of the images in flight the chip is fullest at; this is the same workload through 320 steps at 16 batches per launch (what `python bench.py` and
            # tools/eval.py run), driver-run beside it.  `value` stays the region the flags asked for.
            if args.coalesce == 16 and args.steps >= 320:
                raw["naic_steady_state"] = res                   # (the headline region IS the steady-state shape)
            else:
                a = copy.copy(args); a.coalesce, a.steps, a.warmup, a.from_host = 16, 320, 64, False
                raw["naic_steady_state"] = run_naic(a, ctx, log, False, False)
                torch.cuda.empty_cache()
            a = copy.copy(args); a.coalesce, a.steps, a.warmup = 1, 80, 16       # one batch of 64 per launch, 4 launches in flight (round-1 headline form)
            raw["naic_one_batch_per_launch"] = run_naic(a, ctx, log, False, False)
            torch.cuda.empty_cache()
            a = copy.copy(args); a.mode, a.batch, a.steps, a.warmup = "xe", 64, 20, 5
            raw["xe_config3"] = run_xe(a, ctx, log, cpu)
            torch.cuda.empty_cache()
            a = copy.copy(args); a.mode, a.batch, a.steps, a.warmup = "rl", 10, 10, 3
            raw["rl_config4"] = run_rl(a, ctx, log, cpu)
            torch.cuda.empty_cache()
            a = copy.copy(args); a.batch, a.refine, a.steps, a.warmup, a.coalesce = 256, 3, 40, 8, 1
            raw["refine_config5"] = run_naic(a, ctx, log, cpu, False)
            _run_cpu_legs(log)                                   # (every GPU leg is through: now the CPU oracle's)
            sec = {k: _compact(v) for k, v in raw.items()}
            res["secondary"] = sec
            # the driver's record keeps `config` whole: the secondary headline scalars ride there too
            res["config"].update(steady_state_img_s=sec["naic_steady_state"]["value"], steady_state_ms_per_step=sec["naic_steady_state"]["ms_per_step"],
                                 steady_state_roofline_frac=sec["naic_steady_state"]["roofline"].get("frac"),
                                 one_batch_per_launch_img_s=sec["naic_one_batch_per_launch"]["value"],
                                 xe_config3_ms_per_step=sec["xe_config3"]["ms_per_step"], xe_config3_img_s=sec["xe_config3"]["value"],
                                 xe_config3_frac_executed=sec["xe_config3"]["roofline"].get("frac_executed"),
                                 rl_config4_ms_per_step=sec["rl_config4"]["ms_per_step"],
                                 refine_config5_img_s=sec["refine_config5"]["value"], refine_config5_ms_per_step=sec["refine_config5"]["ms_per_step"])
            res["secondary_note"] = ("naic_steady_state: the headline workload through 320 steps at 16 batches per launch, 4 launches in flight (the shape `python bench.py` "
                                     "and tools/eval.py run; the headline `value` is the region the flags asked for); naic_one_batch_per_launch: the headline workload with one batch of 64 per engine launch; then "
                                     "driver-run lines of BASELINE configs 3, 4, 5 (short runs in the same process, after the headline measurement; every CPU-oracle "
                                     "leg after all of them); config 5's 'autoregressive fallback' has no counterpart: a UIC checkpoint has no AR decode path in the reference "
                                     "(TransformerModel.py:1791-1804 needs EncoderDecoder.decode, :1287-1310)")
        if world > 1 and plain and not args.no_secondary:
            # the decode shards with no collective: the scaling run exercises RCCL through this secondary (the XE step's gradient exchange).
            # Never at the cost of the headline line, and never silently: the line says how many ranks the collective backend saw
            # (config.rccl_ranks / dist_backend) and carries config.xe_dp_error when the exchange failed.  A rank that fails INSIDE a step leaves the
            # others in a collective -- a hang, not an exception -- so every rank arms a watchdog: past its limit rank 0 prints the headline line
            # with the error and every process leaves.
            import threading
            import torch.distributed as dist
            if rank == 0 and res is not None:
                res["config"].update(rccl_ranks=dist.get_world_size(), dist_backend=dist.get_backend())
            limit = float(os.environ.get("BOFI_BENCH_DP_LIMIT_S", "420"))

            def give_up():
                # a rank stuck in a collective (or in a GPU kernel) after touching the GPU: the headline line still goes out (rank 0), with the phase that was
                # reached, and EVERY process ends with a non-zero status -- the launcher and the driver must see the failure
                sys.stderr.write(f"[bench rank {rank}] data-parallel secondary: no result within {limit:.0f} s; last phase reached: {DP_PHASE[0]}\n")
                sys.stderr.flush()
                if rank == 0 and res is not None:
                    res["config"]["xe_dp_error"] = f"no result within {limit:.0f} s: a rank hung or failed inside the exchange (rank 0's last phase: {DP_PHASE[0]}); exit status 3"
                    print(json.dumps(res), flush=True)
                os._exit(3)
            dog = threading.Timer(limit, give_up)
            dog.daemon = True
            dog.start()
            try:
                dp_res = run_xe_dp(args, ctx, log)
                if rank == 0 and res is not None:
                    res.setdefault("secondary", {})["xe_config3_dp"] = dp_res
                    res["config"].update(xe_dp_step_ms=dp_res["fp32_ring_all_reduce"]["step_ms"], xe_dp_exposed_collective_ms=dp_res["fp32_ring_all_reduce"]["exposed_collective_ms"],
                                         xe_dp_bf16_wire_step_ms=dp_res["bf16_mesh_direct"]["step_ms"], xe_dp_img_s=dp_res["fp32_ring_all_reduce"]["images_per_sec"])
            except Exception as e:
                log(f"xe dp secondary failed: {type(e).__name__}: {e}")
                if rank == 0 and res is not None:
                    res.setdefault("secondary", {})["xe_config3_dp"] = {"error": f"{type(e).__name__}: {e}"}
                    res["config"]["xe_dp_error"] = f"{type(e).__name__}: {e}"
            finally:
                dog.cancel()
    _run_cpu_legs(log)                                           # (those not run yet: the headline's, or a single mode's)
    if rank == 0 and res is not None:
        if os.environ.get("BOFI_BENCH_REHEARSAL") == "1":
            res["config"]["rehearsal"] = "all ranks on one device over gloo: a walk through the N > 1 code path, not a measurement"
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
