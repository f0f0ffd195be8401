"""N > 1 path on CPU: two gloo processes shard a batch by rank; no collective on the data path."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import boficap_oracle as O
    from boficap_amd import dp, weights as W
    from boficap_amd.config import TINY
    torch.set_num_threads(2)
    r, lr, w = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = TINY
    sd = O.as_torch(W.make_state_dict(cfg, 0, gen_scale=6.0))
    att = torch.from_numpy(W.synthetic_att_feats(10, 36, cfg.att_feat_size, seed=3))
    a, b = dp.shard_range(10, rank, world)
    # the checker stands in for the device path here: per-shard decode with the per-row fill mask
    seq, lp, pn, pl, ps, _ = O.sample_naic(sd, cfg, att[a:b], fix_q1=True)
    pad = torch.zeros(5 - (b - a), seq.size(1), dtype=seq.dtype)
    allseq = dp.gather_rows(torch.cat([seq, pad]))                       # equal shapes for all_gather
    n_img = dp.reduce_scalar(b - a, "sum")
    t_max = dp.reduce_scalar(1.0 + rank, "max")
    if rank == 0:
        full = O.sample_naic(sd, cfg, att, fix_q1=True)[0]
        got = torch.cat([allseq[0:5], allseq[5:10]])
        ret.put((bool(torch.equal(got, full)), n_img, t_max))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_sharded_decode_matches_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    ok, n_img, t_max = ret.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and n_img == 10 and t_max == 2.0


def test_shard_range_covers_everything_once():
    from boficap_amd.dp import shard_range
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _bucket_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from boficap_amd import dp
    from boficap_amd.trainer import FlatBucket
    torch.set_num_threads(1)
    dp.init_from_env("gloo")
    torch.manual_seed(0)                                                  # same initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    before = [p.detach().clone() for p in net.parameters()]
    bucket = FlatBucket(net)
    same_values = all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    x = torch.full((4, 7), float(rank + 1))
    bucket.zero_grad()
    net(x).sum().backward()                                               # autograd accumulates INTO the flat views
    in_bucket = all(p.grad.data_ptr() == bucket.grad.data_ptr() + o * 4 for p, o in zip(bucket.params, bucket.offsets))
    local = bucket.grad.clone()
    scale = bucket.all_reduce()
    mean = bucket.grad * scale
    gathered = dp.gather_rows(local.unsqueeze(0))
    if rank == 0:
        ret.put((same_values, in_bucket, bool(torch.allclose(mean, gathered.mean(0))), scale, bucket.numel))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_gradient_bucket_all_reduce_two_ranks():
    """The training exchange step: one all-reduce over the flat gradient bucket = mean of the per-rank gradients."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    same_values, in_bucket, ok, scale, numel = ret.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same_values and in_bucket and ok and scale == 0.5
    assert numel == 4 * 64                                                # every parameter padded to 64 elements (35, 5, 15, 3)


def _exchange_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from boficap_amd import dp
    from boficap_amd.trainer import FlatBucket
    torch.set_num_threads(1)
    dp.init_from_env("gloo")
    torch.manual_seed(0)

    class Net(torch.nn.Module):                                         # parameter names of the UIC model, dead copies included
        def __init__(self):
            super().__init__()
            mk = lambda *shape: torch.nn.Parameter(torch.randn(*shape))
            self.model = torch.nn.Module()
            self.model.length_predictor = torch.nn.Module()
            lp = self.model.length_predictor
            lp.length_attn = torch.nn.Module(); lp.length_attn.w = mk(300, 40)          # never read: no gradient on any rank
            lp.ff = torch.nn.Module(); lp.ff.w = mk(50, 7)
            lp.live = mk(129, 65)
            self.model.decoder = mk(9000, 3)
            self.att_embed = mk(70, 130)
    net = Net()
    bucket = FlatBucket(net)
    dead_last = bucket.names[-2:] == ["model.length_predictor.length_attn.w", "model.length_predictor.ff.w"]
    live = bucket.live_numel
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.zeros(bucket.numel)
    local[:live] = torch.randn(live, generator=g)
    bucket.grad.copy_(local)
    scale = bucket.all_reduce()                                          # the reference form: one collective over everything
    want = bucket.grad.clone() * scale
    out = {}
    for chunks, wire in ((1, None), (4, None), (7, None), (3, "bf16")):
        bucket.grad.copy_(local)
        seen = []
        for a, b, sc in bucket.exchange(None, chunks, wire):
            seen.append((a, b))
            bucket.grad[a:b] *= sc
        covered = sorted(seen)
        tiled = covered[0][0] == 0 and covered[-1][1] == live and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
        err = float((bucket.grad[:live] - want[:live]).abs().max())
        untouched = bool(torch.equal(bucket.grad[live:], local[live:]))
        ranks_agree = bool(torch.equal(dp.gather_rows(bucket.grad.unsqueeze(0))[0], dp.gather_rows(bucket.grad.unsqueeze(0))[1]))
        out[(chunks, wire)] = (len(seen), tiled, err, untouched, ranks_agree, seen[0][0] > seen[-1][0] if len(seen) > 1 else True)
    if rank == 0:
        ret.put((dead_last, live, bucket.numel, out, float(want.abs().max())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_chunked_and_bf16_exchange_equal_the_flat_all_reduce():
    """The step's exchange -- live prefix only, several collectives started last-chunk-first, optionally the bf16 mesh-direct
    form with float32 accumulation -- against the single flat all-reduce (2 gloo ranks)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    dead_last, live, numel, out, scale = ret.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert dead_last and live < numel and numel - live >= 300 * 40 + 50 * 7
    for (chunks, wire), (n, tiled, err, untouched, agree, reversed_order) in out.items():
        assert tiled and untouched and agree and reversed_order, (chunks, wire)
        assert n <= chunks
        assert err == 0.0 if wire is None else err < 2e-2 * scale, (chunks, wire, err)


def _two_phase_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import captioning.models as models
    from boficap_amd import dp
    from boficap_amd.config import TINY
    from boficap_amd.trainer import FlatBucket
    torch.set_num_threads(1)
    dp.init_from_env("gloo")
    torch.manual_seed(0)
    bucket = FlatBucket(models.setup(TINY.to_opt()))
    enc, live = bucket.encoder_end(), bucket.live_numel
    k = bucket.offsets.index(enc)
    boundary_ok = (all(n.startswith(("att_embed.", "model.encoder.")) for n in bucket.names[:k])
                   and not any(n.startswith(("att_embed.", "model.encoder.")) for n in bucket.names[k:]) and 0 < enc < live)
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.zeros(bucket.numel)
    local[:live] = torch.randn(live, generator=g)
    bucket.grad.copy_(local)
    bucket.all_reduce()
    want = bucket.grad.clone()
    # the overlapped step's order: the decoder side [enc, live) goes on the wire while the encoder's gradients are still being written ...
    bucket.grad.copy_(local)
    bucket.grad[:enc] = float("nan")                                      # ... so nothing of [0, enc) may be read by the first phase
    late = bucket.exchange_range(enc, live, None, 3)
    bucket.grad[:enc] = local[:enc]                                       # "the second stage of the backward" fills them in
    early = bucket.exchange_range(0, enc, None, 2)
    spans = []
    for a, b, w in late + early:
        w.wait()
        spans.append((a, b))
    cov = sorted(spans)
    tiled = cov[0][0] == 0 and cov[-1][1] == live and all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1))
    if rank == 0:
        ret.put((boundary_ok, tiled, bool(torch.equal(bucket.grad, want)), len(late), len(early), bucket.exchange_range(5, 5) == []))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_exchange_in_two_phases_around_the_encoder_boundary_equals_the_flat_all_reduce():
    """The exchange started inside backward (XETrainer._step_overlapped): [encoder_end, live) first, [0, encoder_end) after the second
    stage, each as its own asynchronous collectives, on the UIC model's own parameter bucket -- bit-equal to the flat all-reduce."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_two_phase_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    boundary_ok, tiled, equal, n_late, n_early, empty = ret.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert boundary_ok and tiled and equal and empty
    assert 1 <= n_late <= 3 and 1 <= n_early <= 2


def _bench_dry(n, *flags, env=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, BOFI_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1", **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), *flags], capture_output=True, text=True, timeout=600, cwd=root, env=e)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out.returncode, [json.loads(l) for l in lines], out.stderr


def test_bench_eight_ranks_walk_the_multi_gpu_path_without_a_gpu():
    """`python bench.py --gpus 8` as the driver starts it -- self-launch through torch.distributed.run, eight processes, gloo -- with the GPU work of
    every rank replaced by a sleep (BOFI_BENCH_REHEARSAL=dry): the ranks the collective backend saw, the per-rank shard sizes, value = ALL ranks' images over the SLOWEST
    rank's time, exactly one line from rank 0; and a rank that raises before the exchange ends the job with a non-zero status and no line.  (The N = 2 walk WITH the GPU
    work is tests/test_gpu_bench.py; a one-GPU box cannot hold eight ranks.)  Replaces nn.DataParallel's scatter / gather, tools/train.py:97-101."""
    rc, lines, err = _bench_dry(8, "--steps", "20", "--warmup", "5")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = lines[0]
    c = d["config"]
    assert d["n_gpus"] == 8 and c["rccl_ranks"] == 8 and c["dist_backend"] == "gloo" and "dry" in c["rehearsal"] and d["data"] == "none"
    assert c["shard_images_per_rank"] == [64 * 20] * 8                             # images by rank, equal shards, every image once
    assert abs(d["value"] - 8 * 64 * 20 / (d["ms_per_step"] * 20e-3)) <= 0.01 * d["value"]       # whole-job aggregate over the measured (max over ranks) time
    assert d["ms_per_step"] >= c["slowest_rank_ms_per_step"] * 0.99                 # ... which is the SLOWEST rank's (rank 7 sleeps 2.75 ms per step, rank 0 one)
    rc, lines, err = _bench_dry(8, "--steps", "20", "--warmup", "5", env={"BOFI_BENCH_DRY_FAIL_RANK": "5"})
    assert rc != 0 and not lines and "injected failure" in err
    rc, lines, err = _bench_dry(1, "--steps", "4", "--warmup", "1")
    assert rc == 0 and lines[0]["n_gpus"] == 1 and lines[0]["config"]["shard_images_per_rank"] == [256]
