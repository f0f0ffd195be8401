"""Two data-parallel ranks on ONE MI355X (process group: gloo over device tensors -- RCCL refuses two ranks on one device, and the
GPU box has a single card): XETrainer.step with the chunked gradient exchange and the per-chunk optimiser, eager and as a captured
step graph.  The ranks' parameters must stay bit-equal, and equal -- to float32 summation order -- those of ONE process stepping on
the concatenated batch (the mean of the per-rank gradients is the reference's loss.mean() over DataParallel replicas,
tools/train.py:217; with equal token counts per shard that is the gradient of the concatenated batch)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_IMG, SPI, STEPS = 2, 3, 3


def _setup(train_dtype):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import captioning.models as models
    from boficap_amd import weights as W
    from boficap_amd.config import TINY
    cfg = TINY
    sd = W.make_state_dict(cfg, 0, gen_scale=6.0)
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.cuda().eval()                                # eval: no dropout, so sharded and concatenated steps see the same function
    model.train_dtype = train_dtype
    model.opt.noamopt_warmup = 10                              # learning rates large enough to move float32 weights visibly
    return cfg, model


def _shard(cfg, rank, step):
    """Rank `rank`'s batch of step `step`: the SAME captions on every rank (equal token counts), its own region features."""
    from boficap_amd.collate import synthetic_training_batch
    from boficap_amd.weights import synthetic_att_feats
    hb = synthetic_training_batch(cfg, N_IMG, SPI, seed=40 + step)
    att = synthetic_att_feats(N_IMG, 36, cfg.att_feat_size, seed=1000 * rank + step)
    return hb, att


def _run_steps(cfg, model, graph, ranks, drop_worst=False):
    from boficap_amd.trainer import XETrainer
    tr = XETrainer(model, graph=graph)
    for step in range(STEPS):
        parts = [_shard(cfg, r, step) for r in ranks]
        hb = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
        batch = {k: torch.from_numpy(v).cuda() for k, v in hb.items()}
        batch["att_feats"] = torch.from_numpy(np.concatenate([p[1] for p in parts])).cuda()
        batch["max_phrase_num"] = int(hb["phrase_num"].max())
        tr.step(tr.add_token_rows(batch, hb), drop_worst=drop_worst)
    torch.cuda.synchronize()
    return tr.bucket.flat[:tr.bucket.live_numel].detach().cpu()


def _worker(rank, world, port, ret, graph, wire, overlap=True, dtype="float32", drop_worst=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    from boficap_amd import dp
    torch.cuda.set_device(0)
    dp.init_from_env("gloo")
    cfg, model = _setup(getattr(torch, dtype))
    if wire:
        model.opt.bofi_dp_wire = wire
    model.opt.bofi_dp_overlap = overlap
    flat = _run_steps(cfg, model, graph, [rank], drop_worst)
    ret.put((rank, flat.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_on_one_gpu_equal_one_process_on_the_concatenated_batch(graph):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")                              # fresh children: nothing of this process's HIP state is inherited
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, graph, None)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1]), "the ranks' parameters diverged"
    cfg, model = _setup(torch.float32)
    from boficap_amd.trainer import XETrainer                 # (the live prefix of a fresh bucket = the initial weights)
    before = XETrainer(model).bucket
    init = before.flat[:before.live_numel].detach().cpu().numpy().copy()
    cfg, model = _setup(torch.float32)
    one = _run_steps(cfg, model, graph, [0, 1]).numpy()
    _assert_same_training(got[0], one, init)


def _assert_same_training(got, one, init):
    step_size = np.abs(one - init)
    moved = step_size.max()
    assert moved > 1e-4, "the optimiser did not move the weights"
    # Adam normalises every gradient by its own magnitude: an element whose gradient is exactly zero in exact arithmetic (the key
    # biases: softmax is shift-invariant) moves by +-lr on summation noise alone, with whichever sign the order of the additions
    # produced.  So: nearly all elements agree tightly, the disagreeing rest is a sliver, and nothing differs by more than the
    # two sign choices can explain.
    d = np.abs(got - one)
    assert float((d > 2e-3 * moved).mean()) < 5e-3, (float((d > 2e-3 * moved).mean()), float(d.max()), float(moved))
    assert float(d.mean()) < 2e-3 * float(step_size.mean()), (float(d.mean()), float(step_size.mean()))
    assert float(d.max()) <= 2.2 * moved


def _two_ranks(graph, overlap, dtype):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, graph, None, overlap, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1]), "the ranks' parameters diverged"
    return got[0]


@pytest.mark.parametrize("graph,dtype", [(False, "float32"), (True, "float32"), (True, "bfloat16")])
def test_exchange_started_inside_backward_equals_the_plain_step(graph, dtype):
    """bofi_dp_overlap: backward in two stages around the encoder's output (two captured graphs in graph mode), the decoder-side
    gradients all-reduced while the encoder's backward runs, Adam per chunk -- against the one-backward step with the exchange behind
    it, after three steps.  The same function and the same collectives' sums; not bit for bit, because the weight-gradient GEMMs
    combine their row splits with float32 atomics (order not fixed from run to run) and the grouped launch picks its split count
    from the tiles in the group, of which there are now two: the bars are those of the two-ranks-against-one-process test."""
    cfg, model = _setup(getattr(torch, dtype))
    from boficap_amd.trainer import XETrainer
    b = XETrainer(model).bucket
    init = b.flat[:b.live_numel].detach().cpu().numpy().copy()
    del b, model
    _assert_same_training(_two_ranks(graph, True, dtype), _two_ranks(graph, False, dtype), init)


def test_drop_worst_under_data_parallel_is_the_top_k_of_the_whole_batch():
    """tools/train.py:216-220 gathers the replicas' per-caption losses and takes ONE top-k over the whole batch: two ranks (the threshold
    from the all-gathered losses, each rank's kept captions, the loss scaled for the rank average) against one process on the concatenated
    batch -- 12 captions, the 9 best kept, whichever rank they sit on."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, False, None, True, "float32", True)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1]), "the ranks' parameters diverged"
    cfg, model = _setup(torch.float32)
    from boficap_amd.trainer import XETrainer
    before = XETrainer(model).bucket
    init = before.flat[:before.live_numel].detach().cpu().numpy().copy()
    cfg, model = _setup(torch.float32)
    one = _run_steps(cfg, model, False, [0, 1], drop_worst=True).numpy()
    _assert_same_training(got[0], one, init)
