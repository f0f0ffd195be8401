"""The persistent bounding-loop kernel (boficap_amd/csrc/bound_loop.hip; core_NAIC's loop TransformerModel.py:1833-1869 as ONE launch, one
workgroup per 16 images) at the full size: its in-kernel slot bookkeeping, early exit and iteration count against a host replay of the
reference's bookkeeping driven by the kernel's own stage form; independence of what else is in the launch; ragged and empty region lists,
region counts above 36 (the 64-key instantiation), batches that are no multiple of 16; the kernel choice by hint and environment."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(weight_cache, B, R):
    from boficap_amd.engine import BofiEngine
    cfg, sd = weight_cache("FULL", 0, 1.0)
    eng = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
    eng.load_state_dict(sd)
    return cfg, sd, eng


def _first_nan_else_first_max(lp):
    """torch.max on CPU (TransformerModel.py:380-383): the first NaN wins, else the first maximum."""
    out = []
    for row in lp:
        nan = np.isnan(row)
        out.append(int(np.argmax(nan)) if nan.any() else int(np.argmax(row)))
    return np.array(out)


def _host_replay(eng, cfg, B, R, att_len):
    """core_NAIC's bookkeeping (TransformerModel.py:1843-1869) on the host, every iteration's log-probs from the kernel's stage form
    (bofi_engine_bound_step: one iteration on a given layout, no update)."""
    S, L = cfg.seq_length, cfg.seq_length + 2
    ext = np.zeros((B, L), np.int32); ext[:, 0] = cfg.len_idx
    last = np.ones(B, np.int32)
    fin = np.zeros(B, bool)
    pn = np.zeros(B, np.int32)
    plen = np.zeros((B, S), np.int32)
    psyn = np.zeros((B, S), np.int64)
    iters = 0
    for it in range(S):
        if fin.all():
            break
        iters += 1
        llp, slp = eng.bound_step(torch.from_numpy(ext).cuda(), torch.from_numpy(last).cuda(), R, att_len)
        ln_all, sn_all = _first_nan_else_first_max(llp.cpu().numpy()), _first_nan_else_first_max(slp.cpu().numpy())
        for b in range(B):
            if fin[b]:
                continue
            ln, sn, la = int(ln_all[b]), int(sn_all[b]), int(last[b])
            if ln == 0 or sn < 4 or sn > 6:
                fin[b] = True
                continue
            if ln + la >= S + 1:
                ln = S + 1 - la
                fin[b] = True
            plen[b, pn[b]], psyn[b, pn[b]] = ln, sn
            pn[b] += 1
            ext[b, la:la + ln] = sn
            last[b] = la + ln
    return pn, plen, psyn, iters


@pytest.mark.parametrize("B,R,ragged", [(37, 36, True), (64, 36, False), (21, 50, True), (19, 100, True), (5, 67, False)])
def test_loop_kernel_bookkeeping_equals_a_host_replay(B, R, ragged, weight_cache, monkeypatch):
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    H.lib().bofi_reload_env()
    cfg, sd, eng = _engine(weight_cache, B, R)
    assert eng.bound_loop_active(R)
    att_np = W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=31 + B)
    att_len = None
    if ragged:
        rng = np.random.default_rng(B)
        lens = rng.integers(R // 3, R + 1, B).astype(np.int32)
        lens[3] = 0                                           # an image without regions: NaN through the cross-attention, the first NaN picks class 0 -> EOS at once
        lens[0] = R
        for b in range(B):
            att_np[b, lens[b]:] = 0
        att_len = torch.from_numpy(lens).cuda()
    att = torch.from_numpy(att_np).cuda().to(torch.bfloat16)
    out = eng.decode_naic(att, att_len, strict_q1=False)
    torch.cuda.synchronize()
    pn, plen, psyn, iters = _host_replay(eng, cfg, B, R, att_len)       # (the decode's encode left its K|V in the engine's workspace: the stage form reads them)
    assert (out["phrase_num"].cpu().numpy() == pn).all()
    assert (out["phrase_length"].cpu().numpy() == plen).all() and (out["phrase_syn"].cpu().numpy() == psyn).all()
    assert int(out["bound_iters"]) == iters
    if ragged:
        assert pn[3] == 0                                     # the empty image ended at once
    assert pn.max() >= 4 and iters >= 5, (pn, iters)          # the loop really ran
    # the same images in another order and another batch size: an image's layout does not depend on what else is in the launch
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))[: max(1, B - 5)]
    out2 = eng.decode_naic(att[perm].contiguous(), None if att_len is None else att_len[perm.cuda()].contiguous(), strict_q1=False)
    assert torch.equal(out2["phrase_length"].cpu(), out["phrase_length"].cpu()[perm]) and torch.equal(out2["phrase_syn"].cpu(), out["phrase_syn"].cpu()[perm])
    # graph replay = eager
    out3 = eng.decode_naic(att, att_len, strict_q1=False, graph=True)
    out3 = eng.decode_naic(att, att_len, strict_q1=False, graph=True, out=out3)
    torch.cuda.synchronize()
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
        assert torch.equal(out3[k], out[k]), k
    assert torch.equal(out3["seq_logprob"].isnan(), out["seq_logprob"].isnan()) and torch.equal(out3["seq_logprob"].nan_to_num(), out["seq_logprob"].nan_to_num())
    monkeypatch.undo()
    H.lib().bofi_reload_env()


def test_loop_kernel_against_the_five_launch_chain_and_the_oracle(weight_cache, monkeypatch):
    """Both forms of the bounding loop decode the same 64 images: each within its bar of the float32 oracle's layouts, and they agree with each
    other wherever the oracle's decision margins are not razor-thin (fp16 against bf16 operands: different roundings, one algorithm)."""
    import boficap_oracle as O
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    cfg, sd, eng = _engine(weight_cache, 64, 36)
    att_np = W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=99)
    att = torch.from_numpy(att_np).cuda().to(torch.bfloat16)
    with torch.no_grad():
        _, _, opn, opl, ops, _ = O.sample_naic(O.as_torch(sd), cfg, torch.from_numpy(att_np), fix_q1=True)
    res = {}
    for knob in ("2", "0"):
        monkeypatch.setenv("BOFI_BOUND_LOOP", knob)
        H.lib().bofi_reload_env()
        assert eng.bound_loop_active(36) == (knob == "2")
        r = eng.decode_naic(att, strict_q1=False)
        res[knob] = (r["phrase_length"].cpu(), r["phrase_syn"].cpu(), int(r["bound_iters"]))
    flips = {k: int(((v[0] != opl).any(1) | (v[1] != ops).any(1)).sum()) for k, v in res.items()}
    between = int(((res["2"][0] != res["0"][0]).any(1) | (res["2"][1] != res["0"][1]).any(1)).sum())
    print(f"layouts differing from the float32 oracle's: loop kernel {flips['2']}/64, five-launch chain {flips['0']}/64; between the two {between}/64")
    assert flips["2"] <= 5 and flips["0"] <= 19 and between <= flips["2"] + flips["0"]
    monkeypatch.undo()
    H.lib().bofi_reload_env()


def test_kernel_choice_follows_hint_and_knob(weight_cache, monkeypatch):
    from boficap_amd import hip as H
    from boficap_amd.engine import BofiEngine
    cfg, sd, eng = _engine(weight_cache, 16, 36)
    assert eng.bound_loop_active(36) and eng.bound_loop_active(100) and not eng.bound_loop_active(129)
    eng.set_decodes_in_flight(1)                               # a decode that runs alone keeps the five launches per iteration (shorter chain)
    assert not eng.bound_loop_active(36)
    eng.set_decodes_in_flight(4)
    assert eng.bound_loop_active(36)
    monkeypatch.setenv("BOFI_BOUND_LOOP", "0")
    H.lib().bofi_reload_env()
    assert not eng.bound_loop_active(36)
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    H.lib().bofi_reload_env()
    eng.set_decodes_in_flight(1)
    assert eng.bound_loop_active(36)
    monkeypatch.undo()
    H.lib().bofi_reload_env()
    f32 = BofiEngine(cfg, torch.float32, max_batch=4, max_regions=36)
    f32.load_state_dict(sd)
    assert not f32.bound_loop_active(36)                       # the float32 engine keeps its float32 kernels (bit-exact ids against the reference)


def test_refresh_from_device_rebuilds_the_loop_kernels_operands(weight_cache):
    """bofi_engine_refresh_device derives the fp16 copies and the float32 tables on the device from the caller's tensors: after a refresh with OTHER weights the
    bound step agrees with that of an engine finalized with those weights."""
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    cfg, sd_a = weight_cache("FULL", 0, 1.0)
    sd_b = W.make_state_dict(cfg, seed=3, bound_preset=False)      # (other weights: no calibrated heads needed, the stage form evaluates a given layout)
    att = torch.from_numpy(W.synthetic_att_feats(20, 36, cfg.att_feat_size, seed=5)).cuda().to(torch.bfloat16)
    ext = torch.zeros(20, cfg.seq_length + 2, dtype=torch.int32, device="cuda"); ext[:, 0] = cfg.len_idx
    ext[:, 1:4] = 5
    last = torch.full((20,), 4, dtype=torch.int32, device="cuda")
    outs = []
    for how in ("finalize", "refresh"):
        eng = BofiEngine(cfg, torch.bfloat16, max_batch=20, max_regions=36)
        eng.load_state_dict(sd_b if how == "finalize" else sd_a)
        if how == "refresh":
            eng.refresh_from_device({k: torch.from_numpy(v).cuda() for k, v in sd_b.items()})
        assert eng.bound_loop_active(36)
        eng.encode(att)
        llp, slp = eng.bound_step(ext, last, 36)
        outs.append((llp.cpu(), slp.cpu()))
    # (the two paths fold the LayerNorms with sums of different order: a few folded biases / bf16 weights of the ENCODER differ in their last bit, as in
    # test_device_side_weight_refresh_equals_a_fresh_load; stale operands would be off by whole units)
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 5e-3 and float((outs[0][1] - outs[1][1]).abs().max()) < 5e-3


def test_decode_in_phases_equals_the_whole_decode(weight_cache):
    """BOFI_FLAG_PHASE_ENCODE / _BOUND / _FILL: the three parts of a decode as separate calls on one engine (a pipelining caller's form) give the whole
    decode's outputs bit for bit, eager and replayed from their own graphs."""
    from boficap_amd import weights as W
    cfg, sd, eng = _engine(weight_cache, 48, 36)
    att = torch.from_numpy(W.synthetic_att_feats(48, 36, cfg.att_feat_size, seed=3)).cuda().to(torch.bfloat16)
    whole = eng.decode_naic(att, q1_group=16)
    ref = {k: v.clone() for k, v in whole.items() if torch.is_tensor(v)}
    for graph in (False, True, True):
        out = {k: (torch.zeros_like(v) if torch.is_tensor(v) else v) for k, v in whole.items()}
        for ph in ("e", "b", "f"):
            eng.decode_naic(att, q1_group=16, out=out, phases=ph, graph=graph)
        torch.cuda.synchronize()
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
            assert torch.equal(out[k], ref[k]), (graph, k)
        assert torch.equal(out["seq_logprob"].nan_to_num(), ref["seq_logprob"].nan_to_num())
    with pytest.raises(Exception):
        eng.decode_naic(att, phases="x")


@pytest.mark.parametrize("d_ff", [512, 1024])
def test_loop_kernel_with_a_narrower_feed_forward(d_ff, weight_cache):
    """The loop kernel serves d_ff = 512 ... 2048 in steps of 512 (chunks of w_1 per wavefront, K segments of w_2): a model with a narrower
    feed-forward layer everywhere, its first bounding step against the float32 oracle."""
    import dataclasses
    import boficap_oracle as O
    from boficap_amd import weights as W
    from boficap_amd.config import FULL
    from boficap_amd.engine import BofiEngine
    cfg = dataclasses.replace(FULL, d_ff=d_ff, vocab_size=996)
    sd = W.make_state_dict(cfg, seed=1, bound_preset=False)
    B = 19
    att_np = W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=8)
    w = O.as_torch(sd)
    L = cfg.seq_length + 2
    ext = torch.zeros(B, L, dtype=torch.long); ext[:, 0] = cfg.len_idx
    ext[:, 1:3] = 5; ext[:, 3:6] = 4
    last = 6
    tm = torch.zeros(B, L, L, dtype=torch.bool); tm[:, :, 0] = True
    tm[:, 0, :last] = True                                      # row 0 sees the keys laid out so far (TransformerModel.py:1859-1867; only row 0 is read)
    with torch.no_grad():
        memory, src_mask = O.memory_of(w, cfg, torch.from_numpy(att_np))
        _, o_llp, _, o_slp = O.bound_step_na(w, cfg, ext, memory, src_mask, tm)
    eng = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=36)
    eng.load_state_dict(sd)
    assert eng.bound_loop_active(36)
    eng.encode(torch.from_numpy(att_np).cuda().to(torch.bfloat16))
    llp, slp = eng.bound_step(ext.to(torch.int32).cuda(), torch.full((B,), last, dtype=torch.int32, device="cuda"), 36)
    e_len, e_syn = float((llp.cpu() - o_llp).abs().max()), float((slp.cpu() - o_slp).abs().max())
    print(f"d_ff {d_ff}: loop kernel vs the float32 oracle, bound step on a three-slot layout: |dlogp| len {e_len:.2e} syn {e_syn:.2e}")
    assert e_len < 5e-2 and e_syn < 5e-2, (e_len, e_syn)     # (uncalibrated heads: all 20 / 10 classes live; bf16 encoder in front)
