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


@pytest.mark.parametrize("R,form", [(36, "<5>: 4 batches of 9 region rows"), (64, "<8>: up to 64 regions"), (100, "<0>: any count, online softmax")])
def test_loop_kernel_on_the_oracles_own_trajectory(R, form, weight_cache, monkeypatch):
    """VERDICT r5 item 3 (weak 1): every instantiation of bound_loop_kernel against the float32 ORACLE -- not against its own stage form -- at the full width
    (d_ff 2048) and on layouts of LATER iterations, where the (position, label) score / value tables of the row-0 self-attention really run
    (bound_loop.hip S1; TransformerModel.py:357-383 on the state TransformerModel.py:1843-1869 left): O.core_naic's trace gives (ext_syn, last) at the start of
    iterations 0, 2, 5 and 8 and the log-probs the reference computes there; the engine's bounding step on that state is held to the live-class bars of
    test_bf16_logits_within_tolerance_on_every_image.  Then the free bf16 decode against O.sample_naic(fix_q1=True): layouts within the flip bar, the filling
    pass teacher-forced on the oracle's layout within north_star's 2e-2.  Ragged region counts (one image with a single region) throughout."""
    import boficap_oracle as O
    from conftest import record_parity
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    H.lib().bofi_reload_env()
    B = 32
    cfg, sd, eng = _engine(weight_cache, B, R)
    assert eng.bound_loop_active(R)
    w = O.as_torch(sd)
    att_np = W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=77 + R)
    rng = np.random.default_rng(R)
    lens = rng.integers(max(2, R // 2), R + 1, size=B).astype(np.int32)
    lens[0], lens[5], lens[B - 1] = R, 1, R - 3
    masks = np.zeros((B, R), np.float32)
    for b, n in enumerate(lens):
        masks[b, :n] = 1
        att_np[b, n:] = 5.0                                   # (garbage in the padding must not leak)
    att, am = torch.from_numpy(att_np), torch.from_numpy(masks)
    trace = []
    with torch.no_grad():
        memory, src_mask = O.memory_of(w, cfg, att, am)
        phrase, opn, opl, ops, dg = O.core_naic(w, cfg, memory, src_mask, fix_q1=True, trace=trace)
        olp = torch.log_softmax(O.logit(w, phrase), dim=2)      # (AttModel.py:203-210)
    assert len(trace) >= 6, len(trace)
    att_len = torch.from_numpy(lens).cuda()
    mem_e = eng.encode(att.cuda().to(torch.bfloat16), att_len).cpu()
    live_len, live_syn = [0, 1, 2, 3, 4, 9], [1, 4, 5, 6]
    L = cfg.seq_length + 2
    worst = {"total": [0.0, 0.0], "encoder": [0.0, 0.0], "own": [0.0, 0.0]}
    for k in [k for k in (0, 2, 5, 8) if k < len(trace)]:
        t = trace[k]
        rows = t["active"]                                     # (a finished image's state is frozen: its log-probs are computed and ignored, TM:1844-1845)
        assert int(rows.sum()) > 0
        llp, slp = eng.bound_step(t["ext_syn"].to(torch.int32).cuda(), t["last"].to(torch.int32).cuda(), R, att_len)
        tm = torch.zeros(B, L, L, dtype=torch.bool); tm[:, :, 0] = True
        for b in range(B):
            tm[b, 0, :int(t["last"][b])] = True                # row 0 sees the keys laid out so far (TM:1859-1867; only row 0 is read)
        with torch.no_grad():                                  # the ENGINE's memory through the float32 bounding layer + heads on this layout: the encoder's share
            _, c_llp, _, c_slp = O.bound_step_na(w, cfg, t["ext_syn"], mem_e, src_mask, tm)
        pick = lambda d, cls: float(d[rows][:, cls].abs().max())
        tot = [pick(llp.cpu() - t["len_logp"], live_len), pick(slp.cpu() - t["syn_logp"], live_syn)]
        enc = [pick(c_llp - t["len_logp"], live_len), pick(c_slp - t["syn_logp"], live_syn)]
        own = [pick(llp.cpu() - c_llp, live_len), pick(slp.cpu() - c_slp, live_syn)]
        print(f"R {R} {form}: iteration {k}, {int(rows.sum())} live images, layouts up to {int(t['last'][rows].max())} positions: |dlogp| live classes (len, syn): "
              f"total {tot[0]:.2e} {tot[1]:.2e}; encoder's memory alone {enc[0]:.2e} {enc[1]:.2e}; the loop kernel's own {own[0]:.2e} {own[1]:.2e}")
        for name, v in (("total", tot), ("encoder", enc), ("own", own)):
            worst[name] = [max(worst[name][0], v[0]), max(worst[name][1], v[1])]
    # bars = 1.3 x the measurement (profiles/r06_parity_errors.json): the kernel's OWN share (engine vs the float32 chain on the engine's memory) is 0.017-0.024 in all three
    # forms (the every-image test's 64 full images: 0.015-0.017) -- the 64-region and any-count forms are as good as the 36-region one; the encoder's share (bf16 operands in
    # front of heads whose gain is 12.7 on a span of 16.4) grows with ragged and longer region lists: 0.029 at 36, 0.036 at 64, 0.041 at 100 regions,
    # the total 0.029 / 0.039 / 0.051 -- north_star's 2e-2 on THESE heads is not met by a bf16-operand encoder (stated in DESIGN.md sections 0 and 2)
    own_bar, tot_bar = 0.031, 0.067
    for name, bar in (("total", tot_bar), ("encoder", tot_bar), ("own", own_bar)):
        record_parity(f"bf16_loop_kernel_trajectory_{name}_len_R{R}", worst[name][0], bar, f"bound_loop_kernel{form}: iterations 0/2/5/8 of the oracle's trajectory, live classes, {B} ragged images")
        record_parity(f"bf16_loop_kernel_trajectory_{name}_syn_R{R}", worst[name][1], bar, "as above, label head")
    assert max(worst["own"]) < own_bar and max(worst["total"]) < tot_bar and max(worst["encoder"]) < tot_bar, worst
    free = eng.decode_naic(att.cuda().to(torch.bfloat16), att_len, strict_q1=False)
    flips = int(((free["phrase_length"].cpu() != opl).any(1) | (free["phrase_syn"].cpu() != ops).any(1)).sum())
    _, lp = eng.fill_naic(dg["ext_syn"].to(torch.int32).cuda(), dg["last"].to(torch.int32).cuda(), R, att_len, strict_q1=False)
    lp = lp.cpu()
    assert torch.equal(lp.isnan(), olp.isnan())
    e_fill = float((lp - olp).nan_to_num().abs().max())
    print(f"R {R}: free decode {flips}/{B} layouts differ from the float32 oracle's; teacher-forced fill |dlogp| {e_fill:.2e}")
    record_parity(f"bf16_loop_kernel_free_decode_flips_R{R}", flips, 3, f"images of {B} whose slot layout differs from the float32 oracle's")
    record_parity(f"bf16_fill_teacher_forced_R{R}", e_fill, 2e-2, "ragged regions, all images x positions x V")
    assert flips <= 3 and e_fill < 2e-2
    monkeypatch.undo()
    H.lib().bofi_reload_env()


def test_fp16_saturation_is_reported_and_falls_back(weight_cache, monkeypatch):
    """the loop kernel clamps its fp16 operands to +-65 504 -- and now SAYS so.  (1) the benchmark model: the word stays 0.  (2) a bounding layer whose hidden rows leave
    fp16's range (w_1 of the bounding layer scaled; its fp16 weight copies still fit): bit 0 of the decode's status word; decode_naic_checked warns and returns exactly what the five-launch
    bf16 iterations give (the form bf16's float32 exponent range serves), the engine back on its default afterwards; the same through DecodePipeline.  (3) a weight that does not fit
    fp16 itself: bit 1 (pack time), the fallback is kept until the weights change.  TransformerModel.py:357-383 runs in float32 in the reference: this guards the library's own choice."""
    import warnings
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine, DecodePipeline
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    H.lib().bofi_reload_env()
    B, R = 32, 36
    cfg, sd, eng = _engine(weight_cache, B, R)
    att = torch.from_numpy(W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=3)).cuda().to(torch.bfloat16)
    r = eng.decode_naic(att, strict_q1=False)
    assert eng.bound_loop_active(R) and eng.saturated(r) == 0

    w1 = "model.length_predictor.LengthPredictor.0.ff.w_1.weight"
    hot = dict(sd); hot[w1] = sd[w1] * 3.0e4                   # |w| ~ 0.05 * 3e4 = 1.5e3 fits fp16; the hidden rows (sums of 512 such products) do not
    assert float(np.abs(hot[w1]).max()) < 6.0e4
    e2 = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
    e2.load_state_dict(hot)
    r2 = e2.decode_naic(att, strict_q1=False)
    assert e2.bound_loop_active(R) and e2.saturated(r2) == 1
    keep = {k: r2[k].clone() for k in ("seq", "phrase_num", "phrase_length", "phrase_syn")}
    e2.set_bound_loop(0)
    assert not e2.bound_loop_active(R)
    ref = {k: v.clone() for k, v in e2.decode_naic(att, strict_q1=False).items() if torch.is_tensor(v)}
    assert e2.saturated(ref) == 0                               # (the five-launch iterations never set it)
    e2.set_bound_loop(-1)
    with pytest.warns(RuntimeWarning, match="clamped an activation"):
        r3 = e2.decode_naic_checked(att, strict_q1=False)
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
        assert torch.equal(r3[k], ref[k]), k
    assert torch.equal(r3["seq_logprob"].nan_to_num(), ref["seq_logprob"].nan_to_num())
    assert e2.saturated(r3) == 0 and e2.bound_loop_active(R)    # this decode was clean; the engine is back on its default form
    differs = any(not torch.equal(keep[k], ref[k]) for k in keep)
    print(f"hidden rows beyond fp16: the clamped loop kernel's layouts {'differ from' if differs else 'happen to equal'} the bf16 iterations'")
    # the same through the pipeline (tools/eval.py, decode_many)
    pipe = DecodePipeline(e2, in_flight=2, batches_per_launch=2, strict_q1=False, stats=False)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = list(pipe.run([att[:16].float().cpu(), att[16:].float().cpu()]))
    assert any("clamped" in str(c.message) for c in caught)
    e2.set_bound_loop(0)
    for i, sl in enumerate((slice(0, 16), slice(16, 32))):
        one = e2.decode_naic(att[sl].contiguous(), strict_q1=False)
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(got[i][k], one[k].cpu()), (i, k)

    big = dict(sd); big[w1] = sd[w1].copy(); big[w1][7, 11] = 1.0e5          # one weight beyond fp16: clamped when the fp16 copies are packed
    e3 = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
    e3.load_state_dict(big)
    assert e3.saturated(e3.decode_naic(att, strict_q1=False)) & 2
    with pytest.warns(RuntimeWarning, match="weight copies"):
        e3.decode_naic_checked(att, strict_q1=False)
    assert not e3.bound_loop_active(R)                          # stays on the bf16 iterations ...
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert e3.saturated(e3.decode_naic_checked(att, strict_q1=False)) == 0
    e3.load_state_dict(sd)                                      # ... until the weights change
    assert e3.bound_loop_active(R) and e3.saturated(e3.decode_naic(att, strict_q1=False)) == 0
    monkeypatch.undo()
    H.lib().bofi_reload_env()


@pytest.mark.parametrize("R", [36, 64, 100])
def test_pair_of_workgroups_per_group_is_bit_equal_to_one(R, weight_cache, monkeypatch):
    """Round 6: two workgroups per group of 16 images share the feed-forward's weight stream -- each runs one half of the hidden units and the two partial
    sums of y3 meet once per iteration through memory (BOFI_BL_PAIR: by default for launches of at most 384 images, 2 = always, 0 = never).  The first workgroup to arrive never waits for one that is not running (it goes on alone
    when the partner has not arrived by its first feed-forward stage), and a workgroup that runs alone forms the SAME two partial sums in the same order: whatever the
    dispatcher does, the decode's bits do not depend on it.  Shown: (1) pairs really form (the diagnostic counter), (2) BOFI_BL_PAIR=2 and =0 give identical slot layouts,
    ids, iteration counts and log-probs, eager and as a replayed graph, ragged regions and a batch that is no multiple of 16, (3) several decodes in flight on forks."""
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    B = 150                                                     # (R: every instantiation of the kernel -- <= 36, <= 64, any count)
    cfg, sd, eng = _engine(weight_cache, B, R)
    att = torch.from_numpy(W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=31)).cuda().to(torch.bfloat16)
    lens = torch.full((B,), R, dtype=torch.int32); lens[3], lens[40], lens[149] = 11, 1, R - 6
    lens = lens.cuda()

    def pairs_formed(e):
        buf = torch.zeros(8, dtype=torch.int32, device="cuda")
        H.check(H.lib().bofi_engine_debug_copy(e._h, b"counters", H.ptr(buf), 32, H.stream_ptr()), "debug_copy")
        torch.cuda.synchronize()
        return int(buf[5])

    runs = {}
    for pair in ("2", "0"):
        monkeypatch.setenv("BOFI_BL_PAIR", pair)
        H.lib().bofi_reload_env()
        outs, prev, n_pairs = [], None, 0
        groups = (B + 15) // 16
        for graph in (False, True, True):
            r = eng.decode_naic(att, lens, strict_q1=False, graph=graph, out=prev if graph else None)      # (the second graph call replays the captured launch)
            torch.cuda.synchronize()
            assert eng.saturated(r) == 0
            prev = r if graph else None
            outs.append({k: v.clone() for k, v in r.items() if torch.is_tensor(v)})
            formed = pairs_formed(eng)                           # (of THIS decode: the counters are zeroed with the slot state)
            assert formed <= groups
            n_pairs += formed
        # (whether a pair forms is the dispatcher's business -- a second workgroup that arrives late finds its group taken and leaves --: over three decodes some do)
        assert (n_pairs > 0) == (pair == "2"), (pair, n_pairs)
        print(f"BOFI_BL_PAIR={pair}: {n_pairs} of {3 * groups} groups (three decodes) ran as a pair of workgroups")
        for o in outs[1:]:
            for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
                assert torch.equal(o[k], outs[0][k]), (pair, k)
        runs[pair] = outs[0]
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
        assert torch.equal(runs["2"][k], runs["0"][k]), k
    assert torch.equal(runs["2"]["seq_logprob"].nan_to_num(), runs["0"]["seq_logprob"].nan_to_num())
    # forks in flight: every fork owns its exchange buffers
    monkeypatch.setenv("BOFI_BL_PAIR", "2")
    H.lib().bofi_reload_env()
    forks = [eng.fork() for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in forks]
    res = []
    torch.cuda.synchronize()
    for f, s_ in zip(forks, streams):
        with torch.cuda.stream(s_):
            res.append(f.decode_naic(att, lens, strict_q1=False))
    torch.cuda.synchronize()
    for r in res:
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(r[k], runs["0"][k]), k
    monkeypatch.undo()
    H.lib().bofi_reload_env()
