"""Self-critical path on the MI355X: sampling inside the semi-autoregressive loop, the differentiable re-forward of sampled
captions (the engine's own distributions must come back), the new_self_critical loss and its gradients against autograd
over the oracle, and a full rl_step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import boficap_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _model(weight_cache, manifest, case="tiny_saic_multi"):
    """Weights of the tiny_saic_multi fixture (boficap_amd.weights.with_len_row_shared): the semi-autoregressive mode lays
    out and fills several phrases with them -- pinned to the real reference by tests/golden/tiny_saic_multi.npz."""
    import captioning.models as models
    m = manifest[case]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return cfg, sd, model.cuda().eval()


def _images():
    return torch.from_numpy(load_golden("tiny_saic_multi")["att_feats"])


def _sample(model, att, mode, n, T=1.0):
    fc = torch.zeros(att.size(0), 0, device="cuda")
    with torch.no_grad():
        r = model(fc, att, None, opt={"train_mode": mode, "sample_method": "sample", "sample_n": n, "temperature": T}, mode="sample")
    return dict(zip(("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn"), r[:5]))


def test_saic_sampling_low_temperature_is_greedy(weight_cache, manifest):
    cfg, sd, model = _model(weight_cache, manifest)
    att = _images().cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    with torch.no_grad():
        greedy = model(fc, att, None, opt={"train_mode": "SAIC", "sample_method": "greedy"}, mode="sample")
    assert int((greedy[0] > 0).sum()) > att.size(0)                      # the mode really emits captions with these weights
    cold = _sample(model, att, "SAIC", 1, T=1e-3)
    assert torch.equal(cold["seq"], greedy[0]) and torch.equal(cold["phrase_length"], greedy[3])
    hot = _sample(model, att, "SAIC", 1, T=2.0)
    assert not torch.equal(hot["seq"], greedy[0])


@pytest.mark.parametrize("mode", ["SAIC", "NAIC"])
def test_reforward_of_sampled_captions_returns_the_sampled_distributions(weight_cache, manifest, mode):
    """xe.sampled_logprobs (with the tape) must reproduce, position by position, the log-prob rows the engine sampled from."""
    from boficap_amd import xe
    cfg, sd, model = _model(weight_cache, manifest)
    att = _images().cuda()
    n = 3
    r = _sample(model, att, mode, n, T=1.3)
    assert r["seq"].shape == (att.size(0) * n, cfg.seq_length)
    lp_s, lp_n = xe.sampled_logprobs(xe.Params(model), cfg, att, None, r if mode == "SAIC" else None, r if mode == "NAIC" else None,
                                     sample_n=n, strict_q1=True)
    lp = lp_s if mode == "SAIC" else lp_n
    ntok = r["phrase_length"].sum(1)
    eng = r["seq_logprob"]
    checked = 0
    for i in range(lp.shape[0]):
        k = int(ntok[i]) if mode == "SAIC" else cfg.seq_length           # SAIC keeps the rows of emitted phrases only
        if k == 0 or eng[i, :k].isnan().any():
            continue
        err = float((lp[i, :k].detach() - eng[i, :k]).abs().max())
        assert err < 2e-3, (i, err)
        checked += 1
    assert checked >= lp.shape[0] // 2
    if mode == "SAIC":                                                     # copies of an image diverge, layouts included
        seqs = r["seq"].view(att.size(0), n, -1)
        assert any(not torch.equal(seqs[b, 0], seqs[b, 1]) for b in range(att.size(0)))


def test_self_critical_loss_and_gradients_vs_oracle_autograd(weight_cache, manifest):
    """new_self_critical on re-forwarded log-probs: value and parameter gradients against torch autograd over the oracle's
    decode_sa / decode_na on the CPU with the same sampled captions and the same scores."""
    from boficap_amd import xe
    from boficap_amd.collate import phrase_collate
    cfg, sd, model = _model(weight_cache, manifest)
    att = _images()[:3]
    n = 2
    saic, naic = _sample(model, att.cuda(), "SAIC", n), _sample(model, att.cuda(), "NAIC", n)
    g = torch.Generator().manual_seed(0)
    sc_s, sc_n = torch.rand(att.size(0) * n, generator=g), torch.rand(att.size(0) * n, generator=g)
    lp_s, lp_n = xe.sampled_logprobs(xe.Params(model), cfg, att.cuda(), None, saic, naic, sample_n=n, strict_q1=True)
    loss = xe.new_self_critical(lp_s, saic["seq"], sc_s, n)[0] + xe.new_self_critical(lp_n, naic["seq"], sc_n, n)[0]
    loss.backward()
    # the same on the CPU
    w = {k: torch.from_numpy(v).clone().requires_grad_(k != "model.pos_embed.pe") for k, v in sd.items()}
    memory, src_mask = O.memory_of(w, cfg, att, None)
    memory, src_mask = memory.repeat_interleave(n, 0), src_mask.repeat_interleave(n, 0)
    S = cfg.seq_length

    def collate(r):
        N = r["seq"].shape[0]
        labels = np.zeros((N, S + 2), np.int64)
        labels[:, 0] = cfg.bos_idx
        labels[:, 1:S + 1] = r["seq"].cpu().numpy()
        plen = r["phrase_length"].cpu().numpy().astype(np.int64)
        return phrase_collate(labels, plen, np.where(plen > 0, r["phrase_syn"].cpu().numpy(), 0), len_idx=cfg.len_idx)

    cs, cn = collate(saic), collate(naic)
    ref_s = F.log_softmax(O.logit(w, O.decode_sa(w, cfg, memory, torch.from_numpy(cs["extend_phrase_seq"]),
                                                  torch.from_numpy(cs["extend_phrase_syn_seq"][:, 1:-1].copy()), src_mask,
                                                  torch.from_numpy(cs["extend_phrase_seq_mask"]))), dim=-1)
    last = cn["phrase_length"][:, 1:].sum(1) + 1
    syn_mask = torch.zeros(last.shape[0], S, S, dtype=torch.bool)
    syn_mask[:, :, :int(last[-1]) - 1] = True                             # quirk Q1
    ref_n = F.log_softmax(O.logit(w, O.decode_na(w, cfg, memory, torch.from_numpy(cn["extend_phrase_syn_seq"][:, 1:-1].copy()), src_mask, syn_mask)),
                          dim=-1)

    def nsc(lp, seq, sc):
        mask = (seq > 0).float()
        mask = torch.cat([torch.ones(mask.size(0), 1), mask[:, :-1]], 1)
        s = sc.view(-1, n)
        rew = s - (s.sum(1, keepdim=True) - s) / (n - 1)
        return (-lp.gather(2, seq.unsqueeze(2)).squeeze(2) * mask * rew.view(-1, 1)).sum() / mask.sum()

    ref = nsc(ref_s, saic["seq"].cpu(), sc_s) + nsc(ref_n, naic["seq"].cpu(), sc_n)
    if not torch.isfinite(ref):
        pytest.skip("sampled batch hit the all-masked (NaN) corner of quirk Q1")
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-3 * max(1.0, abs(float(ref.detach())))
    for name, p in model.named_parameters():
        r = w[name].grad
        if r is None or float(r.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, name
            continue
        err = float((p.grad.cpu() - r).abs().max())
        assert err <= 3e-3 * max(1e-3, float(r.abs().max())), (name, err)


def test_rl_step_runs_and_moves_the_weights(weight_cache, manifest):
    from boficap_amd.trainer import XETrainer
    cfg, sd, model = _model(weight_cache, manifest)
    opt = cfg.to_opt()
    opt.noamopt, opt.learning_rate = False, 1e-4
    tr = XETrainer(model, opt)
    att = _images().cuda()
    w0 = tr.bucket.flat.clone()
    target = 11

    def score(seq):                                                        # toy scorer: share of tokens equal to one id
        return (seq == target).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1)

    model.train()
    for _ in range(3):
        loss, rs, rn = tr.rl_step(att, None, score, sample_n=4, temperature=1.0)
        assert torch.isfinite(loss)
    assert model.training and float((tr.bucket.flat - w0).abs().max()) > 0
    length_w = dict(model.named_parameters())["model.length_predictor.Length_classifier2.weight"]
    assert torch.equal(length_w.detach().cpu(), torch.from_numpy(sd["model.length_predictor.Length_classifier2.weight"]))   # no RL gradient reaches the bound heads


def test_rl_step_with_ragged_regions_and_bf16(weight_cache, manifest):
    """att_masks (ragged region counts) through both samplers and the re-forward, bf16 operands, graph-mode trainer falling
    back to eager for the self-critical step."""
    from boficap_amd import xe
    from boficap_amd.trainer import XETrainer
    cfg, sd, model = _model(weight_cache, manifest)
    model.train_dtype = torch.bfloat16
    att = _images().cuda()
    B = att.size(0)
    masks = torch.ones(B, 36, device="cuda")
    for i, n in enumerate((36, 30, 25, 36, 19, 33)[:B]):
        masks[i, n:] = 0
        att[i, n:] = 0
    # the re-forward still returns the sampled distributions with masks in play (float32 check first)
    model.train_dtype = torch.float32
    fc = torch.zeros(B, 0, device="cuda")
    with torch.no_grad():
        r = model(fc, att, masks, opt={"train_mode": "SAIC", "sample_method": "sample", "sample_n": 2}, mode="sample")
    rs = dict(zip(("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn"), r[:5]))
    lp, _ = xe.sampled_logprobs(xe.Params(model), cfg, att, masks, rs, None, sample_n=2)
    ntok = rs["phrase_length"].sum(1)
    for i in range(lp.shape[0]):
        k = int(ntok[i])
        if k and not rs["seq_logprob"][i, :k].isnan().any():
            assert float((lp[i, :k].detach() - rs["seq_logprob"][i, :k]).abs().max()) < 2e-3
    model.train_dtype = torch.bfloat16
    opt = cfg.to_opt()
    opt.noamopt, opt.learning_rate = False, 1e-4
    tr = XETrainer(model, opt, graph=True)
    model.train()
    model.opt.bofi_rl_reference_estimator = False               # the fast form of rounds 1-4 (opt-in since round 5): engine samples, re-forward with the tape
    try:
        loss, _, _ = tr.rl_step(att, masks, lambda seq: (seq % 7 == 0).float().mean(1), sample_n=3)
        assert torch.isfinite(loss) and "reference_gap" not in tr._last_rl
    finally:
        del model.opt.bofi_rl_reference_estimator
    # the DEFAULT: the reference's estimator on the same inputs: bf16 operands and ragged regions leave the drawn rows the gradient pass's rows
    loss, _, _ = tr.rl_step(att, masks, lambda seq: (seq % 7 == 0).float().mean(1), sample_n=3)
    assert torch.isfinite(loss) and tr._last_rl["reference_gap"] == 0.0, tr._last_rl["reference_gap"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rl_gradient_pass_replayed_as_a_graph(weight_cache, manifest, dtype):
    """The gradient pass of the self-critical step (re-forward of the samples, new_self_critical for both modes, backward)
    captured as a hipGraph and replayed: same loss and gradients as the eager pass on the same samples and scores."""
    from boficap_amd import xe
    from boficap_amd.trainer import XETrainer
    cfg, sd, eager = _model(weight_cache, manifest)
    _, _, graphed = _model(weight_cache, manifest)
    n = 3
    att = _images().cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    ks = ("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn")
    eager.eval(); graphed.eval()
    with torch.no_grad():
        opt = {"sample_method": "sample", "sample_n": n, "temperature": 1.0}
        saic = dict(zip(ks, eager(fc, att, None, opt=dict(opt, train_mode="SAIC"), mode="sample")[:5]))
        naic = dict(zip(ks, eager(fc, att, None, opt=dict(opt, train_mode="NAIC"), mode="sample")[:5]))
    score = lambda seq: (seq % 5 == 0).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1)
    b = {"att_feats": att, "seq_saic": saic["seq"].long(), "seq_naic": naic["seq"].long(),
         "sc_saic": score(saic["seq"].cpu()).cuda(), "sc_naic": score(naic["seq"].cpu()).cuda()}
    b.update(xe.rl_prepare(cfg, saic, naic, sample_n=n, device="cuda"))
    for m in (eager, graphed):
        m.train_dtype = dtype
    te, tg = XETrainer(eager), XETrainer(graphed, graph=True)
    le, _, _ = te._rl_forward_backward(b, None, n)
    for _ in range(2):                                                     # capture, then replay
        lg, _, _ = tg._rl_replay(b, n)
    assert torch.isfinite(le) and abs(float(le) - float(lg)) < 1e-5 * max(1.0, abs(float(le)))
    tol = 1e-4 if dtype == torch.float32 else 4e-4
    assert float((tg.bucket.grad - te.bucket.grad).abs().max()) <= tol * max(1e-3, float(te.bucket.grad.abs().max()))
    assert float(te.bucket.grad.abs().max()) > 0


def test_rl_step_with_kl_term(weight_cache, manifest):
    """opt.rl_kl (loss_wrapper.py:216-222): the step's loss gains KL(SAIC || NAIC) over the SAIC captions' tokens -- positive, with a
    gradient through the NAIC branch; eager and captured gradient passes agree."""
    from boficap_amd import xe
    from boficap_amd.trainer import XETrainer
    cfg, sd, plain = _model(weight_cache, manifest)
    _, _, withkl = _model(weight_cache, manifest)
    _, _, graphed = _model(weight_cache, manifest)
    n = 3
    att = _images().cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    ks = ("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn")
    with torch.no_grad():
        o = {"sample_method": "sample", "sample_n": n, "temperature": 1.0}
        saic = dict(zip(ks, plain(fc, att, None, opt=dict(o, train_mode="SAIC"), mode="sample")[:5]))
        naic = dict(zip(ks, plain(fc, att, None, opt=dict(o, train_mode="NAIC"), mode="sample")[:5]))
    score = lambda seq: (seq % 5 == 0).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1)
    b = {"att_feats": att, "seq_saic": saic["seq"].long(), "seq_naic": naic["seq"].long(),
         "sc_saic": score(saic["seq"].cpu()).cuda(), "sc_naic": score(naic["seq"].cpu()).cuda()}
    b.update(xe.rl_prepare(cfg, saic, naic, sample_n=n, device="cuda"))
    opt = cfg.to_opt(rl_kl=True)
    t0, t1, t2 = XETrainer(plain), XETrainer(withkl, opt), XETrainer(graphed, opt, graph=True)
    assert not t0.rl_kl and t1.rl_kl
    l0, _, _ = t0._rl_forward_backward(b, None, n)
    l1, _, _ = t1._rl_forward_backward(b, None, n)
    for _ in range(2):
        l2, _, _ = t2._rl_replay(b, n)
    if not torch.isfinite(l0):
        pytest.skip("sampled batch hit the all-masked (NaN) corner of quirk Q1")
    assert float(l1) > float(l0) and abs(float(l1) - float(l2)) < 1e-5 * max(1.0, abs(float(l1)))
    assert float((t1.bucket.grad - t0.bucket.grad).abs().max()) > 0
    assert float((t2.bucket.grad - t1.bucket.grad).abs().max()) <= 1e-4 * float(t1.bucket.grad.abs().max())


def test_sample_pair_equals_the_two_sample_calls(weight_cache, manifest):
    """TransformerModel.sample_pair overlaps the two modes' sampling decodes (engine fork, second stream); it must return what
    the reference's two mode='sample' calls (loss_wrapper.py:193-209) return with the same seeds."""
    cfg, sd, model = _model(weight_cache, manifest)
    att = _images().cuda()
    n = 3
    for rep in range(2):                                      # the second round replays the captured graph
        model._sample_calls = 10 * rep
        a_s = _sample(model, att, "SAIC", n, T=1.2)
        a_n = _sample(model, att, "NAIC", n, T=1.2)
        model._sample_calls = 10 * rep
        with torch.no_grad():
            b_s, b_n = model.sample_pair(att, None, n, 1.2)
        torch.cuda.synchronize()
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(a_s[k], b_s[k]) and torch.equal(a_n[k], b_n[k]), k
        for a, b in ((a_s, b_s), (a_n, b_n)):
            assert torch.equal(a["seq_logprob"].nan_to_num(0.0), b["seq_logprob"].nan_to_num(0.0))


def test_capped_semi_autoregressive_loop_with_its_continuation_equals_the_whole_loop(weight_cache, manifest):
    """bofi_engine_set_saic_range / sample_pair(saic_cap=c) + saic_finish: the loop's first c iterations, then -- when the count of live
    iterations says it may not be through -- the rest on the state the engine still holds: ids, layouts and log-probs of the whole loop,
    bit for bit, sampled and greedy, whatever c; a cap past the last live iteration needs no second part."""
    cfg, sd, model = _model(weight_cache, manifest)
    att = _images().cuda()
    n, S = 3, cfg.seq_length
    model._sample_calls = 40
    with torch.no_grad():
        ref_s, _ = model.sample_pair(att, None, n, 1.2)
    ref_s = model.saic_finish(ref_s)
    live = int(ref_s["bound_iters"])
    assert 2 <= live < S                                       # (the fixture's captions end after a few phrases)
    for cap in (1, live - 1, live, live + 1, S - 1, S):
        model._sample_calls = 40
        with torch.no_grad():
            got, _ = model.sample_pair(att, None, n, 1.2, saic_cap=cap)
        second = cap < S and cap <= live                       # live iterations == cap: "may not be through"
        assert ("_capped" in got) == (cap < S)
        partial = int(got["bound_iters"])
        assert partial == min(cap, live)
        got = model.saic_finish(got)
        assert int(got["bound_iters"]) == live, (cap, second)
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(got[k], ref_s[k]), (cap, k)
        assert torch.equal(got["seq_logprob"].nan_to_num(0.0), ref_s["seq_logprob"].nan_to_num(0.0)), cap
    # the engine entry itself, greedy, without the graph
    eng = model.engine()
    feats = model._as_input(att)
    whole = eng.decode_saic(feats, None)
    part = eng.decode_saic(feats, None, it_range=(1, 2))
    assert int(part["bound_iters"]) == 2
    rest = eng.decode_saic(feats, None, out=part, it_range=(3, S))
    torch.cuda.synchronize()
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
        assert torch.equal(rest[k], whole[k]), k
    assert torch.equal(rest["seq_logprob"].nan_to_num(0.0), whole["seq_logprob"].nan_to_num(0.0))
    # the proposal follows the recent decodes, in steps of 2, and stays put (every value is a captured graph of its own)
    model.__dict__.pop("_saic_recent", None); model.__dict__.pop("_saic_cap_cur", None)
    assert model.saic_cap() is None
    for _ in range(2):
        model.saic_finish(dict(ref_s))
    want = -(-(live + 2) // 2) * 2
    assert model.saic_cap() == (want if want < S else None)
    for _ in range(8):                                         # shorter captions for eight decodes in a row: one step down at most, only then
        r = dict(ref_s); r["bound_iters"] = torch.tensor([max(1, live - 5)], dtype=torch.int32)
        before = model.saic_cap()
        model.saic_finish(r)
    assert before == (want if want < S else None)
    low = -(-(max(1, live - 5) + 2) // 2) * 2
    assert model.saic_cap() == (low if low < S else None) and low <= want


@pytest.mark.parametrize("graph", [False, True])
def test_reference_estimator_draws_every_token_from_the_gradient_pass(graph, weight_cache, manifest):
    """opt.bofi_rl_reference_estimator: the reference samples in train mode and differentiates that same pass (loss_wrapper.py:193-209).  Here every
    token of both branches is drawn from the training forward's rows under the step's dropout masks, phrase by phrase, and the gradient pass is that
    forward once more with the tape: its rows at the drawn tokens ARE the rows they were drawn from (gap 0), and they are not the inference engine's
    dropout-free rows.  graph: the gradient pass replayed from a captured hipGraph (device-side step word for the masks) -- the same rows still."""
    from boficap_amd import xe
    from boficap_amd.trainer import XETrainer
    cfg, sd, model = _model(weight_cache, manifest)
    opt = cfg.to_opt()
    opt.noamopt, opt.learning_rate = False, 1e-4
    model.opt.bofi_rl_reference_estimator = True
    tr = XETrainer(model, opt, graph=graph)
    att = _images().cuda()
    w0 = tr.bucket.flat.clone()

    def score(seq):
        return (seq == 11).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1)

    model.train()
    assert cfg.dropout > 0
    n = 4
    for _ in range(2):
        loss, rs, rn = tr.rl_step(att, None, score, sample_n=n, temperature=1.0)
        last = tr._last_rl
        assert torch.isfinite(loss) and last["reference_gap"] == 0.0, last["reference_gap"]
        assert last["gradient_pass_replayed"] == graph
        assert last["training_forwards"] >= 3                                 # several phrases -> several tape-free forwards + the gradient pass
        assert last["seq_saic"].shape == (att.size(0) * n, cfg.seq_length) and int((last["seq_saic"] > 0).sum()) > att.size(0)
        assert int((last["seq_naic"] > 0).sum()) > 0
    assert model.training and float((tr.bucket.flat - w0).abs().max()) > 0
    length_w = dict(model.named_parameters())["model.length_predictor.Length_classifier2.weight"]
    assert torch.equal(length_w.detach().cpu(), torch.from_numpy(sd["model.length_predictor.Length_classifier2.weight"]))
    # the drawn captions' rows under dropout are not the dropout-free rows of the same captions
    saic = {"seq": last["seq_saic"], "phrase_length": last["phrase_length_saic"], "phrase_syn": last["phrase_syn_saic"]}
    with torch.no_grad():
        a, _ = xe.sampled_logprobs(xe.Params(model), cfg, att, None, saic, None, sample_n=n, training=True, seed=12345)
        b, _ = xe.sampled_logprobs(xe.Params(model), cfg, att, None, saic, None, sample_n=n, training=False)
    live = last["seq_saic"] > 0
    assert float((a - b)[live].abs().max()) > 1e-3


def test_reference_estimator_with_a_full_graph_cache(weight_cache, manifest):
    """once the trainer's graph cache holds max_graphs entries (a normal XE run fills it: the key carries the max-phrase / max-token buckets and the GLAT
    rate) the reference-estimator step's gradient pass runs eagerly -- and must still find its own log-probs at the drawn tokens (the "picked" buffers of the dict IT filled),
    whatever a replay left behind earlier: gap 0.0, weights moved, no AttributeError / stale buffers."""
    from boficap_amd.trainer import XETrainer
    cfg, sd, model = _model(weight_cache, manifest)
    opt = cfg.to_opt()
    opt.noamopt, opt.learning_rate = False, 1e-4
    model.opt.bofi_rl_reference_estimator = True
    att = _images().cuda()

    def score(seq):
        return (seq == 11).float().sum(1) / (seq > 0).float().sum(1).clamp(min=1)

    model.train()
    for leftover in (False, True):
        tr = XETrainer(model, opt, graph=True)
        if leftover:                                                            # a replayed step first: _rl_static then points at THAT graph's buffers (another sample_n)
            tr.rl_step(att, None, score, sample_n=2, temperature=1.0)
            assert tr._last_rl["gradient_pass_replayed"] and tr._last_rl["reference_gap"] == 0.0
        while len(tr._graphs) < tr.max_graphs:                                   # what a run of XE steps with different phrase buckets leaves
            tr._graphs[("filler", len(tr._graphs))] = None
        w0 = tr.bucket.flat.clone()
        for _ in range(2):
            loss, rs, rn = tr.rl_step(att, None, score, sample_n=3, temperature=1.0)
            last = tr._last_rl
            assert torch.isfinite(loss) and last["reference_gap"] == 0.0, last["reference_gap"]
            assert last["gradient_pass_replayed"]                               # (the step takes the replay branch; the replay itself fell back to the eager pass)
            assert tr._rl_static is not None and tuple(tr._rl_static["picked_saic"].shape) == (att.size(0) * 3, cfg.seq_length)
            assert float(tr._rl_static["picked_saic"].abs().max()) > 0
        assert len(tr._graphs) == tr.max_graphs
        assert float((tr.bucket.flat - w0).abs().max()) > 0


def test_reference_estimator_cold_and_without_dropout_is_the_greedy_decode(weight_cache, manifest):
    """The phrase-by-phrase process of the reference-estimator step (engine bounding step -> training-forward rows -> draw -> words handed back to the
    engine) at temperature -> 0 in eval mode is core_SAIC's greedy decode (and the greedy fill of the non-autoregressive layout): layouts and tokens
    equal the engine's own greedy results."""
    from boficap_amd.trainer import XETrainer
    cfg, sd, model = _model(weight_cache, manifest)
    opt = cfg.to_opt()
    opt.noamopt, opt.learning_rate = False, 1e-5
    model.opt.bofi_rl_reference_estimator = True
    tr = XETrainer(model, opt)
    att = _images().cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    with torch.no_grad():
        gs = model(fc, att, None, opt={"train_mode": "SAIC", "sample_method": "greedy"}, mode="sample")
        gn = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
    n = 2
    tr.rl_step(att, None, lambda seq: (seq > 0).float().mean(1), sample_n=n, temperature=1e-3)
    last = tr._last_rl
    assert torch.equal(last["phrase_length_saic"].long(), gs[3].long().repeat_interleave(n, 0))
    assert torch.equal(last["seq_saic"], gs[0].repeat_interleave(n, 0))
    assert torch.equal(last["seq_naic"], gn[0].repeat_interleave(n, 0))
    assert last["reference_gap"] == 0.0


def test_device_side_layout_collate_equals_the_host_collate():
    """xe.rl_prepare_saic_device (tensor operations on the device, usable inside a captured graph) against xe.rl_prepare (the host collate through
    boficap_amd.collate.phrase_collate, itself pinned to the reference's collate_func): random layouts with squeezes, stretches, empty captions."""
    from boficap_amd import xe
    from boficap_amd.config import FULL as cfg
    rng = np.random.default_rng(3)
    S, N = cfg.seq_length, 60
    seq = np.zeros((N, S), np.int64); plen = np.zeros((N, S), np.int32); psyn = np.zeros((N, S), np.int64)
    for n in range(N):
        if n % 11 == 0:
            continue                                             # a caption without phrases
        lens = rng.integers(1, 8, int(rng.integers(1, 9)))
        while lens.sum() > S:
            lens = lens[:-1]
        P = len(lens)
        plen[n, :P], psyn[n, :P] = lens, rng.integers(4, 7, P)
        psyn[n, P:] = rng.integers(0, 9, S - P)                  # (entries behind the last phrase are not read)
        seq[n, :lens.sum()] = rng.integers(7, cfg.tgt_vocab, lens.sum())
    r = {"seq": torch.from_numpy(seq).cuda(), "phrase_length": torch.from_numpy(plen).cuda(), "phrase_syn": torch.from_numpy(psyn).cuda()}
    host = xe.rl_prepare(cfg, r, None, sample_n=1, device="cuda")
    for tensor_ops in (False, True):                             # the one-launch kernel (bofi_saic_collate) and its tensor-operation reference
        xe._COLLATE["tensor_ops"] = tensor_ops
        try:
            dev = xe.rl_prepare_saic_device(cfg, r["seq"], r["phrase_length"], r["phrase_syn"])
        finally:
            xe._COLLATE["tensor_ops"] = False
        for k in ("sa_syn", "sa_seq", "sa_klen"):
            assert dev[k].dtype == host[k].dtype and torch.equal(dev[k], host[k]), (tensor_ops, k)
    for strict in (True, False):                                 # the non-autoregressive half (labels + quirk Q1's fill mask)
        host_na = xe.rl_prepare(cfg, None, r, sample_n=1, strict_q1=strict, device="cuda")
        dev_na = xe.rl_prepare_naic_device(cfg, r["phrase_length"], r["phrase_syn"], strict_q1=strict)
        for k in ("na_syn", "na_klen"):
            assert dev_na[k].dtype == host_na[k].dtype and torch.equal(dev_na[k], host_na[k]), (strict, k)
