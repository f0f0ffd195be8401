"""The CPU oracle against the golden vectors recorded from the real reference (no GPU, no reference)."""
import numpy as np
import pytest
import torch

import boficap_oracle as O
from conftest import TINY_CASES, load_golden


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape
    assert (np.isnan(a) == np.isnan(b)).all()
    m = ~np.isnan(a)
    return float(np.abs(a[m] - b[m]).max()) <= tol if m.any() else True


@pytest.mark.parametrize("name", TINY_CASES)
def test_oracle_matches_reference_tiny(name, manifest, weight_cache):
    m = manifest[name]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden(name)
    att = torch.from_numpy(g["att_feats"])
    masks = torch.from_numpy(g["att_masks"]) if "att_masks" in g else None
    mem, sm = O.memory_of(w, cfg, att, masks)
    assert _close(mem.numpy(), g["memory"], 2e-6)
    L = cfg.seq_length + 2
    ext = torch.zeros(att.size(0), L, dtype=torch.long); ext[:, 0] = cfg.len_idx
    tm = torch.zeros(att.size(0), L, L, dtype=torch.bool); tm[:, :, 0] = True
    ln, llp, sn, slp = O.bound_step_na(w, cfg, ext, mem, sm, tm)
    assert (ln.numpy() == g["step0_len_n"]).all() and (sn.numpy() == g["step0_syn_n"]).all()
    assert _close(llp.numpy(), g["step0_len_logp"], 2e-6) and _close(slp.numpy(), g["step0_syn_logp"], 2e-6)
    seq, lp, pn, pl, ps, _ = O.sample_naic(w, cfg, att, masks)
    assert (seq.numpy() == g["naic_seq"]).all()
    assert (pn.numpy() == g["naic_phrase_num"]).all()
    assert (pl.numpy() == g["naic_phrase_length"]).all()
    assert (ps.numpy() == g["naic_phrase_syn"]).all()
    assert _close(lp.numpy(), g["naic_logprob"], 1e-5)
    seq, lp, pn, pl, ps, _ = O.sample_saic(w, cfg, att, masks)
    assert (seq.numpy() == g["saic_seq"]).all()
    assert (pn.numpy() == g["saic_phrase_num"]).all()
    assert (pl.numpy() == g["saic_phrase_length"]).all()
    assert (ps.numpy() == g["saic_phrase_syn"]).all()
    assert _close(lp.numpy(), g["saic_logprob"], 1e-5)


def test_case_mix_covers_reference_branches(manifest):
    """EOS by length 0, EOS by label out of range, truncation at 21, Q1 NaN batch."""
    reasons = set()
    for name in TINY_CASES:
        reasons.update(manifest[name].get("reasons", []))
    assert {"len0", "syn", "trunc"} <= reasons
    g = load_golden("tiny_q1_last_empty_nan")
    assert np.isnan(g["naic_logprob"]).all() and (g["naic_seq"] == 0).all()
    g = load_golden("tiny_q1_last_shortest")
    assert g["naic_last"][-1] < g["naic_last"][:-1].min()
    g = load_golden("tiny_saic_multi")                        # the semi-autoregressive mode beyond its first iteration
    assert g["saic_phrase_num"].max() >= 4 and (g["saic_seq"] > 0).sum() > 20 and not np.isnan(g["saic_logprob"]).any()


def test_oracle_matches_reference_full(manifest, weight_cache):
    from boficap_amd import weights as W
    m = manifest["full_b8"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("full_b8")
    att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]])
    seq, lp, pn, pl, ps, _ = O.sample_naic(w, cfg, att, None)
    assert (seq.numpy() == g["naic_seq"]).all()
    assert (pn.numpy() == g["naic_phrase_num"]).all() and (pl.numpy() == g["naic_phrase_length"]).all()
    assert (ps.numpy() == g["naic_phrase_syn"]).all()
    top = torch.topk(lp, 2, dim=2)
    assert _close(top[0].numpy(), g["naic_top2_val"], 2e-5)
    assert _close(lp[:2, :3].numpy(), g["naic_logprob_rows"], 2e-5)


def test_q1_fix_changes_only_fill(manifest, weight_cache):
    """The strict_reference=False escape hatch (per-row fill mask) leaves the slot layout alone."""
    m = manifest["tiny_q1_last_shortest"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("tiny_q1_last_shortest")
    att = torch.from_numpy(g["att_feats"])
    a = O.sample_naic(w, cfg, att, None)
    b = O.sample_naic(w, cfg, att, None, fix_q1=True)
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert not torch.equal(a[0], b[0])


def test_oracle_xe_forward_and_criterion(manifest, weight_cache):
    """XE training forward (six log-prob tensors) and LanguageModelCriterion_UIC vs the reference's recorded outputs."""
    m = manifest["tiny_train_xe"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("tiny_train_xe")
    t = lambda k: torch.from_numpy(g[k])
    outs = O.forward_uic(w, cfg, t("att_feats"), t("labels"), None, t("phrase_num"), t("phrase_length"), t("phrase_syn"),
                         t("extend_phrase_syn_seq"), t("extend_phrase_seq"), t("extend_phrase_seq_mask"))
    for i, o in enumerate(outs):
        assert _close(o.numpy(), g[f"out{i}"], 1e-5)
    loss, parts = O.criterion_uic(outs, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("labels"))
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4
    assert np.allclose([float(p) for p in parts], g["losses"][1:], atol=1e-5)


def test_full_fixture_exercises_q1_shortening(manifest):
    """full_b8: the LAST image has one token, so quirk Q1 cuts every image's fill mask (the 20-token ones included) to key 0."""
    last = manifest["full_b8"]["last"]
    assert 1 < last[-1] < max(last)
    g = load_golden("full_b8")
    assert not np.isnan(g["naic_top2_val"]).any() and (g["naic_phrase_length"].sum(1).max() == 20)


def test_oracle_saic_multi_phrase_full(manifest, weight_cache):
    """core_SAIC beyond its first iteration at the full size (up to 10 phrases per image)."""
    from boficap_amd import weights as W
    m = manifest["full_saic_multi"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("full_saic_multi")
    assert max(m["saic_phrase_num"]) >= 4
    att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]])
    seq, lp, pn, pl, ps, _ = O.sample_saic(w, cfg, att, None)
    assert (seq.numpy() == g["saic_seq"]).all() and (pn.numpy() == g["saic_phrase_num"]).all()
    assert (pl.numpy() == g["saic_phrase_length"]).all() and (ps.numpy() == g["saic_phrase_syn"]).all()
    assert _close(torch.topk(lp, 2, dim=2)[0].numpy(), g["saic_top2_val"], 2e-5)


def test_oracle_glancing_pass(manifest, weight_cache):
    """EncoderDecoder_UIC.forward with glat_p >= 0 (TM:437-463), the reference's torch.rand draw injected."""
    m = manifest["tiny_glat"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("tiny_glat")
    t = lambda k: torch.from_numpy(g[k])
    args = (t("att_feats"), t("labels"), None, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("extend_phrase_syn_seq"),
            t("extend_phrase_seq"), t("extend_phrase_seq_mask"))
    outs = O.forward_uic(w, cfg, *args, glat_p=float(g["glat_p"]), glat_uniform=t("glat_uniform"))
    for i, o in enumerate(outs):
        assert _close(o.numpy(), g[f"out{i}"], 1e-5), i
    plain = O.forward_uic(w, cfg, *args)
    assert float((plain[5] - outs[5]).abs().max()) > 1e-2 and _close(plain[2].numpy(), g["out2"], 1e-5)     # only the NA tokens change
    loss, _ = O.criterion_uic(outs, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("labels"))
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4


def test_oracle_self_critical_losses(manifest):
    """StructureLosses('new_self_critical') and LossWrapper's UIC struc_flag branch (with and without rl_kl) vs the values and
    gradients recorded from the reference with injected samples and scores."""
    g = load_golden("tiny_rl_loss")
    n = int(g["sample_n"])
    t = lambda k: torch.from_numpy(g[k])
    a = t("saic_logprob").clone().requires_grad_(True)
    loss, reward = O.new_self_critical(a, t("saic_seq"), g["saic_scores"], n)
    loss.backward()
    assert abs(float(loss) - float(g["nsc_loss"])) < 1e-6 and np.allclose(reward.numpy(), g["nsc_reward"])
    assert _close(a.grad.gather(2, t("saic_seq").unsqueeze(2)).squeeze(2).numpy(), g["nsc_grad_picked"], 1e-7)
    for rl_kl, tag in ((False, "lw"), (True, "lw_kl")):
        ls, ln = t("saic_logprob").clone().requires_grad_(True), t("naic_logprob").clone().requires_grad_(True)
        o = O.loss_wrapper_uic_rl(ls, t("saic_seq"), ln, t("naic_seq"), g["saic_scores"], g["naic_scores"], n, rl_kl=rl_kl)
        assert abs(float(o["loss"]) - float(g[tag + "_loss"])) < 1e-5 and abs(float(o["struc_loss"]) - float(g[tag + "_struc_loss"])) < 1e-5
        assert np.allclose(o["reward"].numpy(), g[tag + "_reward"])
        if rl_kl:
            o["loss"].backward()
            assert _close(ln.grad.numpy(), g["lw_kl_grad_naic"], 1e-6)
            assert _close(ls.grad.gather(2, t("saic_seq").unsqueeze(2)).squeeze(2).numpy(), g["lw_kl_grad_saic_picked"], 1e-7)


def _structure_loss_cases(g):
    for key in list(g):
        if key.endswith("_loss") and key != "reward":
            tag = key[:-5]
            ent = float(g[tag + "_entropy_weight"])
            body = tag[:-4] if tag.endswith("_ent") else tag
            loss_type, reduction = body.rsplit("_", 1)
            yield tag, loss_type, reduction, ent


def test_oracle_structure_loss_types(manifest):
    """Every structure_loss_type of StructureLosses (losses.py:72-176), reductions and the entropy reward: loss, reward and the gradient
    at the sampled ids as recorded from the reference's own code (tests/golden/tiny_structure_losses; the fixture also records that
    the reference as shipped raises NameError for every type but new_self_critical)."""
    g = load_golden("tiny_structure_losses")
    n = int(g["sample_n"])
    assert sorted(g["reference_raises_name_error"].tolist()) == sorted(t for t in O.STRUCTURE_LOSS_TYPES if t != "new_self_critical")
    seq = torch.from_numpy(g["seq"])
    cases = list(_structure_loss_cases(g))
    assert {c[1] for c in cases} == set(O.STRUCTURE_LOSS_TYPES) and len(cases) == 12
    for tag, loss_type, reduction, ent in cases:
        a = torch.from_numpy(g["logprob"]).clone().requires_grad_(True)
        loss, reward = O.structure_loss(loss_type, a, seq, g["scores"], n, reduction=reduction, entropy_reward_weight=ent)
        assert np.allclose(loss.detach().numpy(), g[tag + "_loss"], rtol=1e-5, atol=1e-6), tag
        assert np.allclose(reward.numpy(), g["reward"])
        w = torch.linspace(0.5, 1.5, loss.numel()).view_as(loss) if reduction == "none" else None
        ((loss * w).sum() if w is not None else loss).backward()
        assert _close(a.grad.gather(2, seq.unsqueeze(2)).squeeze(2).numpy(), g[tag + "_grad_picked"], 1e-6), tag


def test_oracle_loss_wrapper_xe_branch(manifest, weight_cache):
    """LossWrapper.forward (train_mode UIC, struc_flag False): the seven entries of its out dict."""
    m = manifest["tiny_loss_wrapper_xe"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("tiny_loss_wrapper_xe")
    t = lambda k: torch.from_numpy(g[k])
    outs = O.forward_uic(w, cfg, t("att_feats"), t("labels"), None, t("phrase_num"), t("phrase_length"), t("phrase_syn"),
                         t("extend_phrase_syn_seq"), t("extend_phrase_seq"), t("extend_phrase_seq_mask"))
    loss, parts = O.criterion_uic(outs, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("labels"))
    assert list(g["out_keys"]) == ["loss", "SA_length_loss", "SA_phrase_loss", "SA_syn_loss", "NA_length_loss", "NA_phrase_loss", "NA_syn_loss"]
    assert np.allclose([float(loss)] + [float(p) for p in parts], g["out_values"], atol=1e-5)


def test_oracle_xe_full_size(manifest, weight_cache):
    """The reference's XE forward at the full size (2 images x 5 captions): best-two log-probs, the labels' log-probs, losses."""
    m = manifest["full_train_xe"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    g = load_golden("full_train_xe")
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        outs = O.forward_uic(w, cfg, t("att_feats"), t("labels"), None, t("phrase_num"), t("phrase_length"), t("phrase_syn"),
                             t("extend_phrase_syn_seq"), t("extend_phrase_seq"), t("extend_phrase_seq_mask"))
    real = t("labels").reshape(-1, cfg.seq_length + 2)[:, 1:-1].long()
    for i, o in enumerate(outs):
        if f"out{i}" in g:
            assert _close(o.numpy(), g[f"out{i}"], 1e-5), i
        else:
            assert _close(torch.topk(o, 2, dim=2)[0].numpy(), g[f"out{i}_top2_val"], 2e-5), i
            assert _close(o.gather(2, real.unsqueeze(2)).squeeze(2).numpy(), g[f"out{i}_picked"], 2e-5), i
    loss, parts = O.criterion_uic(outs, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("labels"))
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4 * float(g["losses"][0])


def test_oracle_two_layer_bounding_network(manifest, weight_cache):
    """configs/uic_sd_N2.yml's shape (N_len = 2) at the small size: the oracle's generic bound stack against the reference."""
    m = manifest["tiny_n2"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    assert cfg.N_len == 2 and max(m["phrase_num"]) >= 8
    w = O.as_torch(sd)
    g = load_golden("tiny_n2")
    att = torch.from_numpy(g["att_feats"])
    seq, lp, pn, pl, ps, _ = O.sample_naic(w, cfg, att, None)
    assert (seq.numpy() == g["naic_seq"]).all() and (pn.numpy() == g["naic_phrase_num"]).all()
    assert (pl.numpy() == g["naic_phrase_length"]).all() and (ps.numpy() == g["naic_phrase_syn"]).all()
    assert _close(lp.numpy(), g["naic_logprob"], 1e-5)


def test_oracle_scheduled_sampling(manifest, weight_cache):
    """TransformerModel._forward with ss_prob > 0 (ss_SAIC TM:1988-2121 for the SA branch), the reference's random() draws
    injected: the six outputs, the loss and the gradient norms of the reference's own backward."""
    m = manifest["tiny_ss"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    w = O.as_torch(sd)
    for v in w.values():
        v.requires_grad_(v.is_floating_point())
    g = load_golden("tiny_ss")
    assert min(m["choices"].values()) >= 2 and m["iters"] >= 3          # every input choice occurs, several phrases deep
    t = lambda k: torch.from_numpy(g[k])
    it = iter(g["draws"])
    outs, trace = O.forward_uic_ss(w, cfg, t("att_feats"), t("labels"), None, t("phrase_num"), t("phrase_length"), t("phrase_syn"),
                                   t("extend_phrase_syn_seq"), float(g["ss_prob"]), lambda: float(next(it)))
    assert next(it, None) is None                                       # exactly the reference's number of draws
    for i, o in enumerate(outs):
        assert _close(o.detach().numpy(), g[f"out{i}"], 1e-5)
    assert (trace["seq"].numpy() == g["emitted_seq"]).all()
    loss, parts = O.criterion_uic(outs, t("phrase_num"), t("phrase_length"), t("phrase_syn"), t("labels"))
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4
    loss.backward()
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        p = w[n]
        if ref_norm < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            assert abs(float(p.grad.norm()) - ref_norm) <= 1e-3 * max(float(ref_norm), 1e-3), n


def test_oracle_collate_matches_the_reference_loader(manifest):
    """oracle/training_batch.py::collate_loops (the checker of boficap_amd.collate) against what the reference's own
    Dataset.collate_func returned (tests/golden/tiny_collate.npz, captioning/data/dataloader.py:231-452)."""
    from boficap_amd.config import TINY
    from training_batch import collate_loops
    g = load_golden("tiny_collate")
    S = TINY.seq_length
    n_cap = g["in_seqs"].shape[0]
    labels = np.zeros((n_cap, S + 2), np.int64)
    labels[:, 1:S + 1], labels[:, 0], labels[:, S + 1] = g["in_seqs"], TINY.bos_idx, TINY.eos_idx
    assert (labels == g["ragged_labels"].reshape(n_cap, S + 2)).all()                  # the loader's [BOS] / [EOS] framing (:295-300)
    out = collate_loops(TINY, labels, g["in_phrase_num"], g["in_phrase_length"], g["in_phrase_syn"])
    for k, v in out.items():
        assert (v == g["ragged_" + k].reshape(v.shape)).all() and (v == g["full_" + k].reshape(v.shape)).all(), k
