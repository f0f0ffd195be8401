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
