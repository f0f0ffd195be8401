"""Calibrate the bound-head output layers of the synthetic weight presets.  TEST INFRASTRUCTURE.

Random weights make the bounding pass emit EOS (or one fixed layout) for every image
(SURVEY.md §8c), which would leave the bound loop, the truncation branch and quirk Q1 untested
and the benchmark unrepresentative.  This script measures, with the CPU oracle, the raw logits of
``Length_classifier2`` / ``Syntactic_classifier2`` over a calibration batch and random slot
layouts, then rescales/offsets those two small layers so that the winning class depends on the
image and on the slots laid out so far, around a prior that favours lengths 1..4 and labels 4..6.

Output: ``boficap_amd/presets/bound_heads_<cfg>_seed<k>.npz`` (a few KB, committed).  All other
weights come from ``boficap_amd.weights.make_state_dict`` and are regenerated from the seed.

Run:  python oracle/calibrate_preset.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import boficap_oracle as O                                   # noqa: E402
from boficap_amd import weights as W                         # noqa: E402
from boficap_amd.config import FULL, TINY, TINY_N2           # noqa: E402

LEN_PRIOR = np.full((O.LENGTH_DIM,), -9.0, np.float32)
LEN_PRIOR[:5] = [-0.9, 0.9, 1.1, 0.6, 0.1]
LEN_PRIOR[9] = -1.6            # rare long phrase -> truncation branch (TM:1850-1859)
SYN_PRIOR = np.full((O.SYN_DIM,), -9.0, np.float32)
SYN_PRIOR[4:7] = [0.5, 0.7, 0.3]
SYN_PRIOR[1] = -1.8            # rare out-of-range label -> EOS-by-syn branch (TM:1846)


def row0_features(w, cfg, mem, sm, rng, n_layouts: int = 4, with_last: bool = False):
    """Final-LN row-0 vectors of the bound network for random plausible slot layouts (with_last: and the tokens laid out + 1 of each)."""
    B = mem.size(0)
    L = cfg.seq_length + 2
    lp = "model.length_predictor"
    outs, lasts = [], []
    for _ in range(n_layouts):
        ext = torch.zeros(B, L, dtype=torch.long)
        ext[:, 0] = cfg.len_idx
        mask = torch.zeros(B, L, L, dtype=torch.bool)
        mask[:, :, 0] = True
        for b in range(B):
            last = 1
            for _p in range(int(rng.integers(0, 8))):
                ln, sy = int(rng.integers(1, 5)), int(rng.integers(4, 7))
                if last + ln >= cfg.seq_length + 1:
                    break
                ext[b, last:last + ln] = sy
                mask[b, last:, :last + ln] = True
                last += ln
                mask[b, 0, :last] = True
            lasts.append(last)
        x = O.add_pe(w, O.embed(w, "model.syn_embed", ext, cfg.d_model))
        outs.append(O.bound_row0(w, cfg, x, mem, sm, mask))          # all N_len layers + the final norm, row 0
    if with_last:
        return torch.cat(outs, 0), torch.tensor(lasts, dtype=torch.float32)
    return torch.cat(outs, 0)


# per-config knobs found by trial (see DESIGN.md "synthetic weight preset"): gain on the bound
# layer's two attention output projections (makes row 0 depend more on history and image), the
# target spread of the head logits, and the priors of the two EOS classes.
KNOBS = {
    "tiny": dict(attn_gain=2.0, std=1.5, len0=0.8, syn1=-0.4),
    # full: the end-of-caption class rises with the number of tokens laid out (eos_slope logits per token around eos_at tokens), so that
    # the captions END at mid lengths as a trained head's do -- with a constant prior an image either stops at once or runs into the
    # truncation at 21 (round-2 VERDICT: no mid-length LAST row at full size).  syn1 is kept rare for the same reason.
    # The out-of-range label class (EOS by label, TM:1846) is switched off at this size: it won at the FIRST step for two thirds of the
    # images whatever its prior (the empty layout lies outside the states its logit was standardised on); the branch is pinned at the
    # tiny size (tiny_mix: reasons len0 / syn / trunc).
    "full": dict(attn_gain=2.0, std=1.5, len0=1.5, syn1=-3.5, eos_slope=0.3, eos_at=5.0, eos_noise=0.6, eos_rounds=5, syn1_off=True),
    "tiny_n2": dict(attn_gain=2.0, std=1.5, len0=0.8, syn1=-0.4),      # two-layer bounding network (configs/uic_sd_N2.yml)
}


def calibrate_eos(cfg, w0, mem, sm, knobs):
    """The preset whose captions END at mid lengths.  States: random plausible layouts over the calibration images (768 rows, `last`
    from 1 to 20).  Both heads are standardised on them (logit spread `std` over the live classes around the priors); the class
    "length 0" (end of caption) additionally gets eos_slope logits per laid-out token around eos_at tokens, through a ridge regression
    of `last` on the heads' hidden layer -- the [LEN] row attends the laid-out positions, so the hidden layer carries their count."""
    lp = "model.length_predictor"
    len_prior, syn_prior = LEN_PRIOR.copy(), SYN_PRIOR.copy()
    len_prior[0], syn_prior[1] = knobs["len0"], knobs["syn1"]
    feats, last = row0_features(w0, cfg, mem, sm, np.random.Generator(np.random.PCG64(7)), n_layouts=16, with_last=True)
    out = {}
    for rnd in range(int(knobs.get("eos_rounds", 1))):
        if rnd > 0:                                              # pool the states the loop itself visits under the previous round's heads
            w = dict(w0)
            w.update({k: torch.from_numpy(v) for k, v in out.items() if k != "attn_gain"})
            tr = []
            O.core_naic(w, cfg, mem, sm, trace=tr)
            f_on = torch.cat([t["out0"][t["active"]] for t in tr], 0)
            l_on = torch.cat([t["last"][t["active"]].float() for t in tr], 0)
            feats, last = torch.cat([feats, f_on], 0), torch.cat([last, l_on], 0)
        for head, prior, live in (("Length", len_prior, [0, 1, 2, 3, 4, 9]), ("Syntactic", syn_prior, [1, 4, 5, 6])):
            hid = F.relu(O.linear(feats, w0, f"{lp}.{head}_classifier1"))
            w2 = w0[f"{lp}.{head}_classifier2.weight"]
            raw = hid @ w2.T
            sc = knobs["std"] / float(raw.std(0)[live].mean())
            W2, B2 = w2 * sc, torch.from_numpy(prior) - sc * raw.mean(0)
            if head == "Length":
                X = torch.cat([hid, torch.ones(hid.shape[0], 1)], 1).double()
                A = X.T @ X + 1.0 * torch.eye(X.shape[1], dtype=torch.float64)
                beta = torch.linalg.solve(A, X.T @ last.double())
                pred = (X @ beta).float()
                print("  round %d: ridge fit of `last` on the hidden layer: %d states, R^2 %.3f" % (rnd, hid.shape[0], 1.0 - float(((pred - last) ** 2).sum() / ((last - last.mean()) ** 2).sum())))
                W2, B2 = W2.clone(), B2.clone()
                # image-dependent part damped (eos_noise), so that the caption lengths spread by ~ eos_noise * std / eos_slope tokens around eos_at
                W2[0] = knobs["eos_noise"] * W2[0] + knobs["eos_slope"] * beta[:-1].float()
                B2[0] = knobs["eos_noise"] * (B2[0] - float(prior[0])) + float(prior[0]) + knobs["eos_slope"] * (float(beta[-1]) - knobs["eos_at"])
            if head == "Syntactic" and knobs.get("syn1_off"):
                W2, B2 = W2.clone(), B2.clone()
                W2[1] = 0.0
                B2[1] = -20.0
            out[f"{lp}.{head}_classifier2.weight"] = W2.numpy().astype(np.float32)
            out[f"{lp}.{head}_classifier2.bias"] = B2.numpy().astype(np.float32)
    out["attn_gain"] = np.float32(knobs["attn_gain"])
    return out


def calibrate(cfg, seed: int, knobs: dict, B: int = 48, rounds: int = 4):
    """Round 0 calibrates on random layouts, later rounds on the layouts the loop itself visits."""
    sd = W.make_state_dict(cfg, seed=seed, bound_preset=False)
    W.apply_attn_gain(sd, knobs["attn_gain"])
    w0 = O.as_torch(sd)
    len_prior, syn_prior = LEN_PRIOR.copy(), SYN_PRIOR.copy()
    len_prior[0], syn_prior[1] = knobs["len0"], knobs["syn1"]
    att = torch.from_numpy(W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=4321))
    mem, sm = O.memory_of(w0, cfg, att)
    lp = "model.length_predictor"
    w = dict(w0)
    out = {}
    if knobs.get("eos_slope"):
        return calibrate_eos(cfg, w0, mem, sm, knobs)
    for r in range(rounds):
        if r == 0:
            feats = row0_features(w, cfg, mem, sm, np.random.Generator(np.random.PCG64(7)))
        else:
            tr = []
            O.core_naic(w, cfg, mem, sm, trace=tr)
            feats = torch.cat([t["out0"][t["active"]] for t in tr], 0)
        for head, prior, live in (("Length", len_prior, [0, 1, 2, 3, 4, 9]), ("Syntactic", syn_prior, [1, 4, 5, 6])):
            hid = F.relu(O.linear(feats, w0, f"{lp}.{head}_classifier1"))
            w2 = w0[f"{lp}.{head}_classifier2.weight"]
            raw = hid @ w2.T
            s = knobs["std"] / float(raw.std(0)[live].mean())
            w[f"{lp}.{head}_classifier2.weight"] = w2 * s
            w[f"{lp}.{head}_classifier2.bias"] = torch.from_numpy(prior) - s * raw.mean(0)
            out[f"{lp}.{head}_classifier2.weight"] = w[f"{lp}.{head}_classifier2.weight"].numpy().astype(np.float32)
            out[f"{lp}.{head}_classifier2.bias"] = w[f"{lp}.{head}_classifier2.bias"].numpy().astype(np.float32)
    out["attn_gain"] = np.float32(knobs["attn_gain"])
    return out


def main():
    import collections
    os.makedirs(os.path.join(os.path.dirname(HERE), "boficap_amd", "presets"), exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    only = set(sys.argv[1:])
    for name, cfg in (("tiny", TINY), ("full", FULL), ("tiny_n2", TINY_N2)):
        if only and name not in only:
            continue
        for seed in (0,):
            heads = calibrate(cfg, seed, KNOBS[name])
            path = os.path.join(os.path.dirname(HERE), "boficap_amd", "presets", f"bound_heads_{name}_seed{seed}.npz")
            np.savez(path, **heads)
            # report the resulting slot statistics on a fresh batch
            sd = W.make_state_dict(cfg, seed=seed, preset_name=name)
            w = O.as_torch(sd)
            att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size))
            mem, sm = O.memory_of(w, cfg, att)
            _, pn, pl, ps, dg = O.core_naic(w, cfg, mem, sm)
            tok = pl.sum(1)
            print(name, "iters", dg["iters"], "mean phrases %.2f" % float(pn.float().mean()),
                  "mean tokens %.1f" % float(tok.float().mean()),
                  dict(collections.Counter(dg["reason"])), "phrase_num hist", np.bincount(pn.numpy()).tolist(),
                  "tokens hist", np.bincount(tok.numpy(), minlength=21).tolist())
            tr = []
            O.core_naic(w, cfg, mem, sm, trace=tr)
            spans = torch.cat([(t["len_logp"][t["active"]].max(1).values - t["len_logp"][t["active"]].min(1).values) for t in tr])
            live = torch.cat([t["len_logp"][t["active"]][:, [0, 1, 2, 3, 4, 9]] for t in tr])
            print(name, "length head log-prob span over all 20 classes: median %.1f; over the live classes: median %.1f, max %.1f"
                  % (float(spans.median()), float((live.max(1).values - live.min(1).values).median()), float((live.max(1).values - live.min(1).values).max())))


if __name__ == "__main__":
    main()
