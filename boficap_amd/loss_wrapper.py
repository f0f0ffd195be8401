"""Host-side mirror of the reference's ``LossWrapper`` for ``train_mode: UIC`` (captioning/modules/loss_wrapper.py:181-244):
same constructor, same ``forward`` signature, same keys in the returned dict, so that ``tools/train.py``'s
``lw_model(fc, att, labels, masks, att_masks, gts, gt_indices, sc_flag, struc_flag, drop_worst_flag, phrase, ...)`` call
(tools/train.py:212-213) works on the drop-in model.  The arithmetic is the HIP path's: ``model(..., mode='forward')`` /
``mode='sample'`` run on libboficap_hip.so, the criteria are the fused kernels and index bookkeeping of ``boficap_amd.xe``.

XE branch (struc_flag False): pinned to the reference by tests/golden/tiny_loss_wrapper_xe (the reference's LossWrapper
around the reference model).  RL branch (struc_flag True): its loss arithmetic is pinned by tests/golden/tiny_rl_loss
(StructureLosses 'new_self_critical' and the rl_kl term with injected samples and scores) and tests/golden/tiny_structure_losses (the other
structure_loss_types); the caption scorer itself
(``get_scores``: CIDEr-D / BLEU of the external ``cider`` and ``coco-caption`` packages, captioning/utils/rewards.py:86-131) is
not part of this build -- pass ``opt.bofi_score_fn(gts, seq) -> [N] floats`` or install one with ``set_scorer``.

RL branch, the estimator (round 6): the reference samples with the autograd tape running (and dropout active) and
differentiates that same pass (loss_wrapper.py:193-209).  In train mode this class does the same by default, as ``XETrainer.rl_step`` does
since round 5 (``reference_estimator_pass`` below): every token of both branches is drawn from the TRAINING forward's rows under
the step's counter-based dropout masks, phrase by phrase, and the returned log-probs are that forward once more with the tape -- at the
drawn tokens they equal the rows they were drawn from bit for bit (``LossWrapper.last_rl["reference_gap"]`` = 0.0).
``opt.bofi_rl_reference_estimator = False`` keeps the fast form of rounds 1-5: samples from the decode engine (no dropout, no tape),
``xe.sampled_logprobs`` recomputes their log-probs with the tape (with dropout): a different estimator, ~2x cheaper.  In eval mode the two
coincide (no dropout anywhere) and the fast form runs.
"""
from __future__ import annotations

import torch

from . import hip, xe

_SCORER = {"fn": None}


def set_scorer(fn) -> None:
    """Install the caption scorer of the RL branch: ``fn(data_gts, gen_result int64 [N, S] on the host) -> [N]`` floats
    (the role of get_scores, captioning/utils/rewards.py:86-131)."""
    _SCORER["fn"] = fn


def reference_estimator_pass(model, att_feats, att_masks, sample_n: int, temperature: float, seed: int):
    """loss_wrapper.py:193-209 (``model(..., mode='sample')`` twice in train mode with the tape, then the loss on THOSE log-probs) on this build's training forward
    -- the eager form of ``XETrainer._rl_reference_step`` (boficap_amd/trainer.py), without a trainer: the caller owns backward and the optimiser.

      * non-autoregressive branch: the engine's bounding loop lays the slots out (greedy heads: one layout per image, no gradient reaches it in the reference either);
        the words of every slot are drawn from the training forward's fill rows;
      * semi-autoregressive branch: the engine's loop runs one iteration per call (bofi_engine_set_saic_range, layout only): its bounding step lays the next phrase
        out on the words so far (TransformerModel.py:1903-1948), a tape-free training forward under ``seed`` gives that phrase's rows, its words are drawn from them
        (Categorical(logits / temperature), CaptionModel.py:405-425) and handed back (bofi_engine_saic_put_words);
      * then the same forward once more WITH the tape: counter-based dropout masks (seed, site, element) make it the same rows bit for bit.
    Returns (saic log-probs [N, S, V] with grad, naic log-probs, saic dict, naic dict, gap) -- ``gap``: max |tape row - drawn row| at the drawn tokens (0.0)."""
    cfg = model.cfg
    S, dev = cfg.seq_length, att_feats.device
    eng = model.engine()
    feats_in, lens_in = model._as_input(att_feats), model._att_len(att_masks)
    N = feats_in.size(0) * sample_n
    if N > model.max_batch:
        raise hip.BofiHipError(f"{N} sampled rows exceed bofi_max_batch={model.max_batch}")
    P = xe.Params(model)

    def rows_of(prep, reuse=None):
        return xe.sampled_logprobs_prepared(P, cfg, att_feats, att_masks, prep, sample_n=sample_n, training=True, seed=seed, compute_dtype=model.train_dtype, reuse=reuse)

    draws = [0]

    def draw(lp, plen, p0, p1, seq, drawn, mask):
        lpf = lp if lp.dtype == torch.float32 and lp.is_contiguous() else lp.float().contiguous()
        n_, s_, V = lpf.shape
        tok = torch.empty(n_, s_, dtype=torch.int64, device=dev)
        draws[0] += 1
        sseed = (seed + 0x5EED0000 + draws[0]) & 0xFFFFFFFFFFFFFFFF
        hip.check(hip.lib().bofi_vocab_sample(hip.ptr(lpf), n_ * s_, V, s_, 1, float(temperature), sseed, None, cfg.pad_idx, hip.ptr(tok), hip.stream_ptr()), "bofi_vocab_sample")
        hip.check(hip.lib().bofi_rl_take_draws(hip.ptr(lpf), hip.ptr(tok), hip.ptr(plen), n_, s_, V, int(p0), int(p1), hip.ptr(seq), hip.ptr(drawn), hip.ptr(mask),
                                               hip.stream_ptr()), "bofi_rl_take_draws")

    with torch.no_grad():
        r = eng.decode_naic(feats_in, lens_in, strict_q1=model.strict_reference)
        na = {k: r[k].repeat_interleave(sample_n, dim=0).contiguous() for k in ("phrase_num", "phrase_length", "phrase_syn")}
        feats_rep = feats_in.repeat_interleave(sample_n, dim=0).contiguous()
        lens_rep = None if lens_in is None else lens_in.repeat_interleave(sample_n).contiguous()
        prep_na = xe.rl_prepare_naic_device(cfg, na["phrase_length"], na["phrase_syn"], strict_q1=model.strict_reference)
        seq_s = torch.zeros(N, S, dtype=torch.int64, device=dev)
        seq_n = torch.zeros(N, S, dtype=torch.int64, device=dev)
        drawn_s, drawn_n = torch.zeros(N, S, device=dev), torch.zeros(N, S, device=dev)
        mask_s = torch.zeros(N, S, dtype=torch.bool, device=dev)
        mask_n = torch.arange(S, device=dev)[None] < na["phrase_length"].long().sum(1)[:, None]
        na_plen = na["phrase_length"].to(torch.int32).contiguous()
        shared: dict = {}                                         # the encoder's memory and the cross K|V: the same tensors in every tape-free forward of this step
        out, passes = None, 0
        for it in range(1, S + 1):
            out = eng.decode_saic(feats_rep, lens_rep, it_range=(it, it), out=out, want_logprob=False, layout_only=True)
            if int(out["bound_iters"]) < it:                      # no caption was open in this iteration: the loop is through
                break
            prep = xe.rl_prepare_saic_device(cfg, seq_s, out["phrase_length"], out["phrase_syn"])
            prep.update(prep_na)
            lp_s, lp_n = rows_of(prep, shared)
            passes += 1
            draw(lp_s, out["phrase_length"], it - 1, it, seq_s, drawn_s, mask_s)
            eng.saic_put_words(seq_s)
            if it == 1:
                draw(lp_n, na_plen, 0, S, seq_n, drawn_n, None)
        prep = xe.rl_prepare_saic_device(cfg, seq_s, out["phrase_length"], out["phrase_syn"])
        prep.update(prep_na)
    lp_saic, lp_naic = rows_of(prep)                              # the gradient pass: the same rows, with the tape
    with torch.no_grad():
        g_s = (lp_saic.detach().float().gather(2, seq_s[..., None]).squeeze(2) - drawn_s)[mask_s]
        g_n = (lp_naic.detach().float().gather(2, seq_n[..., None]).squeeze(2) - drawn_n)[mask_n]
        gap = max(float(g_s.abs().max()) if g_s.numel() else 0.0, float(g_n.abs().max()) if g_n.numel() else 0.0)
    saic = {"seq": seq_s, "phrase_num": out["phrase_num"], "phrase_length": out["phrase_length"], "phrase_syn": out["phrase_syn"]}
    naic = {"seq": seq_n, "phrase_num": na["phrase_num"], "phrase_length": na["phrase_length"], "phrase_syn": na["phrase_syn"]}
    return lp_saic, lp_naic, saic, naic, {"reference_gap": gap, "training_forwards": passes + 1}


class LanguageModelCriterion_UIC(torch.nn.Module):
    """captioning/modules/losses.py:315-369: (loss, SA length, SA phrase, SA syn, NA length, NA phrase, NA syn); with reduction
    'none' the loss is one value per caption and the six parts are None, as in the reference (:357-361)."""

    def forward(self, sa_len, sa_syn, sa_tok, na_len, na_syn, na_tok, phrase_num, phrase_length, phrase_syn, labels, reduction="mean", self_dis=False):
        loss, parts = xe.criterion_uic((sa_len, sa_syn, sa_tok, na_len, na_syn, na_tok), phrase_num, phrase_length, phrase_syn, labels,
                                       reduction=reduction, self_dis=self_dis)
        return (loss, *(parts if parts is not None else [None] * 6))


class StructureLosses(torch.nn.Module):
    """captioning/modules/losses.py:29-179: every ``structure_loss_type`` (configs/uic_sd_kd100_sd_nscl.yml uses 'new_self_critical'),
    the entropy reward and reduction 'none' (xe.structure_loss).  Not built: the self-CIDEr reward term (:167-171, another external scorer)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.loss_type = getattr(opt, "structure_loss_type", "seqnll")

    def forward(self, input, seq, data_gts, reduction="mean"):
        if getattr(self.opt, "self_cider_reward_weight", 0) > 0:
            raise NotImplementedError("the self-CIDEr reward term is not built (0 in the shipped configs)")
        n = input.size(0) // len(data_gts)
        assert n == self.opt.train_sample_n, n                   # losses.py:45
        fn = getattr(self.opt, "bofi_score_fn", None) or _SCORER["fn"]
        if fn is None:
            raise hip.BofiHipError("no caption scorer: the reference's CIDEr-D scorer is an external package (captioning/utils/rewards.py:163-170); "
                                   "pass opt.bofi_score_fn or call boficap_amd.loss_wrapper.set_scorer")
        scores = fn(data_gts, seq.detach().cpu())
        loss, reward = xe.structure_loss(self.loss_type, input, seq, scores, n, reduction=reduction,
                                         entropy_reward_weight=float(getattr(self.opt, "entropy_reward_weight", 0) or 0))
        return {"loss": loss, "reward": reward}


class LossWrapper(torch.nn.Module):
    def __init__(self, model, opt):
        super().__init__()
        self.opt = opt
        self.model = model
        self.train_mode = getattr(opt, "train_mode", "AIC")
        self.self_dis = getattr(opt, "self_dis", False)
        self.rl_kl = getattr(opt, "rl_kl", False)
        if self.train_mode != "UIC":
            raise NotImplementedError(f"train_mode {self.train_mode!r}: the unified bound+fill model (UIC) is built")
        self.crit = LanguageModelCriterion_UIC()
        self.struc_crit = StructureLosses(opt)
        self.last_rl = None                                       # diagnostics of the last struc_flag step: {"reference_gap", "training_forwards"}

    def forward(self, fc_feats, att_feats, labels, masks, att_masks, gts, gt_indices, sc_flag, struc_flag, drop_worst_flag, phrase=None,
                phrase_num=None, phrase_length=None, phrase_syn=None, extend_phrase_syn_seq=None, extend_phrase_seq=None,
                extend_phrase_seq_mask=None, glat_p=0.3):
        opt = self.opt
        reduction = "none" if drop_worst_flag else "mean"         # loss_wrapper.py:39: the caller keeps the best captions (tools/train.py:216-220)
        out = {}
        xe_args = (fc_feats, att_feats, labels, att_masks, phrase_num, phrase_length, phrase_syn, extend_phrase_syn_seq, extend_phrase_seq,
                   extend_phrase_seq_mask)
        if not struc_flag:                                        # loss_wrapper.py:231-244
            outs = self.model(*xe_args, glat_p)
            loss, *parts = self.crit(*outs, phrase_num, phrase_length, phrase_syn, labels, reduction=reduction, self_dis=self.self_dis)
            for k, v in zip(("SA_length_loss", "SA_phrase_loss", "SA_syn_loss", "NA_length_loss", "NA_phrase_loss", "NA_syn_loss"), parts):
                out[k] = v
            out["loss"] = loss
            return out
        # ---- struc_flag (loss_wrapper.py:181-230)
        w = float(opt.structure_loss_weight)
        if w < 1:
            outs = self.model(*xe_args)
            lm_loss = self.crit(*outs, phrase_num, phrase_length, phrase_syn, labels)[0]
        else:
            lm_loss = torch.tensor(0).type_as(fc_feats)
        if w > 0:
            n = int(opt.train_sample_n)
            sopt = {"sample_method": opt.train_sample_method, "beam_size": getattr(opt, "train_beam_size", 1), "output_logsoftmax": 1, "sample_n": n}
            ks = ("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn")
            self.model._step = getattr(self.model, "_step", 0) + 1
            seed = (int(getattr(opt, "seed", 0)) << 32) + self.model._step if self.model.training else None
            if self.model.training and getattr(opt, "bofi_rl_reference_estimator", True) and opt.train_sample_method == "sample":
                # the reference's estimator (loss_wrapper.py:193-209): tokens drawn from the rows this pass differentiates
                lp_s, lp_n, saic, naic, self.last_rl = reference_estimator_pass(self.model, att_feats, att_masks, n, 1.0, seed)      # (temperature 1.0: the reference's sample opts carry none, loss_wrapper.py:194-199)
            else:
                was_training = self.model.training
                self.model.eval()                                 # the sampler is the decode engine (no dropout, no tape)
                with torch.no_grad():
                    saic = dict(zip(ks, self.model(fc_feats, att_feats, att_masks, opt=dict(sopt, train_mode="SAIC"), mode="sample")[:5]))
                    naic = dict(zip(ks, self.model(fc_feats, att_feats, att_masks, opt=dict(sopt, train_mode="NAIC"), mode="sample")[:5]))
                self.model.train(was_training)
                lp_s, lp_n = xe.sampled_logprobs(xe.Params(self.model), self.model.cfg, att_feats, att_masks, saic, naic, sample_n=n,
                                                 strict_q1=self.model.strict_reference, training=self.model.training, seed=seed,
                                                 compute_dtype=self.model.train_dtype)
                self.last_rl = {"reference_gap": None, "training_forwards": 1}
            gts = [gts[_] for _ in gt_indices.tolist()]
            s_loss = self.struc_crit(lp_s, saic["seq"], gts)
            n_loss = self.struc_crit(lp_n, naic["seq"], gts)
            struc_loss = {"loss": s_loss["loss"] + n_loss["loss"], "reward": s_loss["reward"] + n_loss["reward"]}
            loss = ((1 - w) * lm_loss + w * s_loss["loss"]) + ((1 - w) * lm_loss + w * n_loss["loss"])
            if self.rl_kl:                                        # :216-222
                # positions the SAIC sample never emitted hold zero rows in the reference's seq_logprobs (TM:1883); they are
                # masked by (SAIC_seq > 0) in the KL term, so the re-forwarded rows there do not matter
                loss = loss + xe.rl_kl_term(lp_n, lp_s, saic["seq"])
        else:
            raise NotImplementedError("structure_loss_weight 0 under struc_flag leaves the reference's own code with an undefined variable (:213)")
        out["lm_loss"] = lm_loss
        out["struc_loss"] = struc_loss["loss"]
        out["reward"] = struc_loss["reward"]
        out["loss"] = loss
        return out
