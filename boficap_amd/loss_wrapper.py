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

One difference in kind from the reference's RL branch, as in ``XETrainer.rl_step``: the reference samples with the
autograd tape running (and dropout active) and differentiates that same pass; here the samples come from the decode engine
(no tape) and ``xe.sampled_logprobs`` recomputes their log-probs with the tape.
"""
from __future__ import annotations

import torch

from . import hip, xe

_SCORER = {"fn": None}


def set_scorer(fn) -> None:
    """Install the caption scorer of the RL branch: ``fn(data_gts, gen_result int64 [N, S] on the host) -> [N]`` floats
    (the role of get_scores, captioning/utils/rewards.py:86-131)."""
    _SCORER["fn"] = fn


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
            was_training = self.model.training
            self.model.eval()                                     # the sampler is the decode engine (no dropout, no tape)
            with torch.no_grad():
                saic = dict(zip(ks, self.model(fc_feats, att_feats, att_masks, opt=dict(sopt, train_mode="SAIC"), mode="sample")[:5]))
                naic = dict(zip(ks, self.model(fc_feats, att_feats, att_masks, opt=dict(sopt, train_mode="NAIC"), mode="sample")[:5]))
            self.model.train(was_training)
            self.model._step = getattr(self.model, "_step", 0) + 1
            seed = (int(getattr(opt, "seed", 0)) << 32) + self.model._step if self.model.training else None
            lp_s, lp_n = xe.sampled_logprobs(xe.Params(self.model), self.model.cfg, att_feats, att_masks, saic, naic, sample_n=n,
                                             strict_q1=self.model.strict_reference, training=self.model.training, seed=seed,
                                             compute_dtype=self.model.train_dtype)
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
