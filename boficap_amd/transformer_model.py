"""Drop-in module for the reference's ``captioning.models.TransformerModel`` (UIC bound+fill model).

Contract mirrored from the reference:
  * ``TransformerModel(opt)`` reads the same ``opt`` attributes (AttModel.py:56-79,
    TransformerModel.py:1631-1640) and exposes ``vocab, seq_length, ss_prob, d_model, train_mode``;
  * ``state_dict()`` has the reference's 311 entries, same names / shapes / order, so a
    ``model.pth`` written by either side loads into the other with ``strict=True``;
  * ``model(*args, mode='sample'|'forward', **kw)`` dispatches to ``_sample`` / ``_forward``
    (CaptionModel.py:42-46); ``_sample`` returns the reference's 6-tuple (AttModel.py:429).
The arithmetic runs in libboficap_hip.so (HIP, gfx950).  This class holds no torch compute for the
decode path and refuses to run without the HIP library and a HIP device.
"""
from __future__ import annotations

import time
from typing import Optional

import torch
import torch.nn as nn

from . import hip
from .config import BofiConfig
from .weights import schema

bad_endings = ['a', 'an', 'the', 'in', 'for', 'at', 'of', 'with', 'before', 'after', 'on', 'upon', 'near', 'to', 'is', 'are', 'am', 'the']


def _build_param_tree(root: nn.Module, cfg: BofiConfig) -> None:
    """Register parameters/buffers under the reference's dotted names, in the reference's order."""
    for name, shape in schema(cfg).items():
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        if name == "model.pos_embed.pe":
            mod.register_buffer(parts[-1], torch.zeros(shape))
        else:
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))


class TransformerModel(nn.Module):
    def __init__(self, opt, *, compute_dtype: Optional[torch.dtype] = None, max_batch: Optional[int] = None,
                 max_regions: Optional[int] = None, strict_reference: bool = True):
        super().__init__()
        self.opt = opt
        self.cfg = BofiConfig.from_opt(opt)
        cfg = self.cfg
        # attributes callers read (tools/train.py:100,162; utils/misc.py:250; eval_utils.py)
        self.vocab_size, self.tgt_vocab = cfg.vocab_size, cfg.tgt_vocab
        self.seq_length, self.max_length = cfg.seq_length, cfg.seq_length
        self.d_model, self.d_ff, self.h = cfg.d_model, cfg.d_ff, cfg.h
        self.N_enc, self.N_dec, self.N_len = cfg.N_enc, cfg.N_dec, cfg.N_len
        self.train_mode = cfg.train_mode
        self.pad_idx, self.bos_idx, self.eos_idx, self.len_idx = cfg.pad_idx, cfg.bos_idx, cfg.eos_idx, cfg.len_idx
        self.ss_prob = 0.0
        self.vocab = opt.vocab
        self.bad_endings_ix = [int(k) for k, v in self.vocab.items() if v in bad_endings]
        # engine knobs (not part of the reference's opt; read with defaults so a reference opt works)
        self.compute_dtype = compute_dtype or getattr(opt, "bofi_compute_dtype", torch.float32)
        self.max_batch = max_batch or getattr(opt, "bofi_max_batch", 64)
        self.max_regions = max_regions or getattr(opt, "bofi_max_regions", getattr(opt, "max_boxes", 100))
        self.strict_reference = strict_reference       # reproduce quirk Q1 (TransformerModel.py:1872-1873)
        self.train_dtype = getattr(opt, "bofi_train_dtype", torch.float32)   # GEMM operand dtype of the XE step
        _build_param_tree(self, cfg)
        self.reset_parameters()
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------ init / weights
    def reset_parameters(self) -> None:
        """Glorot for matrices (TransformerModel.py:1621-1623), nn.Linear-style biases, LN (1, 0)."""
        from .weights import positional_table
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".a_2"):
                    p.fill_(1.0)
                elif name.endswith(".b_2"):
                    p.zero_()
                elif p.dim() > 1:
                    nn.init.xavier_uniform_(p)
                else:
                    fan_in = dict(self.named_parameters())[name[:-4] + "weight"].shape[1]
                    bound = 1.0 / fan_in ** 0.5
                    p.uniform_(-bound, bound)
            self.model.pos_embed.pe.copy_(torch.from_numpy(positional_table(self.cfg.max_pe, self.cfg.d_model)))

    def _weights_key(self):
        return (sum(p._version for p in self.parameters()) + getattr(self, "_weights_epoch", 0), self.compute_dtype, self.max_batch, self.max_regions,
                next(self.parameters()).device)

    def engine(self):
        """The HIP engine holding this module's current weights (rebuilt when they change)."""
        from .engine import BofiEngine
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise hip.BofiHipError("the model must be on a HIP device (model.cuda()); there is no CPU decode path")
        key = self._weights_key()
        if self._engine is None or self._engine_key != key:
            if self._engine is None or self._engine_key[1:] != key[1:]:
                self._engine = BofiEngine(self.cfg, self.compute_dtype, self.max_batch, self.max_regions, device=dev)
                self._engine.load_state_dict(self.state_dict())
                # mode='sample' synchronises after every decode as the reference does (AttModel.py:337): one decode at a time on the device
                self._engine.set_decodes_in_flight(int(getattr(self.opt, "bofi_decodes_in_flight", 1)))
            else:                                              # same engine, new values: re-pack on the device (no host round trip)
                self._engine.refresh_from_device(dict(self.named_parameters()))
            self._engine_key = key
        return self._engine

    # ------------------------------------------------------------------ reference call convention
    def forward(self, *args, **kwargs):
        mode = kwargs.pop("mode", "forward")                      # CaptionModel.py:42-46
        return getattr(self, "_" + mode)(*args, **kwargs)

    @staticmethod
    def _att_len(att_masks):
        return None if att_masks is None else att_masks.long().sum(1).to(torch.int32).contiguous()

    def _prepare_feature(self, fc_feats, att_feats, att_masks):
        """TransformerModel.py:1674-1679: (fc[...,:0], att[...,:0], memory, att_masks[B,1,R])."""
        eng = self.engine()
        if att_masks is not None:                                 # clip_att, AttModel.py:113-120
            max_len = int(att_masks.long().sum(1).max())
            att_feats = att_feats[:, :max_len].contiguous()
            att_masks = att_masks[:, :max_len].contiguous()
        memory = eng.encode(self._as_input(att_feats), self._att_len(att_masks))
        if att_masks is None:
            att_masks = torch.ones(att_feats.shape[:2], dtype=torch.bool, device=att_feats.device)
        return fc_feats[..., :0], att_feats[..., :0], memory, att_masks.unsqueeze(-2)

    def _as_input(self, att_feats):
        if att_feats.dtype not in (torch.float32, self.compute_dtype):
            att_feats = att_feats.float()
        return att_feats.contiguous()

    def _decode_saic_graphed(self, eng, feats, lens, raw_logits, sample, cap=None):
        """The semi-autoregressive decode enqueues all seq_length iterations (≈60 launches each; iterations past the last live
        one return at once), so launched one by one it is bound by the host's launch rate.  It is replayed as a hipGraph over
        static input / output buffers per batch shape; the caller gets its own copies, as from the reference."""
        key = (tuple(feats.shape), feats.dtype, lens is not None, bool(raw_logits), sample is not None)
        cache = self.__dict__.setdefault("_saic_io", {})
        if key not in cache:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            cache[key] = dict(feats=torch.empty_like(feats), lens=None if lens is None else torch.empty_like(lens), out=None)
        io = cache[key]
        io["feats"].copy_(feats)
        if lens is not None:
            io["lens"].copy_(lens)
        S = self.cfg.seq_length
        cap = None if cap is None or cap >= S else max(1, int(cap))
        io["out"] = eng.decode_saic(io["feats"], io["lens"], raw_logits=raw_logits, sample=sample, graph=True, out=io["out"],
                                    it_range=None if cap is None else (1, cap))
        res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in io["out"].items()}
        if cap is not None:                                     # what saic_finish needs to enqueue the rest of the loop
            res["_capped"] = (eng, key, cap, bool(raw_logits), sample)
        return res

    # ---- the semi-autoregressive loop with fewer iterations enqueued than seq_length (the self-critical step: XETrainer.rl_step).  Every
    # iteration past the last live one returns at once but still costs ~60 launches' dispatch (0.27 ms); captions that end after 12
    # iterations leave 8 of them.  saic_cap() proposes a bound from the recent decodes, saic_finish() looks at the count of live iterations
    # the capped decode reports and, if the loop may not be through, enqueues the REST on the state the engine still holds -- the same
    # computation as the whole loop, exactly (tests/test_gpu_rl.py).
    def saic_cap(self):
        """The budget for the next decode: recent live iterations + 2, in steps of 2 and STICKY -- every distinct value is a graph of its own
        (a capture costs tens of milliseconds): raised as soon as the recent decodes ask for more, lowered only after eight decodes that
        would all have fitted the next lower step.  None: no budget (the first two decodes, or captions that use the whole loop).
        (An enqueued iteration past the last live one is ~40 launches that return at once, ~0.18 ms: steps of 4 cost the self-critical step
        up to 0.7 ms.)"""
        S = self.cfg.seq_length
        recent = self.__dict__.get("_saic_recent")
        if not recent or len(recent) < 2:
            return None
        want = -(-(max(recent) + 2) // 2) * 2
        cur = self.__dict__.get("_saic_cap_cur")
        if cur is None or want > cur or (len(recent) >= 8 and want <= cur - 2):
            cur = self.__dict__["_saic_cap_cur"] = want
        return None if cur >= S else cur

    def saic_finish(self, res):
        """``res``: a result of _decode_saic_graphed.  Returns the complete result (after a device -> host read of the iteration count)."""
        capped = res.pop("_capped", None)
        live = int(res["bound_iters"])
        if capped is not None and live >= capped[2]:
            eng, key, cap, raw_logits, sample = capped
            io = self.__dict__["_saic_io"][key]
            eng.decode_saic(io["feats"], io["lens"], raw_logits=raw_logits, sample=sample, graph=True, out=io["out"], it_range=(cap + 1, self.cfg.seq_length))
            res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in io["out"].items()}
            live = int(res["bound_iters"])
        from collections import deque
        self.__dict__.setdefault("_saic_recent", deque(maxlen=8)).append(live)
        return res

    def sample_pair(self, att_feats, att_masks=None, sample_n=5, temperature=1.0, saic_cap=None):
        """What the self-critical step asks of the model (loss_wrapper.py:193-209): ``sample_n`` sampled captions per image in SAIC
        mode and in NAIC mode.  Same results as two ``mode='sample'`` calls (same seeds); the two decodes are independent, so the
        NAIC one runs on a fork of the engine on a second stream, under the semi-autoregressive loop's replayed graph.
        Returns (saic, naic) dicts with seq, seq_logprob, phrase_num, phrase_length, phrase_syn.  ``saic_cap``: enqueue that many iterations
        of the semi-autoregressive loop only -- the caller then passes ``saic`` through saic_finish() before it reads it."""
        eng = self.engine()
        side = self.__dict__.get("_side_engine")
        if side is None or side[0] is not eng:
            from .engine import pick_stream_beside              # (a stream that overlaps the caller's, found by probing: not any new stream does)
            side = self.__dict__["_side_engine"] = (eng, eng.fork(), pick_stream_beside(torch.cuda.current_stream(eng.device), eng.device))
        _, eng2, s2 = side
        feats, lens = self._as_input(att_feats), self._att_len(att_masks)
        main = torch.cuda.current_stream(feats.device)
        s2.wait_stream(main)                                      # inputs and (re-packed) weights are ready there
        base = int(getattr(self.opt, "seed", 0)) << 32
        self._sample_calls = getattr(self, "_sample_calls", 0) + 2
        seed_saic, seed_naic = base + self._sample_calls - 1, base + self._sample_calls
        rows, rlens = feats, lens
        if sample_n > 1:
            rows = feats.repeat_interleave(sample_n, dim=0).contiguous()
            rlens = None if lens is None else lens.repeat_interleave(sample_n).contiguous()
        if rows.size(0) > self.max_batch:
            raise hip.BofiHipError(f"{rows.size(0)} sampled rows exceed bofi_max_batch={self.max_batch}")
        saic = self._decode_saic_graphed(eng, rows, rlens, False, (float(temperature), seed_saic), cap=saic_cap)
        with torch.cuda.stream(s2):
            r = eng2.decode_naic(feats, lens, strict_q1=self.strict_reference)
            naic = {k: r[k] for k in ("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn")}
            if sample_n > 1:                                      # the bound is deterministic: n identical layouts (models/utils.py:3-14)
                naic = {k: v.repeat_interleave(sample_n, dim=0) for k, v in naic.items()}
            naic["seq"] = eng2.sample_tokens(r, sample_n, float(temperature), seed_naic)
            # the non-autoregressive samples are ready long before the semi-autoregressive decode ends: their host copies leave on THIS stream
            # (pinned memory, an event behind them), so that a caller can score / collate them while the other decode still runs
            host = self.__dict__.setdefault("_naic_host", {})
            for k, v in naic.items():
                hb = host.get(k)
                if hb is None or hb.shape != v.shape or hb.dtype != v.dtype:
                    hb = host[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                hb.copy_(v, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(s2)
        self._naic_ready = (ready, dict(host))
        main.wait_stream(s2)
        for v in list(naic.values()) + [feats] + ([lens] if lens is not None else []):
            v.record_stream(main)                                 # allocated / read on s2, used on the caller's stream from here on
        return saic, naic

    def decode_many(self, batches, *, batches_per_launch: int = 16, in_flight: Optional[int] = None, stats: bool = True, keep_logprob: bool = False):
        """Greedy NAIC decode of MANY loader batches at the engine's throughput -- the eval loop of the reference (eval_utils.py:456-460: one synchronised
        ``model(..., mode='sample')`` per batch, which is what ``_sample`` reproduces) as a pipeline: ``in_flight`` engine forks on overlapping streams,
        ``batches_per_launch`` consecutive batches per launch (quirk Q1 per batch: every batch's result is its own decode's), features copied from (pinned)
        host memory on a copy stream ahead of the launches (``boficap_amd.engine.DecodePipeline``).

        ``batches``: iterable of ``att_feats`` or ``(att_feats, att_masks)`` -- [b, R, F] float32 / compute-dtype tensors or arrays on the host or the device,
        masks [b, R] as the loader builds them (prefix-structured, clipped to the batch's longest image by the caller as AttModel.clip_att does).
        Yields one dict per batch, in order: seq [b, S] int64, phrase_num [b] int32, phrase_length [b, S] int32, phrase_syn [b, S] int64 (+ entropy,
        perplexity [b] as eval_utils.py:463-464 with ``stats``; + seq_logprob [b, S, V] on the device with ``keep_logprob``), host tensors."""
        from .engine import DecodePipeline
        eng = self.engine()
        key = (id(eng), batches_per_launch, in_flight, stats, keep_logprob, self.strict_reference)
        cached = self.__dict__.get("_pipeline")
        if cached is None or cached[0] != key:
            cached = self.__dict__["_pipeline"] = (key, DecodePipeline(eng, in_flight=in_flight, batches_per_launch=batches_per_launch, strict_q1=self.strict_reference,
                                                                       stats=stats, keep_logprob=keep_logprob))

        def items():
            for it in batches:
                att, masks = it if isinstance(it, (tuple, list)) else (it, None)
                att = torch.from_numpy(att) if not torch.is_tensor(att) else att
                if att.dtype not in (torch.float32, self.compute_dtype):
                    att = att.float()
                if masks is None:
                    yield att.contiguous()
                else:
                    masks = torch.from_numpy(masks) if not torch.is_tensor(masks) else masks
                    yield att.contiguous(), masks.long().sum(1).to(torch.int32).contiguous()
        return cached[1].run(items())

    def _sample(self, fc_feats, att_feats, att_masks=None, opt={}):
        """AttModel.py:307-338, 419-437 for train_mode 'NAIC' (bound+fill) and 'SAIC' (phrase by phrase)."""
        sample_method = opt.get("sample_method", "greedy")
        beam_size = opt.get("beam_size", 1)
        sample_n = int(opt.get("sample_n", 1))
        group_size = opt.get("group_size", 1)
        output_logsoftmax = opt.get("output_logsoftmax", 1)
        train_mode = opt.get("train_mode", "AIC")
        if beam_size > 1 or group_size > 1:
            raise NotImplementedError("beam / diverse sampling are AR-only host-side paths (out of scope, SURVEY.md §2 row 10)")
        if train_mode not in ("NAIC", "SAIC"):
            raise NotImplementedError(f"inference mode {train_mode!r}: a UIC model decodes in 'NAIC' or 'SAIC' mode")
        temperature = float(opt.get("temperature", 1.0))
        if sample_method not in ("greedy", "sample"):
            raise NotImplementedError(f"sample_method {sample_method!r}: greedy and 'sample' (Categorical) are built; "
                                      "gumbel / top-k / top-p are autoregressive-path options")
        eng = self.engine()
        torch.cuda.synchronize()                                  # the reference does (AttModel.py:337)
        start = time.time()
        if train_mode == "NAIC":
            # the bounding loop is enqueued for as many iterations as the recent decodes' captions took + 2, not for all seq_length (an
            # iteration past the last live one is five launches that return at once); the count of live iterations tells whether that was
            # enough -- if not, the decode is repeated without the cap, so the result is the reference's either way (opt.bofi_naic_iter_cap:
            # None = adaptive, 0 = never cap, c = always c)
            S = self.cfg.seq_length
            forced = getattr(self.opt, "bofi_naic_iter_cap", None)
            recent = self.__dict__.setdefault("_naic_recent", [])
            cap = int(forced) if forced is not None else (min(S, max(recent) + 2) if len(recent) >= 3 else 0)
            cap = 0 if cap >= S else cap
            feats_in, lens_in = self._as_input(att_feats), self._att_len(att_masks)
            if cap and eng.bound_loop_active(feats_in.size(1)):  # the persistent loop kernel ends by itself and ignores the budget: the first decode IS complete
                cap = 0                                          # (a repeat on live >= cap would be a redundant second decode)
            r = eng.decode_naic(feats_in, lens_in, strict_q1=self.strict_reference, raw_logits=not output_logsoftmax, iter_cap=cap)
            live = int(r["bound_iters"])                          # (a device -> host read: the reference synchronises here too, AttModel.py:337)
            sat = eng.saturated(r)                                # fp16 saturation word of the loop kernel (0 for every model seen so far): fall back to the bf16 iterations
            if sat:
                r = eng._redo_without_loop_kernel(sat, lambda: eng.decode_naic(feats_in, lens_in, strict_q1=self.strict_reference, raw_logits=not output_logsoftmax, out=r, iter_cap=0))
                live = int(r["bound_iters"])
            if cap and live >= cap:
                r = eng.decode_naic(feats_in, lens_in, strict_q1=self.strict_reference, raw_logits=not output_logsoftmax, out=r, iter_cap=0)
                live = int(r["bound_iters"])
            recent.append(live)
            del recent[:-8]
        elif sample_method == "greedy":                           # core_SAIC, AttModel.py:430-437
            # (the loop is enqueued for the iterations recent decodes needed, the rest only if this one needs them: saic_cap / saic_finish)
            adaptive = getattr(self.opt, "bofi_saic_iter_budget", True)
            r = self._decode_saic_graphed(eng, self._as_input(att_feats), self._att_len(att_masks), not output_logsoftmax, None,
                                          cap=self.saic_cap() if adaptive else None)
            r = self.saic_finish(r)
        else:
            # sampled tokens feed the next bound step, so the sample_n copies of an image diverge: decode B * n rows
            # (the reference repeats features and masks the same way, AttModel.py:331-334)
            feats, lens = self._as_input(att_feats), self._att_len(att_masks)
            if sample_n > 1:
                feats = feats.repeat_interleave(sample_n, dim=0).contiguous()
                lens = None if lens is None else lens.repeat_interleave(sample_n).contiguous()
            if feats.size(0) > self.max_batch:
                raise hip.BofiHipError(f"{feats.size(0)} sampled rows exceed bofi_max_batch={self.max_batch}")
            self._sample_calls = getattr(self, "_sample_calls", 0) + 1
            seed = (int(getattr(self.opt, "seed", 0)) << 32) + self._sample_calls
            adaptive = getattr(self.opt, "bofi_saic_iter_budget", True)
            r = self._decode_saic_graphed(eng, feats, lens, not output_logsoftmax, (temperature, seed), cap=self.saic_cap() if adaptive else None)
            r = self.saic_finish(r)
        torch.cuda.synchronize()
        end = time.time()
        if train_mode == "SAIC" and sample_method == "sample":
            return r["seq"], r["seq_logprob"], r["phrase_num"], r["phrase_length"], r["phrase_syn"], end - start
        outs = [r["seq"], r["seq_logprob"], r["phrase_num"], r["phrase_length"], r["phrase_syn"]]
        if sample_n > 1:                                          # the bound is deterministic: n identical layouts (models/utils.py:3-14)
            outs = [o.repeat_interleave(sample_n, dim=0) for o in outs]
        if sample_method == "sample":                             # tokens drawn per copy from the fill distribution (CaptionModel.py:405-425)
            self._sample_calls = getattr(self, "_sample_calls", 0) + 1
            seed = (int(getattr(self.opt, "seed", 0)) << 32) + self._sample_calls
            outs[0] = eng.sample_tokens(r, sample_n, temperature, seed)
        return (*outs, end - start)

    def _forward(self, fc_feats, att_feats, seq, att_masks=None, phrase_num=None, phrase_length=None, phrase_syn=None,
                 extend_phrase_syn_seq=None, extend_phrase_seq=None, extend_phrase_seq_mask=None, glat_p=-1.0):
        """TransformerModel.py:1713-1724,1759-1775 (train_mode 'UIC'; ss_prob > 0: the scheduled-sampling form): the six log-prob tensors, with the
        autograd tape running over the HIP kernels (boficap_amd/xe.py).  float32."""
        from . import xe
        if phrase_num is None:
            raise hip.BofiHipError("the UIC forward needs the phrase tensors of the loader (dataloader.py:343-428)")
        if next(self.parameters()).device.type != "cuda":
            raise hip.BofiHipError("the model must be on a HIP device (model.cuda()); there is no CPU training path")
        self._step = getattr(self, "_step", 0) + 1
        step_word = getattr(self, "_drop_step_word", None)     # set by a graph-capturing trainer: the step lives on the device
        base = int(getattr(self.opt, "seed", 0)) << 32
        seed = (base if step_word is not None else base + self._step) if self.training else None
        if self.ss_prob > 0:                                   # scheduled sampling (tools/train.py:159-162): ss_SAIC for the SA branch
            return xe.forward_uic_ss(xe.Params(self), self.cfg, att_feats, seq, att_masks, phrase_num, phrase_length, phrase_syn,
                                     extend_phrase_syn_seq, ss_prob=float(self.ss_prob), draw=getattr(self, "_ss_draw", None),
                                     training=self.training, seed=seed, compute_dtype=self.train_dtype, step_word=step_word)
        return xe.forward_uic(xe.Params(self), self.cfg, att_feats, seq, att_masks, phrase_num, phrase_length, phrase_syn,
                              extend_phrase_syn_seq, extend_phrase_seq, extend_phrase_seq_mask, glat_p=float(glat_p),
                              training=self.training, seed=seed, compute_dtype=self.train_dtype, step_word=step_word)


def setup(opt):
    """captioning/models/__init__.py:14-24."""
    if getattr(opt, "caption_model", "transformer") != "transformer":
        raise Exception("Caption model not supported: {}".format(opt.caption_model))
    return TransformerModel(opt)
