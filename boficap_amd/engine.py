"""Python handle on the C++/HIP decode engine (``bofi_engine_*`` in include/boficap_hip.h).

One engine = one model replica's packed weights + workspace in the HBM of the current device.
PyTorch is only used here to own device buffers and to name the HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import hip
from .config import BofiConfig


def pick_concurrent_streams(n: int, device=None, candidates: int = 16, spin_cycles: int = 400_000, priorities=None):
    """``n`` HIP streams that really run side by side.

    ROCm multiplexes streams onto a few hardware queues and two streams on one queue serialise; which
    stream lands where depends on creation order.  Instead of guessing, time a spin kernel on pairs of
    candidate streams and keep a set whose members pairwise overlap.  Falls back to fewer streams than
    asked for if no such set exists (the caller then keeps fewer decodes in flight)."""
    import time
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    with torch.cuda.device(dev):
        # (priorities: experiment -- candidate i is created with priorities[i % len]; 0 = default, -1 = high)
        cands = [torch.cuda.Stream(device=dev) if not priorities else torch.cuda.Stream(device=dev, priority=int(priorities[i % len(priorities)])) for i in range(candidates)]

        def span(streams):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin_cycles)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        span(cands[:1])
        one = min(span(cands[:1]) for _ in range(3))
        chosen = [cands[0]]
        for c in cands[1:]:
            if len(chosen) >= n:
                break
            if all(min(span([c, s]) for _ in range(2)) < 1.5 * one for s in chosen):
                chosen.append(c)
    return chosen


def pick_stream_beside(main, device=None, candidates: int = 12, spin_cycles: int = 400_000):
    """A new HIP stream that really runs BESIDE ``main`` (same probe as pick_concurrent_streams: a freshly created stream may share main's
    hardware queue -- which one it gets depends on how many streams the process has made before -- and then serialises behind it).  Falls back
    to the last candidate if none overlaps."""
    import time
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    with torch.cuda.device(dev):
        def span(streams):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin_cycles)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        span([main])
        one = min(span([main]) for _ in range(3))
        cand = None
        for _ in range(candidates):
            cand = torch.cuda.Stream(device=dev)
            if min(span([main, cand]) for _ in range(2)) < 1.5 * one:
                return cand
    return cand


class BofiEngine:
    def __init__(self, cfg: BofiConfig, dtype: torch.dtype = torch.bfloat16, max_batch: int = 64,
                 max_regions: int = 36, device: Optional[torch.device] = None):
        cfg.validate()
        if cfg.d_k != 64:
            raise hip.BofiHipError("the HIP attention kernel is specialised for head dim 64 (d_model / num_att_heads)")
        if not torch.cuda.is_available():
            raise hip.BofiHipError("no HIP device: the decode engine has no CPU fallback")
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.max_batch, self.max_regions = max_batch, max_regions
        self._lib = hip.lib()
        c = hip.BofiConfigC(
            vocab=cfg.tgt_vocab, feat=cfg.att_feat_size, d_model=cfg.d_model, d_ff=cfg.d_ff, heads=cfg.h,
            n_enc=cfg.N_enc, n_dec=cfg.N_dec, seq_length=cfg.seq_length, pad_idx=cfg.pad_idx, bos_idx=cfg.bos_idx,
            eos_idx=cfg.eos_idx, len_idx=cfg.len_idx, head_hidden=cfg.head_hidden, max_batch=max_batch,
            max_regions=max_regions, dtype={torch.float32: hip.DT_F32, torch.bfloat16: hip.DT_BF16}[dtype], n_len=cfg.N_len)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            hip.check(self._lib.bofi_engine_create(C.byref(c), C.byref(self._h)), "bofi_engine_create")
        self._finalized = False

    def fork(self, max_batch: Optional[int] = None) -> "BofiEngine":
        """A second engine sharing this one's weights with its own workspace (one per in-flight batch); ``max_batch``: images per call the fork's
        workspace is sized for (default: this engine's) -- a pipeline that coalesces several loader batches per call forks larger."""
        if not self._finalized:
            raise hip.BofiHipError("fork() needs loaded weights")
        f = object.__new__(BofiEngine)
        f.cfg, f.dtype, f.device, f.max_batch, f.max_regions = self.cfg, self.dtype, self.device, int(max_batch or self.max_batch), self.max_regions
        f._lib, f._finalized, f._parent = self._lib, True, getattr(self, "_parent", self)        # keeps the weights' owner alive
        f._iter_cap, f._q1_group, f._live_word = 0, 0, None           # per-call knobs start from the defaults (bofi_engine_fork resets them too)
        f._h = C.c_void_p()
        with torch.cuda.device(self.device):
            hip.check(self._lib.bofi_engine_fork_sized(self._h, int(max_batch or 0), C.byref(f._h)), "bofi_engine_fork_sized")
        return f

    def stream(self) -> "torch.cuda.Stream":
        """The engine's own HIP stream wrapped for torch (one hardware queue per in-flight engine)."""
        return torch.cuda.ExternalStream(self._lib.bofi_engine_stream(self._h), device=self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.bofi_engine_destroy(h)
            except Exception:
                pass

    # ---------------------------------------------------------------- weights
    def load_state_dict(self, sd: Dict[str, object]) -> None:
        """``sd``: reference-schema state dict (torch tensors on any device, or numpy arrays)."""
        for name, v in sd.items():
            a = v.detach().to("cpu", torch.float32).contiguous().numpy() if torch.is_tensor(v) else np.ascontiguousarray(v, np.float32)
            hip.check(self._lib.bofi_engine_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
                      f"set_weight({name})")
        with torch.cuda.device(self.device):
            hip.check(self._lib.bofi_engine_finalize(self._h), "bofi_engine_finalize")
        self._finalized = True
        self._weights_changed()

    def refresh_from_device(self, named: Dict[str, torch.Tensor]) -> None:
        """Re-pack the weights from float32 tensors on the device (e.g. ``dict(model.named_parameters())``): kernels only,
        existing allocations, captured graphs and forks stay valid (``bofi_engine_refresh_device``)."""
        if not self._finalized:
            raise hip.BofiHipError("the first load goes through load_state_dict")
        items = [(k, v) for k, v in named.items() if k != "model.pos_embed.pe"]
        # (a training loop refreshes after every optimiser step with the same tensors: the checked argument arrays are kept per set of addresses)
        # (the key holds everything the check looks at: the caching allocator hands addresses out again -- tensors of another dtype, size or name at
        # the same addresses must not ride on an old validation)
        key = tuple((k, v.data_ptr(), v.numel(), v.dtype, v.is_cuda, v.is_contiguous()) for k, v in items)
        cached = getattr(self, "_refresh_args", None)
        if cached is None or cached[0] != key:
            for k, v in items:
                if not v.is_cuda or v.dtype != torch.float32 or not v.is_contiguous():
                    raise hip.BofiHipError(f"{k}: contiguous float32 tensor on the HIP device expected")
            n = len(items)
            cached = self._refresh_args = (key, n, (C.c_char_p * n)(*[k.encode() for k, _ in items]), (C.c_void_p * n)(*[v.data_ptr() for _, v in items]),
                                           (C.c_int64 * n)(*[v.numel() for _, v in items]))
        _, n, names, ptrs, numels = cached
        with torch.cuda.device(self.device):
            hip.check(self._lib.bofi_engine_refresh_device(self._h, n, names, ptrs, numels, hip.stream_ptr()), "bofi_engine_refresh_device")
        self._weights_changed()

    def _weights_changed(self):
        if getattr(self, "_loop_mode_sticky", False):            # (a fallback taken because the OLD weights' fp16 copies were clamped: the new ones get their own verdict)
            self._loop_mode_sticky = False
            self.set_bound_loop(-1)

    # ---------------------------------------------------------------- calls
    def _check_feats(self, att_feats, att_len):
        if att_feats.dim() != 3 or att_feats.size(2) != self.cfg.att_feat_size:
            raise hip.BofiHipError(f"att_feats must be [B, R, {self.cfg.att_feat_size}], got {tuple(att_feats.shape)}")
        if not att_feats.is_cuda or not att_feats.is_contiguous():
            raise hip.BofiHipError("att_feats must be a contiguous tensor on the HIP device")
        if att_len is not None and (att_len.dtype != torch.int32 or not att_len.is_cuda or att_len.numel() != att_feats.size(0)):
            raise hip.BofiHipError("att_len must be an int32 device tensor of B elements")

    def decode_naic(self, att_feats: torch.Tensor, att_len: Optional[torch.Tensor] = None, *, strict_q1: bool = True,
                    want_logprob: bool = True, want_memory: bool = False, raw_logits: bool = False, graph: bool = False,
                    refine_rounds: int = 0, out: Optional[dict] = None, q1_group: int = 0, iter_cap: int = 0, phases: str = "", row_stats: bool = False) -> dict:
        """Greedy NAIC bound+fill decode.  Returns a dict of device tensors: seq [B,S] int64,
        seq_logprob [B,S,V] float32 (or None), phrase_num [B] int32, phrase_length [B,S] int32,
        phrase_syn [B,S] int64, bound_iters [1] int32, memory [B,R,d] float32 (or None).
        Pass the previous result as ``out`` to reuse its buffers (required for graph replay).
        ``refine_rounds`` extra filling passes feed the previous ids back as decoder input (BASELINE config 5).
        ``q1_group`` > 0: the call carries B / q1_group independent batches (dynamic batching); quirk Q1 applies per batch,
        so each batch's outputs equal its own separate decode.
        ``iter_cap`` > 0 (the five-launch bounding iterations only; the persistent loop kernel -- ``bound_loop_active(R)`` -- ends by itself, ignores the
        budget and its result is always complete): enqueue that many bounding iterations instead of seq_length (bofi_engine_set_bound_iter_cap); the result is
        then the reference's iff ``bound_iters`` < iter_cap afterwards -- the caller checks and decodes again without the cap otherwise.
        ``phases``: "" = the whole decode; a subset of "ebf" = only the encode / bounding loop / filling pass + export of it (BOFI_FLAG_PHASE_*): a pipelining
        caller enqueues the three on the same engine in that order (its streams / events order them) with the same arguments.
        ``row_stats``: the vocabulary epilogue also leaves, per position, sum_v p log p and the log-prob of the emitted id in ``out['row_plogp']`` / ``out['row_chosen']``
        (float32 [B, S]; bofi_engine_set_row_stats_out) -- what ``row_stats(out)`` / ``entropy_perplexity(out)`` otherwise read back out of the log-prob tensor."""
        self._check_feats(att_feats, att_len)
        B, R, _ = att_feats.shape
        S, V, dev = self.cfg.seq_length, self.cfg.tgt_vocab, att_feats.device
        if out is None:
            out = dict(
                seq=torch.empty(B, S, dtype=torch.int64, device=dev),
                seq_logprob=torch.empty(B, S, V, dtype=torch.float32, device=dev) if want_logprob else None,
                phrase_num=torch.empty(B, dtype=torch.int32, device=dev),
                phrase_length=torch.empty(B, S, dtype=torch.int32, device=dev),
                phrase_syn=torch.empty(B, S, dtype=torch.int64, device=dev),
                bound_iters=torch.empty(1, dtype=torch.int32, device=dev),
                memory=torch.empty(B, R, self.cfg.d_model, dtype=torch.float32, device=dev) if want_memory else None)
        if out.get("bound_saturated") is None:                   # fp16 saturation status of this decode's bounding loop (bofi_engine_set_saturation_out): see ``saturated``
            out["bound_saturated"] = torch.zeros(1, dtype=torch.int32, device=dev)
        if out["bound_saturated"].data_ptr() != getattr(self, "_sat_ptr", 0):
            hip.check(self._lib.bofi_engine_set_saturation_out(self._h, hip.ptr(out["bound_saturated"])), "bofi_engine_set_saturation_out")
            self._sat_ptr = out["bound_saturated"].data_ptr()
        if not 0 <= refine_rounds <= 15:
            raise hip.BofiHipError("refine_rounds must be in 0..15")
        if row_stats and raw_logits:
            raise hip.BofiHipError("row_stats: of log-probs, not of raw logits")
        if row_stats and "row_plogp" not in out:
            out["row_plogp"] = torch.empty(B, S, dtype=torch.float32, device=dev)
            out["row_chosen"] = torch.empty(B, S, dtype=torch.float32, device=dev)
        rs = (out["row_plogp"].data_ptr(), out["row_chosen"].data_ptr()) if row_stats else (0, 0)
        if rs != getattr(self, "_row_stats_ptrs", (0, 0)):
            hip.check(self._lib.bofi_engine_set_row_stats_out(self._h, hip.ptr(out["row_plogp"]) if row_stats else None, hip.ptr(out["row_chosen"]) if row_stats else None),
                      "bofi_engine_set_row_stats_out")
            self._row_stats_ptrs = rs
        out["_row_stats_fused"] = bool(row_stats)
        if q1_group != getattr(self, "_q1_group", 0):
            hip.check(self._lib.bofi_engine_set_q1_group(self._h, int(q1_group)), "bofi_engine_set_q1_group")
            self._q1_group = q1_group
        if int(iter_cap) != getattr(self, "_iter_cap", 0):
            hip.check(self._lib.bofi_engine_set_bound_iter_cap(self._h, int(iter_cap)), "bofi_engine_set_bound_iter_cap")
            self._iter_cap = int(iter_cap)
        flags = ((hip.FLAG_STRICT_Q1 if strict_q1 else 0) | (hip.FLAG_RAW_LOGITS if raw_logits else 0) | (hip.FLAG_GRAPH if graph else 0)
                 | (refine_rounds << hip.FLAG_REFINE_SHIFT))
        if phases:
            if set(phases) - set("ebf"):
                raise hip.BofiHipError("phases: a subset of 'ebf'")
            flags |= ((hip.FLAG_PHASE_ENCODE if "e" in phases else 0) | (hip.FLAG_PHASE_BOUND if "b" in phases else 0) | (hip.FLAG_PHASE_FILL if "f" in phases else 0))
        hip.check(self._lib.bofi_engine_decode_naic(
            self._h, hip.ptr(att_feats), hip.dtype_code(att_feats), hip.ptr(att_len), B, R, flags, hip.ptr(out["seq"]),
            hip.ptr(out["seq_logprob"]), hip.ptr(out["phrase_num"]), hip.ptr(out["phrase_length"]), hip.ptr(out["phrase_syn"]),
            hip.ptr(out["memory"]), hip.ptr(out["bound_iters"]), hip.stream_ptr()), "bofi_engine_decode_naic")
        return out

    def set_decodes_in_flight(self, n: int) -> None:
        """A hint for the kernel choice (bofi_engine_set_decodes_in_flight): 1 = this engine's launches run alone on the device (shorter
        workgroup chains), 0 / > 1 = throughput forms (fewer weight bytes per row).  Part of the graph key: a captured launch is replayed under the hint it was captured with."""
        hip.check(self._lib.bofi_engine_set_decodes_in_flight(self._h, int(n)), "bofi_engine_set_decodes_in_flight")

    def bound_loop_active(self, R: int) -> bool:
        """Does a decode of R regions per image run the bounding loop as the persistent per-16-image kernel (bofi_engine_bound_loop_active)?"""
        return bool(self._lib.bofi_engine_bound_loop_active(self._h, int(R)))

    def set_bound_loop(self, mode: int) -> None:
        """The bounding loop's form for this engine's following decodes (bofi_engine_set_bound_loop): -1 = BOFI_BOUND_LOOP and the decodes-in-flight hint decide,
        0 = the five-launch bf16 iterations, 2 = the persistent fp16 loop kernel whatever the hint."""
        hip.check(self._lib.bofi_engine_set_bound_loop(self._h, int(mode)), "bofi_engine_set_bound_loop")
        self._loop_mode = int(mode)

    def saturated(self, out: dict) -> int:
        """fp16 saturation status of the decode that produced ``out`` (a device -> host read: call it where the caller synchronises anyway): 0 = clean; bit 0 = the
        persistent bounding-loop kernel clamped an activation to +-65 504 on its way into an fp16 MFMA operand; bit 1 = its fp16 weight copies were clamped when they
        were packed; bit 2 = a pair of workgroups that share a group's feed-forward weights lost its exchange (a safety net: never seen).  Either way the slot layout may differ from what bf16 operands (float32's exponent range) give: decode again under ``set_bound_loop(0)``
        -- ``decode_naic_checked`` does both."""
        w = out.get("bound_saturated")
        return int(w) if w is not None else 0

    def decode_naic_checked(self, att_feats, att_len=None, **kw) -> dict:
        """``decode_naic`` + the saturation check + the fallback: when the loop kernel reports a clamp, a warning is issued and the batch is decoded again
        with the five-launch bf16 bounding iterations; a clamp of the WEIGHT copies (bit 1) keeps this engine on that form until its weights change.  Synchronises (reads one word)."""
        r = self.decode_naic(att_feats, att_len, **kw)
        sat = self.saturated(r)
        if sat:
            r = self._redo_without_loop_kernel(sat, lambda: self.decode_naic(att_feats, att_len, **dict(kw, out=r)))
        return r

    def _redo_without_loop_kernel(self, sat: int, decode):
        import warnings
        what = ("clamped its weight copies to +-65 504 (a bounding layer outside fp16's range)" if sat & 2 else
                "clamped an activation to +-65 504 (a bounding layer outside fp16's range)" if sat & 1 else
                "lost a partner workgroup of a pair (exchange timed out: status bit 2)")
        warnings.warn("boficap_amd: the fp16 bounding-loop kernel " + what + "; this decode is repeated with the five-launch bf16 iterations" +
                      (" and the engine stays on them until its weights change" if sat & 2 else ""), RuntimeWarning, stacklevel=3)
        before = getattr(self, "_loop_mode", -1)
        self.set_bound_loop(0)
        try:
            r = decode()
        finally:
            if not sat & 2:
                self.set_bound_loop(before)
            else:
                self._loop_mode_sticky = True
        return r

    def watch_live_iterations(self, word: Optional[torch.Tensor]) -> None:
        """``word`` (int32 [1] on the device, or None to stop): every following decode_naic folds its live-iteration count into it by atomic
        max -- a pipeline of capped decodes (``iter_cap``) is verified with ONE read after the last of them (clear the word first)."""
        if word is not None and (word.dtype != torch.int32 or word.numel() != 1 or not word.is_cuda):
            raise hip.BofiHipError("watch_live_iterations: an int32 [1] device tensor")
        self._live_word = word                                  # (kept alive)
        hip.check(self._lib.bofi_engine_set_live_iterations_max(self._h, hip.ptr(word)), "bofi_engine_set_live_iterations_max")

    def row_stats(self, out: dict):
        """(sum_v p log p, log-prob of the emitted id) per position, float32 [B, S] each, of the decode that produced ``out`` --
        from its seq_logprob tensor, or from the engine's own workspace when that was not materialised."""
        seq = out["seq"]
        B, S = seq.shape
        if out.get("_row_stats_fused"):                          # the decode's own epilogue left them (decode_naic(row_stats=True))
            return out["row_plogp"], out["row_chosen"]
        lp = out.get("seq_logprob")
        src = hip.ptr(lp) if lp is not None else self._lib.bofi_engine_logprob(self._h)
        plogp = torch.empty(B, S, dtype=torch.float32, device=seq.device)
        chosen = torch.empty(B, S, dtype=torch.float32, device=seq.device)
        hip.check(self._lib.bofi_vocab_stats(src, hip.ptr(seq), B * S, self.cfg.tgt_vocab, hip.ptr(plogp), hip.ptr(chosen), hip.stream_ptr()),
                  "bofi_vocab_stats")
        return plogp, chosen

    def entropy_perplexity(self, out: dict, vocab_lower: int = 0):
        """Per-image entropy and perplexity exactly as eval computes them (captioning/utils/eval_utils.py:463-464)."""
        plogp, chosen = self.row_stats(out)
        denom = (out["seq"] > vocab_lower).to(torch.float32).sum(1) + 1
        return -plogp.sum(1) / denom, -chosen.sum(1) / denom

    def sample_tokens(self, out: dict, n: int = 1, temperature: float = 1.0, seed: int = 0) -> torch.Tensor:
        """``n`` sampled captions per image on the slot layout of ``out`` (sample_method 'sample', CaptionModel.py:405-425):
        int64 [B * n, S], image b's draws in rows b*n .. b*n+n-1."""
        seq = out["seq"]
        B, S = seq.shape
        lp = out.get("seq_logprob")
        src = hip.ptr(lp) if lp is not None else self._lib.bofi_engine_logprob(self._h)
        ntok = out["phrase_length"].sum(1).to(torch.int32).contiguous()
        res = torch.empty(B * n, S, dtype=torch.int64, device=seq.device)
        hip.check(self._lib.bofi_vocab_sample(src, B * S, self.cfg.tgt_vocab, S, n, float(temperature), seed & 0xFFFFFFFFFFFFFFFF, hip.ptr(ntok),
                                              self.cfg.pad_idx, hip.ptr(res), hip.stream_ptr()), "bofi_vocab_sample")
        return res

    def decode_saic(self, att_feats: torch.Tensor, att_len: Optional[torch.Tensor] = None, *, raw_logits: bool = False,
                    want_logprob: bool = True, sample: Optional[tuple] = None, graph: bool = False, out: Optional[dict] = None,
                    it_range: Optional[tuple] = None, layout_only: bool = False) -> dict:
        """Semi-autoregressive decode (core_SAIC), greedy or -- ``sample=(temperature, seed)`` -- with every phrase's tokens
        drawn from Categorical(logits / temperature) (the bound heads stay greedy, as in the reference).  Same result
        layout as ``decode_naic``.  ``graph``: the launch sequence is captured once per argument set and replayed (pass the
        previous result as ``out`` and keep the inputs in place); the sampling seed is read from device memory, so replays
        draw anew.  ``it_range`` = (first, last) iterations of the loop (bofi_engine_set_saic_range): (1, c) enqueues c iterations,
        (c + 1, S) with the same arguments and ``out`` continues that decode where it stopped -- together the whole loop, exactly;
        ``out["bound_iters"]`` (live iterations so far) < c says the first part was all of it.  ``layout_only``: the enqueued iterations lay their phrases out and stop
        (BOFI_FLAG_SAIC_LAYOUT_ONLY: no decoder pass, no tokens) -- for a caller that draws the words itself and hands them back with ``saic_put_words``."""
        self._check_feats(att_feats, att_len)
        B, R, _ = att_feats.shape
        S, V, dev = self.cfg.seq_length, self.cfg.tgt_vocab, att_feats.device
        if it_range is not None and (out is None and it_range[0] > 1):
            raise hip.BofiHipError("a continued semi-autoregressive decode needs the first part's `out`")
        if out is None:
            out = dict(
                seq=torch.empty(B, S, dtype=torch.int64, device=dev),
                seq_logprob=torch.empty(B, S, V, dtype=torch.float32, device=dev) if want_logprob else None,
                phrase_num=torch.empty(B, dtype=torch.int32, device=dev),
                phrase_length=torch.empty(B, S, dtype=torch.int32, device=dev),
                phrase_syn=torch.empty(B, S, dtype=torch.int64, device=dev),
                bound_iters=torch.empty(1, dtype=torch.int32, device=dev), memory=None)
        flags = (hip.FLAG_RAW_LOGITS if raw_logits else 0) | (hip.FLAG_GRAPH if graph else 0) | (hip.FLAG_SAIC_LAYOUT_ONLY if layout_only else 0)
        if sample is not None:
            hip.check(self._lib.bofi_engine_set_sampling(self._h, float(sample[0]), int(sample[1]) & 0xFFFFFFFFFFFFFFFF), "bofi_engine_set_sampling")
            flags |= hip.FLAG_SAMPLE
        first, last = it_range if it_range is not None else (1, 0)
        hip.check(self._lib.bofi_engine_set_saic_range(self._h, int(first), int(last) if last < S else 0), "bofi_engine_set_saic_range")
        try:
            hip.check(self._lib.bofi_engine_decode_saic(
                self._h, hip.ptr(att_feats), hip.dtype_code(att_feats), hip.ptr(att_len), B, R, flags,
                hip.ptr(out["seq"]), hip.ptr(out["seq_logprob"]), hip.ptr(out["phrase_num"]), hip.ptr(out["phrase_length"]),
                hip.ptr(out["phrase_syn"]), hip.ptr(out["bound_iters"]), hip.stream_ptr()), "bofi_engine_decode_saic")
        finally:
            self._lib.bofi_engine_set_saic_range(self._h, 1, 0)
        return out

    def saic_put_words(self, seq: torch.Tensor) -> None:
        """Between two partial ``decode_saic`` calls (``it_range``): replace the tokens emitted so far by ``seq`` (int64 [B, S], the layout of
        ``out["seq"]``) -- the loop continues on the caller's words (bofi_engine_saic_put_words)."""
        if seq.dtype != torch.int64 or seq.dim() != 2 or seq.size(1) != self.cfg.seq_length or not seq.is_cuda or not seq.is_contiguous():
            raise hip.BofiHipError("saic_put_words: contiguous int64 [B, seq_length] on the device")
        hip.check(self._lib.bofi_engine_saic_put_words(self._h, hip.ptr(seq), int(seq.size(0)), hip.stream_ptr()), "bofi_engine_saic_put_words")

    def encode(self, att_feats: torch.Tensor, att_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Encoder output (float32 [B, R, d]); also leaves memory + cross K/V in the engine workspace."""
        self._check_feats(att_feats, att_len)
        B, R, _ = att_feats.shape
        mem = torch.empty(B, R, self.cfg.d_model, dtype=torch.float32, device=att_feats.device)
        hip.check(self._lib.bofi_engine_encode(self._h, hip.ptr(att_feats), hip.dtype_code(att_feats), hip.ptr(att_len), B, R,
                                               hip.ptr(mem), hip.stream_ptr()), "bofi_engine_encode")
        return mem

    def bound_step(self, ext_syn: torch.Tensor, last: torch.Tensor, R: int, att_len: Optional[torch.Tensor] = None):
        """One bounding step on the memory of the preceding ``encode``: (len_logp [B,20], syn_logp [B,10])."""
        B = ext_syn.size(0)
        if ext_syn.dtype != torch.int32 or last.dtype != torch.int32 or ext_syn.size(1) != self.cfg.bound_len:
            raise hip.BofiHipError("ext_syn must be int32 [B, S+2] and last int32 [B]")
        llp = torch.empty(B, 20, dtype=torch.float32, device=ext_syn.device)
        slp = torch.empty(B, 10, dtype=torch.float32, device=ext_syn.device)
        hip.check(self._lib.bofi_engine_bound_step(self._h, hip.ptr(ext_syn.contiguous()), hip.ptr(last.contiguous()), B, R,
                                                   hip.ptr(att_len), hip.ptr(llp), hip.ptr(slp), hip.stream_ptr()), "bofi_engine_bound_step")
        return llp, slp

    def fill_naic(self, ext_syn: torch.Tensor, last: torch.Tensor, R: int, att_len: Optional[torch.Tensor] = None, *, strict_q1: bool = True,
                  raw_logits: bool = False, refine_rounds: int = 0):
        """The filling pass alone on a given slot layout (``ext_syn`` int32 [B, S+2], ``last`` int32 [B]) and the memory of the
        preceding ``encode``: (seq int64 [B, S], seq_logprob float32 [B, S, V])."""
        B = ext_syn.size(0)
        if ext_syn.dtype != torch.int32 or last.dtype != torch.int32 or ext_syn.size(1) != self.cfg.bound_len or last.numel() != B:
            raise hip.BofiHipError("ext_syn must be int32 [B, S+2] and last int32 [B]")
        S, V, dev = self.cfg.seq_length, self.cfg.tgt_vocab, ext_syn.device
        seq = torch.empty(B, S, dtype=torch.int64, device=dev)
        lp = torch.empty(B, S, V, dtype=torch.float32, device=dev)
        flags = (hip.FLAG_STRICT_Q1 if strict_q1 else 0) | (hip.FLAG_RAW_LOGITS if raw_logits else 0) | (refine_rounds << hip.FLAG_REFINE_SHIFT)
        hip.check(self._lib.bofi_engine_fill_naic(self._h, hip.ptr(ext_syn.contiguous()), hip.ptr(last.contiguous()), B, R, hip.ptr(att_len), flags,
                                                  hip.ptr(seq), hip.ptr(lp), hip.stream_ptr()), "bofi_engine_fill_naic")
        return seq, lp


class DecodePipeline:
    """Many loader batches through the NAIC bound+fill decode at the engine's throughput: what ``tools/eval.py`` and
    ``TransformerModel.decode_many`` run instead of one synchronised ``mode='sample'`` call per batch (the reference's eval loop,
    eval_utils.py:456-460 under tools/eval.py:123).

      * ``in_flight`` engine forks on streams that provably overlap, one launch each (default 3: with the copy stream that is the runtime's four
        hardware queues -- a copy stream that shares a queue with a launch stream: 161 against 210 k images/s, profiles/r05_decode_many.txt; 4 when the
        process was started with GPU_MAX_HW_QUEUES >= 5).  Launch streams and the copy stream are found by probing for queues of their own;
      * ``batches_per_launch`` (default 10) consecutive loader batches of one shape ride ONE launch (dynamic batching: quirk Q1 stays per batch, so every
        batch's result is its own decode's -- bit for bit under the same kernel family and hint); ragged batches (region counts given) are padded to the
        next of ``region_buckets`` so that differently clipped batches share launches and captured graphs (the padding is masked by the counts);
      * features are double-buffered per fork and copied from (pinned) host memory on a copy stream that runs ahead of the launches; the small
        outputs (ids, slot layout, per-image entropy / perplexity) come back through pinned buffers behind an event, the 48.6 MB of log-probs per
        batch stay on the device unless asked for.

    ``run(batches)`` is a generator: one dict per input batch, in input order, as soon as its launch is through -- while later launches are
    already in flight."""

    def __init__(self, engine: "BofiEngine", *, in_flight: Optional[int] = None, batches_per_launch: int = 16, strict_q1: bool = True, stats: bool = True,
                 keep_logprob: bool = False, region_buckets=(36, 48, 64, 80, 100, 128)):
        if in_flight is None:                                    # 3 launch streams + the copy stream = the runtime's default of 4 hardware queues; a process started
            import os                                            # with GPU_MAX_HW_QUEUES >= 5 (tools/eval.py sets 8) has room for a fourth launch stream
            in_flight = 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) >= 5 else 3
        if in_flight < 1 or batches_per_launch < 1:
            raise hip.BofiHipError("in_flight and batches_per_launch must be >= 1")
        self.root, self.nf, self.bpl = engine, int(in_flight), int(batches_per_launch)
        self.strict_q1, self.stats, self.keep_logprob = strict_q1, stats, keep_logprob
        self.dev = engine.device
        # ragged loader batches (att_masks given: every batch is clipped to ITS longest image, AttModel.py:113-120) are padded to the next of a few region counts, so that
        # consecutive batches share launches and captured graphs; the padding rows are masked by the region counts like any short image's
        self.buckets = sorted({int(r) for r in region_buckets if int(r) <= engine.max_regions} | {int(engine.max_regions)})
        self._slots = None                                       # built for the first launch's shape

    def _build(self, rows_max: int):
        # launch streams AND the copy stream on hardware queues of their own, found by probing (two streams on one queue serialise: a copy stream that
        # shares a queue with a launch stream costs a quarter of the throughput)
        streams = pick_concurrent_streams(self.nf + 1, self.dev, candidates=24)
        self.copy_stream = streams.pop() if len(streams) == self.nf + 1 else torch.cuda.Stream(self.dev)
        while len(streams) < self.nf:                            # (fewer hardware queues than launches in flight: the extra ones share)
            streams.append(torch.cuda.Stream(self.dev))
        self._slots = []
        for k in range(self.nf):
            e = self.root.fork(max_batch=rows_max)
            e.set_decodes_in_flight(self.nf)
            self._slots.append(dict(eng=e, stream=streams[k], feats=[{}, {}], lens=[None, None], out=None, host=None, copied=[torch.cuda.Event(), torch.cuda.Event()],
                                    done=torch.cuda.Event()))
        self.rows_max = rows_max

    @staticmethod
    def _as_host_or_device(x):
        return torch.from_numpy(x) if not torch.is_tensor(x) else x

    def _bucket(self, r: int) -> int:
        for b in self.buckets:
            if b >= r:
                return b
        raise hip.BofiHipError(f"{r} regions exceed the engine's max_regions={self.root.max_regions}")

    def run(self, batches):
        """``batches``: iterable of ``att_feats`` or ``(att_feats, att_len)``: [b, R, F] tensors / arrays in the engine's compute dtype or float32, on
        the host (pinned: the copy is asynchronous) or the device; ``att_len`` int32 [b] region counts or None (every image has R regions)."""
        pending = []                                             # launches in flight: (slot, [batch sizes])
        group, gkey = [], None
        j = 0

        def flush():
            nonlocal j, group
            if not group:
                return
            k, p = j % self.nf, (j // self.nf) % 2
            if j >= self.nf:
                yield from self._finish(pending.pop(0))
            self._launch(k, p, group, gkey)
            pending.append((k, [g[0].shape[0] for g in group]))
            j += 1
            group = []

        for item in batches:
            att, lens = item if isinstance(item, (tuple, list)) else (item, None)
            att = self._as_host_or_device(att)
            lens = None if lens is None else self._as_host_or_device(lens).to(torch.int32)
            b, r = att.shape[0], att.shape[1]
            key = (b, r if lens is None else self._bucket(r), att.shape[2], att.dtype, lens is None)
            if group and (key != gkey or len(group) >= self.bpl):
                yield from flush()
            group.append((att, lens))
            gkey = key
        yield from flush()
        while pending:
            yield from self._finish(pending.pop(0))

    def _launch(self, k, p, group, key):
        b, R, F, dt, no_len = key
        nb, rows = len(group), len(group) * b
        if self._slots is None:
            self._build(max(rows, self.bpl * b))
        if rows > self.rows_max:
            raise hip.BofiHipError(f"a launch of {rows} images exceeds the pipeline's {self.rows_max} (first batch x batches_per_launch)")
        sl = self._slots[k]
        e, st = sl["eng"], sl["stream"]
        has_len = not no_len
        cs = self.copy_stream
        caller = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(cs):
            # (the buffers are created -- and zero-filled -- ON the copy stream: a fill enqueued on the caller's stream would race the copies below)
            buf = sl["feats"][p].get((R, F, dt))
            if buf is None:
                buf = sl["feats"][p][(R, F, dt)] = torch.zeros(self.rows_max, R, F, dtype=dt, device=self.dev)
            if has_len and sl["lens"][p] is None:
                sl["lens"][p] = torch.zeros(self.rows_max, dtype=torch.int32, device=self.dev)
            feats, lens = buf[:rows], (sl["lens"][p][:rows] if has_len else None)
            cs.wait_event(sl["done"])                            # the launch that last read this slot's buffers is through (it was finished before this one is issued)
            for i, (att, ln) in enumerate(group):
                for src in (att, ln):                            # a DEVICE batch was produced on its caller's stream: the copy waits for that stream, and the caching
                    if src is not None and src.is_cuda:          # allocator must not hand the batch's memory out again while the copy is pending
                        cs.wait_stream(caller)
                        src.record_stream(cs)
                feats[i * b:(i + 1) * b, :att.shape[1]].copy_(att, non_blocking=True)      # (a batch clipped below the bucket leaves padding rows: masked by its counts)
                if has_len:
                    lens[i * b:(i + 1) * b].copy_(ln, non_blocking=True)
            sl["copied"][p].record(cs)
        sl["last"] = (feats, lens, b, nb, rows)                  # (kept for the fallback decode of _finish: this slot's buffers stay untouched until its next launch)
        with torch.cuda.stream(st):
            st.wait_event(sl["copied"][p])
            self._decode_and_copy_out(sl)
            sl["done"].record(st)

    def _decode_and_copy_out(self, sl):
        e = sl["eng"]
        feats, lens, b, nb, rows = sl["last"]
        sl["out"] = e.decode_naic(feats, lens, strict_q1=self.strict_q1, graph=True, out=sl["out"] if sl["out"] is not None and sl["out"]["seq"].shape[0] == rows else None,
                                  q1_group=b if nb > 1 else 0, row_stats=self.stats, want_logprob=self.keep_logprob)
        out = sl["out"]
        small = {k2: out[k2] for k2 in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_saturated")}
        if self.stats:
            small["entropy"], small["perplexity"] = e.entropy_perplexity(out)
        if sl["host"] is None or sl["host"]["seq"].shape[0] != rows:
            sl["host"] = {k2: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k2, v in small.items()}
        for k2, v in small.items():
            sl["host"][k2].copy_(v, non_blocking=True)
        sl["lp"] = out["seq_logprob"].clone() if self.keep_logprob else None

    def _finish(self, launch):
        k, sizes = launch
        sl = self._slots[k]
        sl["done"].synchronize()
        sat = int(sl["host"]["bound_saturated"][0])
        if sat:                                                  # the fp16 loop kernel clamped something (BofiEngine.saturated): this launch again with the bf16 iterations

            def again():
                with torch.cuda.stream(sl["stream"]):
                    self._decode_and_copy_out(sl)
                    sl["done"].record(sl["stream"])
                sl["done"].synchronize()
            sl["eng"]._redo_without_loop_kernel(sat, again)
        mine = {k2: v.clone() for k2, v in sl["host"].items() if k2 != "bound_saturated"}      # (one private copy per launch: the pinned buffers are the next launch's; batches are views of it)
        o = 0
        for b in sizes:
            res = {k2: v[o:o + b] for k2, v in mine.items()}
            if sl["lp"] is not None:
                res["seq_logprob"] = sl["lp"][o:o + b]
            o += b
            yield res
