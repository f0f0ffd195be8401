"""XE train step: forward -> criterion -> backward -> gradient all-reduce -> clip -> Noam/Adam step.

Replaces the loop body of the reference's tools/train.py:198-229 and its ``nn.DataParallel`` wrapper (:97-101)
with one process per GPU: all parameters live in ONE flat float32 bucket in HBM (the module's parameters are views
into it), all gradients in a second one, so that the data-parallel exchange is a single RCCL all-reduce over the
flat gradient bucket and the optimiser is a single HBM-bound kernel (``bofi_adam_step``) over four flat streams.
"""
from __future__ import annotations

import re
from typing import Dict, Optional

import os

import torch

from . import hip

ALIGN = 64           # elements: every parameter -- and its bf16 copy (WeightOperands) -- starts on a 128-byte boundary: the GEMM's LDS-DMA reads weights in 128-byte row slabs, a straddling base costs two cache lines per slab (measured: +18% GEMM time)


class FlatBucket:
    """Re-homes the parameters of ``module`` into one flat buffer (values preserved) and gives every parameter a
    ``.grad`` that is a view into a second flat buffer.  Works on any device (the gloo tests use the CPU)."""

    DEAD_PREFIXES = ("model.length_predictor.length_attn.", "model.length_predictor.ff.")

    def __init__(self, module: torch.nn.Module):
        named = list(module.named_parameters())
        if not named:
            raise ValueError("no parameters")
        # q, k, v projections of one attention block next to each other (weights, then biases): the packed [3d, d]
        # operand of the fused projection GEMM and its gradient are then plain views of the buckets (span()).
        pat = re.compile(r"^(.*)\.linears\.([012])\.(weight|bias)$")
        groups, order = {}, []
        for name, p in named:
            m = pat.match(name)
            key = (m.group(1), m.group(3)) if m else name
            if key not in groups:
                groups[key] = []
                order.append(key)
            groups[key].append((name, p))
        # parameters the UIC model never reads (the unused length_attn / ff copies inside LengthPredictor_UIC,
        # TransformerModel.py:350-355; 3.15 M elements at the full size) go LAST: the gradients of everything in front of them
        # -- the live prefix -- are what the data-parallel exchange moves
        dead = lambda key: ((key[0] if isinstance(key, tuple) else key) + ".").startswith(self.DEAD_PREFIXES)
        order = [k for k in order if not dead(k)] + [k for k in order if dead(k)]
        self.names = [n for key in order for n, _ in groups[key]]
        self.params = [p for key in order for _, p in groups[key]]
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, n = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share one device and dtype")
            self.offsets.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        first_dead = next((i for i, nm in enumerate(self.names) if nm.startswith(self.DEAD_PREFIXES)), None)
        self.live_numel = n if first_dead is None else self.offsets[first_dead]
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.grad = torch.zeros(n, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self._offset = {id(p): o for p, o in zip(self.params, self.offsets)}
        module._bucket = self

    def span(self, parts):
        """(view, gradient view) over parameters that are adjacent in the bucket and stack on dim 0, else None."""
        o0 = self._offset.get(id(parts[0]))
        if o0 is None:
            return None
        o = o0
        for p in parts:
            if self._offset.get(id(p)) != o or p.numel() % ALIGN or p.shape[1:] != parts[0].shape[1:]:
                return None
            o += p.numel()
        shape = (sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:])
        return self.flat[o0:o].view(shape).requires_grad_(), self.grad[o0:o].view(shape)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):             # a backward may have replaced a view; put it back
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def all_reduce(self, group=None) -> float:
        """Sum the WHOLE flat gradient bucket over the ranks in one blocking collective; returns the scale that turns the sum
        into the mean of the per-rank gradients (= the reference's loss.mean() over DataParallel replicas, train.py:217).
        The simplest form of the exchange: kept as the reference the chunked / bf16 forms are tested against."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return 1.0
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / dist.get_world_size(group)

    def encoder_end(self) -> int:
        """Offset behind the last parameter of att_embed + encoder (they open the bucket): gradients in [encoder_end, live_numel) are final
        once the backward has passed the encoder's output."""
        enc = ("att_embed.", "model.encoder.")
        k = 0
        while k < len(self.names) and self.names[k].startswith(enc):
            k += 1
        cut = self.offsets[k] if k < len(self.offsets) else self.numel
        # the two-stage backward sends [cut, live_numel) while the encoder's backward still runs: an encoder parameter behind the cut would be
        # exchanged before its gradient exists -- silently.  The bucket order is a property of this build (weights.schema); hold it here.
        if any(n.startswith(enc) for n in self.names[k:]) or not 0 < cut < self.live_numel:
            raise RuntimeError("gradient bucket: att_embed / encoder parameters must open it as one contiguous run (two-stage exchange)")
        return cut

    def exchange_range(self, a: int, b: int, group=None, chunks: int = 2):
        """Start the float32 sum all-reduce of gradient elements [a, b) as ``chunks`` asynchronous collectives (last part first);
        returns [(start, end, work)] -- wait on a work before touching its range.  World size 1: []."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or b <= a:
            return []
        gran = 4096
        per = ((b - a + max(1, chunks) - 1) // max(1, chunks) + gran - 1) // gran * gran
        cuts, o = [], a
        while o < b:
            cuts.append((o, min(o + per, b)))
            o += per
        return [(x, y, dist.all_reduce(self.grad[x:y], op=dist.ReduceOp.SUM, group=group, async_op=True)) for x, y in cuts[::-1]]

    def chunk_bounds(self, chunks: int):
        """The live prefix cut into ``chunks`` near-equal pieces on 16 KiB boundaries: [(start, end)], in bucket order."""
        chunks = max(1, int(chunks))
        gran = 4096
        per = (self.live_numel + chunks - 1) // chunks
        per = (per + gran - 1) // gran * gran
        out, o = [], 0
        while o < self.live_numel:
            out.append((o, min(o + per, self.live_numel)))
            o += per
        return out

    def exchange(self, group=None, chunks: int = 4, wire: Optional[str] = None):
        """The data-parallel exchange of one step over the LIVE prefix of the gradient bucket only, as ``chunks`` collectives.

        Yields (start, end, scale) per chunk as soon as that chunk's reduced gradients are usable, so that the caller can run the
        optimiser on chunk i while chunks i+1.. are still on the wire.  The chunks are started back to back, last part of the
        bucket first (the decoder's and generator's parameters sit at the end of the live prefix and the backward finishes
        them first).  ``wire='bf16'``: the reduction is done by hand as the mesh-direct exchange that matches xGMI's
        point-to-point links -- every rank sends piece r of its bf16-rounded chunk to rank r (all_to_all), sums the pieces it
        received in float32, rounds once and all-gathers the result: half the bytes per link, one bf16 rounding of each
        addend and one of the sum (float32 accumulation in between), and every rank ends with bit-identical gradients.
        World size 1: one (0, live_numel, 1.0)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            yield 0, self.live_numel, 1.0
            return
        world = dist.get_world_size(group)
        scale = 1.0 / world
        bounds = self.chunk_bounds(chunks)[::-1]
        if wire is None:
            works = [dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True) for a, b in bounds]
            for (a, b), w in zip(bounds, works):
                w.wait()
                yield a, b, scale
            return
        if wire != "bf16":
            raise ValueError("wire must be None or 'bf16'")
        for a, b in bounds:
            n = b - a
            piece = (n + world - 1) // world
            send = torch.zeros(piece * world, dtype=torch.bfloat16, device=self.grad.device)
            send[:n].copy_(self.grad[a:b])
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)                  # reduce-scatter, by hand: piece r of every rank -> rank r
            mine = recv.view(world, piece).float().sum(0).to(torch.bfloat16)   # float32 accumulation on receipt
            full = torch.empty(piece * world, dtype=torch.bfloat16, device=self.grad.device)
            dist.all_gather_into_tensor(full, mine, group=group)
            self.grad[a:b].copy_(full[:n])
            yield a, b, scale


class WeightOperands:
    """bf16 GEMM operands of the weights in a FlatBucket, kept current per STEP instead of per use.

    The bf16 training path used to cast every weight once a step (one launch each) and transpose every weight once a step for
    the input-gradient GEMMs (one launch each): ~180 small launches.  Here the optimiser kernel writes a bf16 copy of the whole
    bucket while it updates it (``shadow``; one cast of the bucket whenever somebody else changed the weights, detected through
    the tensors' version counters), forward operands are views of that copy, and the transposed operands of all weights come from
    ONE launch (bofi_transpose_many) at the start of the step.  Which (offset, N, K) matrices the backward asks for is learnt
    from the first step that asks; tables are append-only so that captured step graphs keep valid pointers."""

    def __init__(self, bucket: "FlatBucket"):
        self.bucket = bucket
        self.shadow = torch.empty(bucket.numel, dtype=torch.bfloat16, device=bucket.flat.device)
        self._version = None
        self._tables = []                                       # (buffer, table int64 [n,5], tile_first int32 [n+1], n, tiles)
        self._views = {}                                        # (offset, N, K) -> [K, Np] view of a buffer
        self._pending = []
        self._base = bucket.flat.data_ptr()
        self._bytes = bucket.numel * 4

    def _offset_of(self, t: torch.Tensor, numel: int):
        o = t.data_ptr() - self._base
        if o < 0 or o + numel * 4 > self._bytes or o % 4 or not t.is_contiguous() or t.numel() != numel:
            return None
        return o // 4

    def refresh_if_stale(self) -> None:
        """Outside any graph: re-cast the bucket if anything but the optimiser kernel touched it since the last look."""
        flat = self.bucket.flat
        # in-place edits bump the version counter of the tensor they go through: the parameters' own (load_state_dict, p.mul_())
        # or the flat buffer's; the optimiser kernel works on raw pointers and bumps neither
        stamp = (flat._version, sum(p._version for p in self.bucket.params))
        if self._version != stamp:
            hip.check(hip.lib().bofi_cast_bf16(hip.ptr(flat), self.bucket.numel, hip.ptr(self.shadow), self.bucket.numel, 1, self.bucket.numel,
                                               None, None, 0.0, 0, None, hip.stream_ptr()), "bofi_cast_bf16")
            self._version = stamp

    def launch_transposes(self) -> None:
        for buf, table, first, n, tiles in self._tables:
            hip.check(hip.lib().bofi_transpose_many(hip.ptr(self.shadow), hip.ptr(buf), hip.ptr(table), hip.ptr(first), n, tiles,
                                                    hip.stream_ptr()), "bofi_transpose_many")

    # ---- what xe._operand / xe._transposed ask (None: not a bucket weight, or not known yet -> the per-use path)
    def operand(self, w: torch.Tensor, N: int, K: int):
        if K % 64 or self._version is None:                     # (no copy of the bucket has been made yet)
            return None
        o = self._offset_of(w, N * K)
        return None if o is None else self.shadow[o:o + N * K].view(N, K)

    def transposed(self, w: torch.Tensor, N: int, K: int):
        o = self._offset_of(w, N * K)
        if o is None or self._version is None:
            return None
        hit = self._views.get((o, N, K))
        if hit is None and (o, N, K) not in self._pending:
            self._pending.append((o, N, K))
        return hit

    def end_step(self) -> None:
        """Matrices asked for the first time during this step join the batched launch from the next step on."""
        if not self._pending or torch.cuda.is_current_stream_capturing():
            return
        dev = self.shadow.device
        rows, first, dst, tiles = [], [0], 0, 0
        for o, N, K in self._pending:
            Np = (N + 63) // 64 * 64
            rows.append((o, dst, N, K, Np))
            dst += K * Np
            tiles += ((N + 63) // 64) * ((K + 63) // 64)
            first.append(tiles)
        buf = torch.zeros(dst, dtype=torch.bfloat16, device=dev)
        for (o, d0, N, K, Np) in rows:
            self._views[(o, N, K)] = buf[d0:d0 + K * Np].view(K, Np)
        self._tables.append((buf, torch.tensor(rows, dtype=torch.int64, device=dev), torch.tensor(first, dtype=torch.int32, device=dev),
                             len(rows), tiles))
        self._pending = []


def noam_rate(step: int, d_model: int, factor: float = 1.0, warmup: int = 2000) -> float:
    """NoamOpt.rate, captioning/utils/misc.py:179-185."""
    return factor * (d_model ** -0.5 * min(step ** -0.5, step * warmup ** -1.5))


class XETrainer:
    """One XE optimisation step of the UIC model per ``step()`` call.

    opt attributes read (defaults as the reference's opts.py / uic_sd.yml): ``noamopt`` (True), ``noamopt_warmup``
    (20000), ``noamopt_factor`` (1), ``learning_rate`` (5e-4, used when noamopt is off), ``optim_alpha/beta/epsilon``
    (plain Adam), ``grad_clip_value`` (0.1) with ``grad_clip_mode`` 'value'."""

    def __init__(self, model, opt=None, group=None, graph: bool = False, unpadded: bool = True, prepared_weights: bool = True,
                 streams: bool = False, grouped_dw: bool = True, paired: bool = True):
        """``graph``: capture zero-grad + forward + criterion + backward of a batch signature (shapes, max phrase count,
        GLAT on/off) into a hipGraph on first use and replay it afterwards -- ~1 200 kernel launches and the whole Python /
        autograd dispatch of a step become one graph launch.  Inputs are copied into static buffers, the dropout step lives
        in a device word the kernels read.  The all-reduce and the optimiser kernel stay outside the graph."""
        if next(model.parameters()).device.type != "cuda":
            raise hip.BofiHipError("the model must be on a HIP device; there is no CPU training path")
        opt = opt if opt is not None else model.opt
        g = lambda k, d: getattr(opt, k, d)
        self.model, self.group = model, group
        self.noam = bool(g("noamopt", True))
        self.warmup, self.factor = int(g("noamopt_warmup", 20000)), float(g("noamopt_factor", 1.0))
        self.lr = float(g("learning_rate", 5e-4))
        if self.noam:
            self.beta1, self.beta2, self.eps = 0.9, 0.98, 1e-9          # get_std_opt, misc.py:245-251
        else:
            self.beta1, self.beta2, self.eps = float(g("optim_alpha", 0.9)), float(g("optim_beta", 0.999)), float(g("optim_epsilon", 1e-8))
        self.clip_mode = g("grad_clip_mode", "value")          # opts.py:98-101; tools/train.py:225-226: torch.nn.utils.clip_grad_<mode>_
        if self.clip_mode not in ("value", "norm"):
            raise ValueError(f"grad_clip_mode {self.clip_mode!r}: 'value' or 'norm'")
        self.clip = float(g("grad_clip_value", 0.1))
        self.self_dis = bool(g("self_dis", False))             # + KL(SA || NA) over the token positions (losses.py:336-339; configs uic_sd*)
        self.drop_worst_rate = float(g("drop_worst_rate", 0.2))   # share of captions a drop_worst step leaves out (opts.py:167, train.py:216-220)
        self.bucket = FlatBucket(model)
        self.m = torch.zeros_like(self.bucket.flat)
        self.v = torch.zeros_like(self.bucket.flat)
        self._step = 0
        self.rl_kl = bool(g("rl_kl", False))                   # self-critical step: + KL(SAIC || NAIC) over the SAIC captions' tokens (loss_wrapper.py:216-222)
        self.dp_chunks = int(g("bofi_dp_chunks", 4))           # collectives per step over the live gradient prefix
        self.dp_wire = g("bofi_dp_wire", None)                 # None: float32 all-reduce; 'bf16': mesh-direct bf16 exchange, fp32 accumulation
        # data-parallel: backward in two stages around the encoder's output; the decoder-side gradients (2/3 of the bucket) go on the wire
        # while the encoder's backward runs (float32 wire, value clipping; see step())
        self.dp_overlap = bool(g("bofi_dp_overlap", True))
        self.graph = bool(graph)
        self.unpadded = bool(unpadded)                         # add_token_rows: run the decoder over the captions' real positions only
        # bf16 mode: weight operands of the GEMMs come from a bf16 copy of the bucket the optimiser kernel maintains
        self.ops = WeightOperands(self.bucket) if prepared_weights else None
        self.paired = bool(paired) and self.unpadded            # add_token_rows: SA and NA branch as one batch (xe._forward_paired)
        self.grouped_dw = bool(grouped_dw)                     # bf16 mode: all weight-gradient GEMMs of a step in a few grouped launches
        # ... or, with dw_every > 0, started during the backward every `dw_every` problems on a side stream.  Off by default: measured SLOWER
        # as a branch of the captured step (7.02 ms -> 7.13 / 7.24 / 7.60 ms at 24 / 12 / 6 problems per launch, profiles/r03_xe_dw_aside.txt --
        # every fork / join of the graph costs about 50 us, more than the overlap returns); kept as a tested option
        self.dw_every = int(os.environ.get("BOFI_XE_DW_EVERY", g("bofi_dw_every", 0)))
        self._dw_stream = None
        # the forward's four branches (and with them the backward's) on HIP streams of their own (xe._Fork); needs the bucket-level
        # weight operands: per-use casts / transposes of a weight two branches share would race
        self._side = [torch.cuda.Stream() for _ in range(3)] if streams and self.ops is not None else None
        self.max_graphs = 8                                    # batch signatures (shapes x max phrase count x GLAT rate) kept as graphs
        self._graphs = {}
        self._rl_static = None                                 # (set by _rl_replay: the dict whose "picked_*" buffers the last gradient pass filled)
        self._fwd_calls = 0
        if self.graph:
            dev = self.bucket.flat.device
            self._step_word = torch.zeros(1, dtype=torch.int64, device=dev)      # dropout step, read by the kernels
            model._drop_step_word = self._step_word

    # ------------------------------------------------------------------ pieces (exposed for the tests)
    def rate(self, step: Optional[int] = None) -> float:
        step = self._step if step is None else step
        return noam_rate(step, self.model.d_model, self.factor, self.warmup) if self.noam else self.lr

    _KEYS = ("att_feats", "labels", "phrase_num", "phrase_length", "phrase_syn", "extend_phrase_syn_seq", "extend_phrase_seq",
             "extend_phrase_seq_mask")
    _OPT_KEYS = ("token_rows", "token_labels", "token_weight", "row_start", "row_count", "row_cap", "row_pos",
                 "pair_start", "pair_count", "pair_src", "pair_na", "pair_labels", "pair_w_sa", "pair_w_na",
                 "prep_tok_b", "prep_syn_b", "prep_klen_b", "prep_tok2", "prep_syn2", "prep_pos2", "prep_klen2", "prep_img_start", "prep_img_count")

    def _overlap_on(self) -> bool:
        import torch.distributed as dist
        return (self.dp_overlap and self.dp_wire is None and self.clip_mode == "value" and dist.is_available() and dist.is_initialized()
                and dist.get_world_size(self.group) > 1 and self.model.cfg.N_len == 1 and not getattr(self.model, "ss_prob", 0.0) > 0)

    def forward_backward(self, batch: Dict[str, torch.Tensor], glat_p: float = -1.0, drop_worst: bool = False, between=None):
        """zero-grad, forward, criterion, backward.  Returns (loss, parts) as device scalars.  ``drop_worst``: the loss is the mean of the
        best (1 - drop_worst_rate) captions' own losses (criterion reduction 'none' + top-k, tools/train.py:216-220; parts are empty)."""
        if self.ops is not None and not self._capturing() and self.model.train_dtype == torch.bfloat16:
            self.ops.refresh_if_stale()
        if drop_worst or self.self_dis or self.model.cfg.N_len != 1:
            # these need the dense [N, S, V] log-probs of the two branches side by side (or, for a bounding network of N_len >= 2 layers,
            # whole-sequence bound passes): the reference's own form of the step, run eagerly
            if self.graph and not self._capturing():
                self._fwd_calls += 1
                self._step_word.fill_(self._fwd_calls)
            return self._forward_backward_eager(batch, glat_p, dense="drop_worst" if drop_worst else "plain")
        if self.graph and not self._capturing():
            self._fwd_calls += 1
            self._step_word.fill_(self._fwd_calls)             # outside any graph: every step draws new dropout masks
            # (scheduled sampling decides on the host between its iterations: it cannot live in a captured graph)
            if batch.get("att_masks") is None and batch.get("max_phrase_num") is not None and not getattr(self.model, "ss_prob", 0.0) > 0:
                return self._replay(batch, glat_p, between)
        return self._forward_backward_eager(batch, glat_p, between=between)

    @staticmethod
    def _bucket(v, cap: int, step: int = 4) -> int:
        """Round a batch-dependent upper bound up to a multiple of ``step``: the bounds only have to be upper bounds (what lies
        past the true value is masked), and a handful of distinct values keeps the number of captured step graphs small."""
        return min(cap, (int(v) + step - 1) // step * step)

    @staticmethod
    def _capturing() -> bool:
        return torch.cuda.is_current_stream_capturing()

    def _replay(self, batch, glat_p, between=None):
        S = self.model.cfg.seq_length
        key = (tuple((k, tuple(batch[k].shape), batch[k].dtype) for k in self._KEYS), self._bucket(batch["max_phrase_num"], S + 1),
               self._bucket(batch["max_tokens"], S) if batch.get("max_tokens") else 0, round(float(glat_p), 6),
               self.model.training, self.model.train_dtype, between is not None)
        opt_keys = [k for k in self._OPT_KEYS if batch.get(k) is not None]
        key = key + tuple((k, tuple(batch[k].shape)) for k in opt_keys)
        entry = self._graphs.get(key)
        if entry is None and len(self._graphs) >= self.max_graphs:
            return self._forward_backward_eager(batch, glat_p, between=between)     # every capture pins its activations' pool: bound their number
        layout = batch.get("_blob_layout")
        if entry is None:
            static = {}
            if layout is not None:                             # the small inputs as views of ONE static buffer
                static["_blob"] = batch["_blob"].clone()
                static.update(self._blob_views(static["_blob"], layout))
            static.update({k: batch[k].clone() for k in list(self._KEYS) + opt_keys if k not in static})
            static["_blob_layout"] = layout
            static["max_phrase_num"] = int(batch["max_phrase_num"])
            static["max_tokens"] = batch.get("max_tokens")
            self._forward_backward_eager(static, glat_p)        # warm-up outside the capture (lazy initialisations, allocator)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            g2 = None
            if between is None:
                with torch.cuda.graph(g):
                    loss, parts = self._forward_backward_eager(static, glat_p)
            else:
                # two graphs around the encoder's output: [zero-grad, forward, criterion, backward down to it, decoder-side weight
                # gradients] and [the encoder's backward, its weight gradients]; the caller's `between` runs between their replays
                stage2 = {}
                with torch.cuda.graph(g):
                    loss, parts = self._forward_backward_eager(static, glat_p, between=stage2)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool()):
                    self._backward_stage2(stage2)
            entry = self._graphs[key] = (g, static, loss, parts, g2)
        g, static, loss, parts, g2 = entry
        in_blob = ()
        if layout is not None and os.environ.get("BOFI_XE_BLOB", "1") != "0" and static.get("_blob_layout") == layout and static["_blob"].data_ptr() != batch["_blob"].data_ptr() \
                and all(batch[k].data_ptr() == batch["_blob"].data_ptr() + o for k, o, _, _, _ in layout):
            static["_blob"].copy_(batch["_blob"], non_blocking=True)      # every small input in one copy
            in_blob = {k for k, _, _, _, _ in layout}
        pairs = [(static[k], batch[k]) for k in list(self._KEYS) + opt_keys if k not in in_blob and static[k].data_ptr() != batch[k].data_ptr()]
        if pairs:                                              # one multi-tensor copy instead of ~20 small launches
            torch._foreach_copy_([d for d, _ in pairs], [s_ for _, s_ in pairs], non_blocking=True)
        g.replay()
        if g2 is not None:
            between()
            g2.replay()
        return loss, parts

    def _backward_stage2(self, stage2) -> None:
        """The encoder's part of the backward (see xe.forward_uic, split_memory) and its weight gradients."""
        from . import xe
        armed = self.ops is not None and self.model.train_dtype == torch.bfloat16
        if armed:
            xe._WEIGHTS["provider"] = self.ops
        if self.grouped_dw:
            xe._DEFER["list"] = []
        try:
            self._arm_dw_stream()
            torch.autograd.backward(stage2["memory"], stage2["memory_detached"].grad)
            xe.flush_weight_grads_aside(final=True)
            xe.join_weight_grads()
        finally:
            xe._DEFER["list"] = None
            xe._DEFER["side"] = None
            xe._DEFER["held"] = []
            if armed:
                xe._WEIGHTS["provider"] = None
        stage2.clear()

    def _forward_backward_eager(self, batch: Dict[str, torch.Tensor], glat_p: float = -1.0, dense=None, between=None):
        """``between``: None = one backward; a callable = two-stage backward with ``between()`` called after the decoder-side gradients
        are final (eager mode); a dict = stage 1 only, the split tensors are left in it (the captured form: _replay runs stage 2)."""
        from . import xe
        self.bucket.zero_grad()
        armed = self.ops is not None and self.model.train_dtype == torch.bfloat16
        if armed:
            if not self._capturing():
                self.ops.refresh_if_stale()
            self.ops.launch_transposes()
            xe._WEIGHTS["provider"] = self.ops
        if self.grouped_dw:
            xe._DEFER["list"] = []                             # weight gradients: grouped launches beside / after backward
            self._arm_dw_stream()
        try:
            stage2 = between if isinstance(between, dict) else ({} if between is not None and dense is None else None)
            out = self._forward_backward_armed(batch, glat_p, dense, stage2)
            xe.flush_weight_grads_aside(final=True)
            xe.join_weight_grads()
            if stage2 is not None and not isinstance(between, dict):
                between()
                if armed:
                    xe._WEIGHTS["provider"] = None
                xe._DEFER["list"] = None
                self._backward_stage2(stage2)
            elif between is not None and not isinstance(between, dict):
                between()                                      # (a dense step has no split: the whole gradient is final here)
            return out
        finally:
            xe._DEFER["list"] = None
            xe._DEFER["side"] = None
            xe._DEFER["held"] = []
            if armed:
                xe._WEIGHTS["provider"] = None
                self.ops.end_step()

    def _arm_dw_stream(self) -> None:
        """Weight-gradient launches go beside the backward on a stream of their own, `dw_every` problems at a time (xe._DEFER)."""
        from . import xe
        if self.dw_every > 0 and self.grouped_dw:
            if self._dw_stream is None:
                self._dw_stream = torch.cuda.Stream()
            xe._DEFER["side"], xe._DEFER["every"], xe._DEFER["held"] = self._dw_stream, self.dw_every, []

    def _forward_backward_armed(self, batch, glat_p, dense=None, stage2=None):
        from . import xe
        if stage2 is not None:
            xe.HINTS["split_memory"] = stage2
        fc = batch.get("fc_feats")
        if fc is None:
            fc = torch.zeros(batch["att_feats"].shape[0], 0, device=batch["att_feats"].device)
        if batch.get("max_phrase_num") is not None:            # known on the host since the collate: spares the forward a device read
            xe.HINTS["max_phrase_num"] = self._bucket(batch["max_phrase_num"], self.model.cfg.seq_length + 1)
        if batch.get("max_tokens") is not None:                # dynamic padding: decoder positions past the longest caption are skipped
            xe.HINTS["max_tokens"] = self._bucket(batch["max_tokens"], self.model.cfg.seq_length)
        compact = batch.get("token_rows") is not None and batch.get("max_tokens") is not None
        if getattr(self.model, "ss_prob", 0.0) > 0 or dense:   # scheduled sampling: the SA branch follows the model's own layout, not the
            compact = False                                    # loader's -- no row lists, the reference's dense criterion (as self_dis / drop_worst)
            xe.HINTS.pop("max_phrase_num", None); xe.HINTS.pop("max_tokens", None)
        if compact:                                            # project only the real tokens' rows onto the vocabulary
            xe.HINTS["token_rows"] = batch["token_rows"]
            if batch.get("row_cap") is not None:               # ... and run the decoder on those rows only
                # add_token_rows pads the list to a multiple of 256: fewer than 256 rows at its end belong to no caption
                xe.HINTS["unpadded"] = (batch["row_start"], batch["row_count"], batch["row_cap"], batch["row_pos"], 256)
                if self._side is not None and xe._WEIGHTS["provider"] is not None and self.ops._tables and not self.ops._pending:
                    xe.HINTS["streams"] = self._side
        paired = compact and batch.get("row_cap") is not None and batch.get("pair_src") is not None
        if paired:
            xe.HINTS["paired"] = (batch["pair_start"], batch["pair_count"], batch["pair_src"], batch["pair_na"], 256)
            xe.HINTS["pick_labels"] = batch["pair_labels"]      # the criterion's token labels: picked inside the forward
            if batch.get("prep_tok2") is not None and batch.get("att_masks") is None and glat_p < 0:
                xe.HINTS["paired_inputs"] = {k[5:]: batch[k] for k in self._OPT_KEYS if k.startswith("prep_")}
            xe.HINTS.pop("streams", None)
        outs = self.model(fc, batch["att_feats"], batch["labels"], batch.get("att_masks"), batch["phrase_num"], batch["phrase_length"],
                          batch["phrase_syn"], batch["extend_phrase_syn_seq"], batch["extend_phrase_seq"], batch["extend_phrase_seq_mask"],
                          glat_p)
        if paired:
            loss, parts = xe.criterion_uic_compact(outs, batch["phrase_num"], batch["phrase_length"], batch["phrase_syn"],
                                                   batch["pair_labels"], (batch["pair_w_sa"], batch["pair_w_na"]))
        elif compact:
            loss, parts = xe.criterion_uic_compact(outs, batch["phrase_num"], batch["phrase_length"], batch["phrase_syn"],
                                                   batch["token_labels"], batch["token_weight"])
        elif dense == "drop_worst":
            per_cap, _ = xe.criterion_uic(outs, batch["phrase_num"], batch["phrase_length"], batch["phrase_syn"], batch["labels"], reduction="none",
                                          self_dis=self.self_dis)
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            if world > 1:
                # the reference gathers every replica's per-caption losses and takes ONE top-k over the whole batch (tools/train.py:216-220): the k-th
                # smallest loss of all ranks is the threshold here, each rank keeps its captions up to it, and the loss is scaled so that the
                # rank-averaged gradient is that of mean(global top-k)
                allc = [torch.empty_like(per_cap) for _ in range(world)]
                dist.all_gather(allc, per_cap.detach())
                allc = torch.cat(allc)
                keep = int(allc.numel() * (1 - self.drop_worst_rate))
                thr = torch.topk(allc, k=keep, largest=False)[0][-1]
                loss, parts = (per_cap * (per_cap.detach() <= thr)).sum() * (world / keep), []
            else:
                keep = int(per_cap.shape[0] * (1 - self.drop_worst_rate))
                loss, parts = torch.topk(per_cap, k=keep, largest=False)[0].mean(), []
        else:
            loss, parts = xe.criterion_uic(outs, batch["phrase_num"], batch["phrase_length"], batch["phrase_syn"], batch["labels"],
                                           self_dis=self.self_dis)
        loss.backward()
        if self._side is not None:
            # the kernels accumulate parameter gradients themselves, so autograd sees no leaf on the side streams and does not
            # join them at the end of backward(): do it here (the optimiser, or the end of a graph capture, comes next)
            main = torch.cuda.current_stream()
            for s_ in self._side:
                main.wait_stream(s_)
        return loss.detach(), [p.detach() for p in parts]

    def add_token_rows(self, batch: Dict[str, torch.Tensor], host_batch) -> Dict[str, torch.Tensor]:
        """Adds ``token_rows`` / ``token_labels`` / ``token_weight`` (see xe.HINTS) to a device batch from the host copy the
        collate produced: the flat (caption, position) indices of the real tokens under THIS trainer's bucketing of
        ``max_tokens``, zero-padded to a multiple of 256 so that a handful of lengths covers a data stream."""
        import numpy as np
        S = self.model.cfg.seq_length
        pl = np.asarray(host_batch["phrase_length"]).reshape(-1, S + 2)
        labels = np.asarray(host_batch["labels"]).reshape(-1, S + 2)
        ntok = pl.sum(1) - 1
        Sd = self._bucket(int(ntok.max()), S)
        n_idx = np.repeat(np.arange(len(ntok)), ntok)
        t_idx = np.concatenate([np.arange(k) for k in ntok]) if len(ntok) else np.zeros(0, np.int64)
        T = len(n_idx)
        Tp = max(256, (T + 255) // 256 * 256)
        rows, lab, w = np.zeros(Tp, np.int64), np.zeros(Tp, np.int64), np.zeros(Tp, np.float32)
        rows[:T], lab[:T], w[:T] = n_idx * Sd + t_idx, labels[n_idx, 1 + t_idx], 1.0
        dev = batch["att_feats"].device
        out = dict(batch)
        host = {}                                                  # every index tensor of the step, uploaded as ONE blob at the end
        put = lambda **kw: host.update(kw)
        out.update(max_tokens=int(ntok.max()))
        put(token_rows=rows, token_labels=lab, token_weight=w)
        if self.unpadded:                                          # the decoder itself runs over these rows only (xe._fill_unpadded)
            cap, pos = np.zeros(Tp, np.int64), np.zeros(Tp, np.int64)
            cap[:T], pos[:T] = n_idx, t_idx
            start = np.concatenate([[0], np.cumsum(ntok)[:-1]]).astype(np.int32)
            put(row_start=start, row_count=ntok.astype(np.int32), row_cap=cap, row_pos=pos)
        n_img = int(batch["att_feats"].shape[0])
        if self.paired and len(ntok) % n_img == 0:
            # both branches' rows as one list, captions image-major: image i's SA copies, then its NA copies
            N, spi = len(ntok), len(ntok) // n_img
            pn = np.arange(2 * N)
            j = pn % (2 * spi)
            cap_n, cap_na = (pn // (2 * spi)) * spi + j % spi, j >= spi
            cnt = ntok[cap_n]
            T2 = int(cnt.sum())
            T2p = max(256, (T2 + 255) // 256 * 256)
            src, na = np.zeros(T2p, np.int64), np.zeros(T2p, bool)
            src[:T2] = np.concatenate([start[n] + np.arange(ntok[n]) for n in cap_n]) if T2 else 0
            na[:T2] = np.repeat(cap_na, cnt)
            lab2, w2 = lab[src], w[src]
            w2[T2:] = 0.0
            pstart = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
            t = lambda a: a
            put(pair_start=t(pstart), pair_count=t(cnt.astype(np.int32)), pair_src=t(src), pair_na=t(na), pair_labels=t(lab2),
                pair_w_sa=t(w2 * ~na), pair_w_na=t(w2 * na))
            if all(k in host_batch for k in ("phrase_num", "extend_phrase_syn_seq", "extend_phrase_seq", "extend_phrase_seq_mask")) \
                    and batch.get("max_phrase_num") is not None:
                # ... and every index tensor the paired forward would otherwise derive on the device (xe._forward_paired): the bound
                # passes' inputs and key counts, the decoder rows' token / label ids, positions and key counts
                cfg = self.model.cfg
                L = S + 2
                pnum = np.asarray(host_batch["phrase_num"]).reshape(-1)
                esyn = np.asarray(host_batch["extend_phrase_syn_seq"]).reshape(N, L)
                eseq = np.asarray(host_batch["extend_phrase_seq"]).reshape(N, -1)
                emask = np.asarray(host_batch["extend_phrase_seq_mask"]).reshape(N, eseq.shape[1], -1)
                Pm = self._bucket(int(batch["max_phrase_num"]), S + 1)
                idx = np.arange(L)[None]
                cum = 1 + np.where((idx >= 1) & (idx < pnum[:, None]), pl, 0).cumsum(1)       # xe.bound_pass_klen
                klen_pass, last = cum[:, :Pm], cum[:, -1]
                word = labels.copy()
                word[:, 0] = cfg.len_idx
                none = np.full((2 * N, L), -1, np.int64)
                rc, rp = cap[:T], pos[:T]                                                # the single list: caption and position per row
                syn_c = np.zeros(Tp, np.int64); seq_c = np.zeros(Tp, np.int64); k_sa = np.zeros(Tp, np.int32); k_na = np.zeros(Tp, np.int32)
                syn_c[:T], seq_c[:T] = esyn[rc, rp + 1], eseq[rc, rp]
                k_sa[:T], k_na[:T] = emask.sum(-1)[rc, rp], (last - 1)[rc]
                syn_c[T:], seq_c[T:], k_sa[T:], k_na[T:] = esyn[0, 1], eseq[0, 0], emask.sum(-1)[0, 0], (last - 1)[0]   # (padding rows read row 0, as on the device)
                pos_all = np.zeros(Tp, np.int64); pos_all[:T] = rp
                put(prep_tok_b=t(np.where(cap_na[:, None], none, word[cap_n])), prep_syn_b=t(np.where(cap_na[:, None], esyn[cap_n], none)),
                    prep_klen_b=t(klen_pass[cap_n].astype(np.int32)),
                    prep_tok2=t(np.where(na, cfg.bos_idx, seq_c[src])), prep_syn2=t(syn_c[src]), prep_pos2=t(pos_all[src]),
                    prep_klen2=t(np.where(na, k_na[src], k_sa[src]).astype(np.int32)),
                    prep_img_start=t(pstart.reshape(-1, 2 * spi)[:, 0].astype(np.int32)),
                    prep_img_count=t(cnt.reshape(-1, 2 * spi).sum(1).astype(np.int32)))
        # the loader's own label tensors ride in the blob too when the host copy matches what is on the device
        for k in self._KEYS[1:]:
            if k in host_batch and k in batch and torch.is_tensor(batch[k]):
                a = np.asarray(host_batch[k])
                if tuple(a.shape) == tuple(batch[k].shape) and torch.from_numpy(a[:0].copy()).dtype == batch[k].dtype:
                    host[k] = a
        out.update(self._upload_blob(host, dev))
        return out

    @staticmethod
    def _upload_blob(host, dev):
        """One host->device copy for all of a step's small input tensors: they become typed views of one byte buffer (16-byte
        aligned pieces), and a captured step refreshes its static inputs with ONE device copy of that buffer instead of one per
        tensor (``_blob`` / ``_blob_layout`` travel with the batch)."""
        import numpy as np
        layout, off = [], 0
        arrays = {k: np.ascontiguousarray(v) for k, v in host.items()}
        for k, a in arrays.items():
            layout.append((k, off, a.nbytes, torch.from_numpy(a[:0].copy()).dtype, tuple(a.shape)))
            off = (off + a.nbytes + 15) & ~15
        blob_h = np.zeros(max(off, 16), np.uint8)
        for (k, o, nb, _, _), a in zip(layout, arrays.values()):
            blob_h[o:o + nb] = a.reshape(-1).view(np.uint8)
        blob = torch.from_numpy(blob_h).to(dev)
        out = XETrainer._blob_views(blob, layout)
        out["_blob"], out["_blob_layout"] = blob, tuple(layout)
        return out

    @staticmethod
    def _blob_views(blob, layout):
        return {k: blob[o:o + nb].view(dt).view(shape) for k, o, nb, dt, shape in layout}

    def _clip_by_norm(self, grad_scale: float) -> None:
        """clip_grad_norm_ (tools/train.py:225-226 with grad_clip_mode 'norm') on the averaged gradient, without a host round trip: the
        live gradients are scaled in place by min(1, clip / (||g||_2 + 1e-6)); the optimiser kernel then runs without its value clip."""
        g = self.bucket.grad[:self.bucket.live_numel]
        if grad_scale != 1.0:
            g.mul_(grad_scale)
        coef = (self.clip / (torch.linalg.vector_norm(g) + 1e-6)).clamp(max=1.0)
        g.mul_(coef)

    def optimizer_step(self, grad_scale: float = 1.0) -> float:
        self._step += 1
        lr = self.rate()
        if self.clip_mode == "norm" and self.clip != 0:
            self._clip_by_norm(grad_scale)
            grad_scale = 1.0
        self._adam_range(0, self.bucket.numel, lr, grad_scale)
        self.model._weights_epoch = getattr(self.model, "_weights_epoch", 0) + 1     # the decode engine repacks on next use
        return lr

    def _adam_range(self, a: int, b_: int, lr: float, grad_scale: float) -> None:
        """Averaging + clip by value + Adam (+ the bf16 copy of the weights) on elements [a, b) of the four flat streams."""
        import ctypes as C
        b = self.bucket
        shadow = self.ops.shadow if self.ops is not None and self.ops._version is not None else None
        at = lambda t, es: None if t is None else C.c_void_p(t.data_ptr() + a * es)
        clip = self.clip if self.clip_mode == "value" else 0.0          # (norm mode: the gradients were scaled before the kernel)
        hip.check(hip.lib().bofi_adam_step(at(b.flat, 4), at(b.grad, 4), at(self.m, 4), at(self.v, 4), at(shadow, 2), b_ - a, lr,
                                           self.beta1, self.beta2, self.eps, self._step, clip, grad_scale, hip.stream_ptr()),
                  "bofi_adam_step")

    def reduce_and_step(self) -> float:
        """The tail of a data-parallel step: the gradient exchange over the live prefix in chunks, the optimiser kernel on each
        chunk as it arrives (the later chunks are still on the wire meanwhile), then the parameters no rank has a gradient for
        (they see a zero gradient: Adam leaves them where they are, as in the reference where their .grad stays None)."""
        self._step += 1
        lr = self.rate()
        if self.clip_mode == "norm" and self.clip != 0:        # the norm is a property of the WHOLE averaged gradient: every chunk first
            ranges = list(self.bucket.exchange(self.group, self.dp_chunks, self.dp_wire))
            self._clip_by_norm(ranges[0][2] if ranges else 1.0)
            ranges = [(a, b_, 1.0) for a, b_, _ in ranges]
        else:
            ranges = self.bucket.exchange(self.group, self.dp_chunks, self.dp_wire)
        for a, b_, scale in ranges:
            self._adam_range(a, b_, lr, scale)
        self.model._weights_epoch = getattr(self.model, "_weights_epoch", 0) + 1
        return lr

    def _step_overlapped(self, batch, glat_p):
        """The data-parallel step with the exchange started INSIDE the backward: once the backward has passed the encoder's output the
        gradients of everything behind it (decoder, generator, embeddings, bounding network: the bucket from encoder_end() on) are
        final and go on the wire as asynchronous all-reduces, while the encoder's backward runs; the encoder's third follows, and
        the optimiser runs per chunk as it arrives.  Same gradients and parameters as the plain step (tests/test_gpu_dp.py)."""
        import torch.distributed as dist
        b = self.bucket
        cut, late = b.encoder_end(), []
        world = dist.get_world_size(self.group)

        def between():
            late.extend(b.exchange_range(cut, b.live_numel, self.group, max(1, self.dp_chunks - 1)))
        loss, parts = self.forward_backward(batch, glat_p, between=between)
        if not late:                                           # (a path without the split, e.g. no graph for this signature and a dense step)
            between()
        early = b.exchange_range(0, cut, self.group, 1)
        self._step += 1
        lr = self.rate()
        for x, y, work in late + early:
            work.wait()
            self._adam_range(x, y, lr, 1.0 / world)
        self.model._weights_epoch = getattr(self.model, "_weights_epoch", 0) + 1
        return loss, parts

    # ------------------------------------------------------------------ the step
    def step(self, batch: Dict[str, torch.Tensor], glat_p: float = -1.0, drop_worst: bool = False):
        """Returns (loss, parts) as device scalars of THIS rank's shard (no host sync inside)."""
        if self._overlap_on() and not drop_worst and not self.self_dis:
            return self._step_overlapped(batch, glat_p)
        loss, parts = self.forward_backward(batch, glat_p, drop_worst)
        self.reduce_and_step()
        return loss, parts

    # ------------------------------------------------------------------ self-critical step (loss_wrapper.py:181-230, structure_loss_weight 1)
    def rl_step(self, att_feats, att_masks, score_fn, *, sample_n: int = 5, temperature: float = 1.0):
        """One self-critical step of a UIC model: sample ``sample_n`` captions per image in SAIC and in NAIC mode on the decode
        engine (no tape), score them with ``score_fn(seq int64 [N, S] on the host) -> [N] floats`` (the external CIDEr-D scorer of
        captioning/utils/rewards.py in the reference), recompute the samples' log-probs with the tape
        (``xe.sampled_logprobs``), loss = new_self_critical(SAIC) + new_self_critical(NAIC), backward, all-reduce, Adam.
        Returns (loss, mean SAIC score, mean NAIC score)."""
        from . import xe
        model = self.model
        # round 5: the REFERENCE's estimator is the default (every token drawn from the rows the gradient pass differentiates, loss_wrapper.py:193-209;
        # _rl_reference_step below); opt.bofi_rl_reference_estimator = False selects the fast form of rounds 1-4 (samples from the dropout-free inference
        # engine, gradient pass with dropout: 11.5 against ~50 ms per step at 10 x 5, an importance-weight log-std of ~0.65 per caption between the two)
        if getattr(model.opt, "bofi_rl_reference_estimator", True):
            return self._rl_reference_step(att_feats, att_masks, score_fn, sample_n, temperature)
        fc = torch.zeros(att_feats.shape[0], 0, device=att_feats.device)
        was_training = model.training
        model.eval()                                           # sampling runs on the inference engine (no dropout)
        with torch.no_grad():
            if getattr(model.opt, "bofi_rl_sample_pair", True):
                # the two modes' decodes overlap; the semi-autoregressive loop enqueues as many iterations as recent steps needed + 2
                # (model.saic_finish below enqueues the rest if this step's captions run longer: exact either way)
                adaptive = bool(getattr(model.opt, "bofi_rl_saic_adaptive", True)) and os.environ.get("BOFI_RL_SAIC_ADAPTIVE", "1") != "0"
                saic, naic = model.sample_pair(att_feats, att_masks, sample_n, temperature, saic_cap=model.saic_cap() if adaptive else None)
            else:                                              # the reference's two calls, one after the other
                opt = {"sample_method": "sample", "sample_n": sample_n, "temperature": temperature, "output_logsoftmax": 1}
                ks = ("seq", "seq_logprob", "phrase_num", "phrase_length", "phrase_syn")
                saic = dict(zip(ks, model(fc, att_feats, att_masks, opt=dict(opt, train_mode="SAIC"), mode="sample")[:5]))
                naic = dict(zip(ks, model(fc, att_feats, att_masks, opt=dict(opt, train_mode="NAIC"), mode="sample")[:5]))
        model.train(was_training)
        dev = att_feats.device
        S = model.cfg.seq_length
        # the scorer and the collate of the sampled layouts run on the host.  The non-autoregressive samples' part of it goes first, on the
        # pinned copies sample_pair left behind an event on its side stream: it runs while the (much longer) semi-autoregressive decode does
        early = getattr(model, "_naic_ready", None) if getattr(model.opt, "bofi_rl_sample_pair", True) and os.environ.get("BOFI_RL_EARLY_NAIC", "1") != "0" else None
        prep = {}
        if early is not None:
            model._naic_ready = None
            early[0].synchronize()
            naic_host = early[1]
            seq_n = naic_host["seq"]
            s_naic = score_fn(seq_n)
            prep.update(xe.rl_prepare(model.cfg, None, naic_host, sample_n=sample_n, strict_q1=model.strict_reference, device=dev))
        else:
            seq_n = naic["seq"].cpu()
            s_naic = score_fn(seq_n)
            prep.update(xe.rl_prepare(model.cfg, None, naic, sample_n=sample_n, strict_q1=model.strict_reference, device=dev))
        if "_capped" in saic or getattr(model.opt, "bofi_rl_sample_pair", True):
            saic = model.saic_finish(saic) if "bound_iters" in saic else saic
        if saic["seq"].is_cuda:                                # the semi-autoregressive samples' collate: one launch on the device, issued before the host waits for the ids
            prep.update(xe.rl_prepare_saic_device(model.cfg, saic["seq"].long(), saic["phrase_length"], saic["phrase_syn"]))
        seq_s = saic["seq"].cpu()
        s_saic = score_fn(seq_s)
        if not saic["seq"].is_cuda:
            prep.update(xe.rl_prepare(model.cfg, saic, None, sample_n=sample_n, strict_q1=model.strict_reference, device=dev))
        self._last_rl = {"saic_tokens": (seq_s > 0).float().sum(1).mean(), "naic_tokens": (seq_n > 0).float().sum(1).mean(),
                         # share of the semi-autoregressive loop's S enqueued iterations in which some caption was still open
                         "active_share": min(S, int(saic["phrase_num"].max()) + 1) / S}
        # the gradient pass reads tensors only: the samples' index tensors (host collate of the sampled layouts) and the scores
        b = {"att_feats": att_feats, "seq_saic": saic["seq"].to(dev).long(), "seq_naic": naic["seq"].to(dev).long(),
             "sc_saic": torch.as_tensor(s_saic, dtype=torch.float32).to(dev), "sc_naic": torch.as_tensor(s_naic, dtype=torch.float32).to(dev)}
        b.update(prep)
        self._fwd_calls += 1
        step_word = getattr(self, "_step_word", None)
        if step_word is not None:
            step_word.fill_(self._fwd_calls)
        if self.ops is not None and model.train_dtype == torch.bfloat16:
            self.ops.refresh_if_stale()
        if self.graph and att_masks is None and not self._capturing():
            loss, m1, m2 = self._rl_replay(b, sample_n)
        else:
            loss, m1, m2 = self._rl_forward_backward(b, att_masks, sample_n)
        self.reduce_and_step()
        return loss, m1, m2

    def _rl_reference_step(self, att_feats, att_masks, score_fn, sample_n, temperature):
        """The self-critical step with the REFERENCE's estimator (``opt.bofi_rl_reference_estimator``; loss_wrapper.py:193-209 calls
        ``model(..., mode='sample')`` in train mode with the tape running and differentiates THAT pass): every sampled token is drawn from the
        distribution the gradient pass differentiates -- the training forward's rows under this step's dropout masks -- instead of the inference
        engine's dropout-free one.

        The masks are counter-based (seed, site, element), so the training forward run twice on the same inputs gives the same rows bit for bit,
        and a row of phrase i depends only on the layout and the words up to phrase i - 1 (key-prefix masks, row-wise sublayers).  So:
          * non-autoregressive branch: the engine's bounding loop gives the layout (greedy heads: one per image, no gradient reaches it in the
            reference either); one tape-free training forward gives the fill rows; the words are drawn from them;
          * semi-autoregressive branch: the engine's loop runs ONE iteration per call (bofi_engine_set_saic_range): its bounding step lays the next
            phrase out on the words so far; a tape-free training forward (same seed) gives that phrase's rows; its words are drawn from them and
            handed back to the engine (bofi_engine_saic_put_words), whose next bounding step reads them -- core_SAIC's own process
            (TransformerModel.py:1903-1984) with the fill distribution of the training pass;
          * then the gradient pass: the same forward once more WITH the tape -- its rows at the drawn tokens are the rows they were drawn from
            (``last_rl["reference_gap"]``, 0) --, new_self_critical on both branches, backward, all-reduce, Adam.
        What stays different from the reference: the bound heads decide on the engine (no dropout in the bounding steps); they are argmax decisions
        without gradient.  Cost: one training forward per phrase (no graph replay: the host scorer and the draws sit between the passes)."""
        from . import xe
        model, cfg = self.model, self.model.cfg
        S, dev = cfg.seq_length, att_feats.device
        eng = model.engine()
        feats_in, lens_in = model._as_input(att_feats), model._att_len(att_masks)
        B = feats_in.size(0)
        N = B * sample_n
        if N > model.max_batch:
            raise hip.BofiHipError(f"{N} sampled rows exceed bofi_max_batch={model.max_batch}")
        with torch.no_grad():
            r = eng.decode_naic(feats_in, lens_in, strict_q1=model.strict_reference)
            na = {k: r[k].repeat_interleave(sample_n, dim=0).contiguous() for k in ("phrase_length", "phrase_syn")}
            na["seq"] = torch.zeros(N, S, dtype=torch.int64, device=dev)
            feats_rep = feats_in.repeat_interleave(sample_n, dim=0).contiguous()
            lens_rep = None if lens_in is None else lens_in.repeat_interleave(sample_n).contiguous()
        self._fwd_calls += 1
        step_word = getattr(self, "_step_word", None)
        if step_word is not None:
            step_word.fill_(self._fwd_calls)
        base = int(getattr(model.opt, "seed", 0)) << 32
        seed = base if step_word is not None else base + self._fwd_calls
        armed = self.ops is not None and model.train_dtype == torch.bfloat16
        self.bucket.zero_grad()
        if armed:
            self.ops.refresh_if_stale()
            self.ops.launch_transposes()
            xe._WEIGHTS["provider"] = self.ops
        if self.grouped_dw:
            xe._DEFER["list"] = []
        P = xe.Params(model)
        pos = torch.arange(S, device=dev)[None]

        shared: dict = {}                                       # the encoder's memory and the cross K|V of the tape-free per-phrase forwards (same in each of them)

        def rows_of(prep, reuse=None, feats=None):
            return xe.sampled_logprobs_prepared(P, cfg, att_feats if feats is None else feats, att_masks, prep, sample_n=sample_n, training=model.training, seed=seed,
                                                compute_dtype=model.train_dtype, step_word=step_word, reuse=reuse)

        # The tape-free per-phrase forwards as TWO captured graphs per input signature (graph-mode trainers: the dropout step lives in a device word, so a replay
        # draws this step's masks): A = encoder + decoder (the step's first forward: fills the memory / cross K|V the others share), B = decoder only.  ~250
        # launches per forward become one replay: 1.6 -> ~0.7 ms each, twelve to fourteen times per step.
        fwd_graphs = None
        if self.graph and att_masks is None and step_word is not None and not self._capturing() and getattr(model.opt, "bofi_rl_graph_forwards", True):
            fwd_graphs = self._rl_forward_graphs(rows_of, att_feats, N, S, dev, sample_n)

        def rows_graphed(seq, out, prep_na, first):
            # (the graphs read the words so far and the engine's layout from static buffers and collate them on the device: no host round trip per phrase)
            g_a, g_b, st_in, st_feats, out_a, out_b = fwd_graphs
            if first:
                st_feats.copy_(att_feats)
                st_in["na_syn"].copy_(prep_na["na_syn"]); st_in["na_klen"].copy_(prep_na["na_klen"])
            torch._foreach_copy_([st_in["seq"], st_in["pl"], st_in["psyn"]], [seq, out["phrase_length"], out["phrase_syn"]], non_blocking=True)
            (g_a if first else g_b).replay()
            return out_a if first else out_b

        draws = [0]

        def draw(lp, plen, p0, p1, seq, drawn, mask):
            # one draw per slot from Categorical(logits = row / temperature) by the library's one-pass Gumbel-max sampler (bofi_vocab_sample: counter-hash uniforms, a NaN
            # log-prob counts as -10 as in CaptionModel.py:419-425), kept at the positions of phrases [p0, p1) (bofi_rl_take_draws: ids, their log-probs under the rows they
            # were drawn from, the sampled mask): no index list, no host synchronisation, two launches per phrase
            lpf = lp if lp.dtype == torch.float32 and lp.is_contiguous() else lp.float().contiguous()
            n_, s_, V = lpf.shape
            tok = torch.empty(n_, s_, dtype=torch.int64, device=dev)
            draws[0] += 1
            sseed = (base + 0x5EED0000 + (self._fwd_calls << 8) + draws[0]) & 0xFFFFFFFFFFFFFFFF
            hip.check(hip.lib().bofi_vocab_sample(hip.ptr(lpf), n_ * s_, V, s_, 1, float(temperature), sseed, None, cfg.pad_idx, hip.ptr(tok), hip.stream_ptr()),
                      "bofi_vocab_sample")
            hip.check(hip.lib().bofi_rl_take_draws(hip.ptr(lpf), hip.ptr(tok), hip.ptr(plen), n_, s_, V, int(p0), int(p1), hip.ptr(seq), hip.ptr(drawn), hip.ptr(mask),
                                                   hip.stream_ptr()), "bofi_rl_take_draws")

        try:
            prep_na = xe.rl_prepare_naic_device(cfg, na["phrase_length"], na["phrase_syn"], strict_q1=model.strict_reference)
            seq_s = torch.zeros(N, S, dtype=torch.int64, device=dev)
            seq_n = torch.zeros(N, S, dtype=torch.int64, device=dev)
            drawn_s = torch.zeros(N, S, device=dev)
            drawn_n = torch.zeros(N, S, device=dev)
            mask_s = torch.zeros(N, S, dtype=torch.bool, device=dev)
            mask_n = pos < na["phrase_length"].long().sum(1)[:, None]
            na_plen = na["phrase_length"].to(torch.int32).contiguous()
            out = None
            self._sample_calls_ref = getattr(self, "_sample_calls_ref", 0) + 1
            passes = 0
            for it in range(1, S + 1):
                with torch.no_grad():
                    # (the engine lays the next phrase out on the words so far and stops: this step draws the words itself, from the training forward's rows)
                    out = eng.decode_saic(feats_rep, lens_rep, it_range=(it, it), out=out, want_logprob=False, layout_only=True)
                    if int(out["bound_iters"]) < it:                     # no caption was open in this iteration: the loop is through
                        break
                    if fwd_graphs is not None:
                        lp_s, lp_n = rows_graphed(seq_s, out, prep_na, passes == 0)
                    else:
                        prep = xe.rl_prepare_saic_device(cfg, seq_s, out["phrase_length"], out["phrase_syn"])
                        prep.update(prep_na)
                        lp_s, lp_n = rows_of(prep, shared)
                    passes += 1
                    draw(lp_s, out["phrase_length"], it - 1, it, seq_s, drawn_s, mask_s)       # phrase `it`'s positions
                    eng.saic_put_words(seq_s)
                    if it == 1:
                        draw(lp_n, na_plen, 0, S, seq_n, drawn_n, None)                    # every laid-out position of the non-autoregressive branch
            saic = {"seq": seq_s, "phrase_length": out["phrase_length"], "phrase_syn": out["phrase_syn"]}
            s_saic, s_naic = score_fn(seq_s.cpu()), score_fn(seq_n.cpu())
            prep = xe.rl_prepare_saic_device(cfg, seq_s, out["phrase_length"], out["phrase_syn"])
            prep.update(prep_na)
            sc_s = torch.as_tensor(s_saic, dtype=torch.float32).to(dev)
            sc_n = torch.as_tensor(s_naic, dtype=torch.float32).to(dev)
            replayed = self.graph and att_masks is None and not self._capturing()
            if not replayed:
                lp_saic, lp_naic = rows_of(prep)                          # the gradient pass: the same rows, with the tape
                with torch.no_grad():
                    g_s = (lp_saic.detach().float().gather(2, seq_s[..., None]).squeeze(2) - drawn_s)[mask_s]
                    g_n = (lp_naic.detach().float().gather(2, seq_n[..., None]).squeeze(2) - drawn_n)[mask_n]
                    gap = max(float(g_s.abs().max()) if g_s.numel() else 0.0, float(g_n.abs().max()) if g_n.numel() else 0.0)
                l1, r1 = xe.new_self_critical(lp_saic, seq_s, sc_s, sample_n)
                l2, r2 = xe.new_self_critical(lp_naic, seq_n, sc_n, sample_n)
                loss = l1 + l2
                if self.rl_kl:
                    loss = loss + xe.rl_kl_term(lp_naic, lp_saic, seq_s)
                loss.backward()
                xe.flush_weight_grads()
                loss, m1, m2 = loss.detach(), r1.mean(), r2.mean()
        finally:
            xe._DEFER["list"] = None
            if armed:
                xe._WEIGHTS["provider"] = None
                self.ops.end_step()
        if replayed:
            # the gradient pass as the captured graph of the fast form (the same forward under the same seed and dropout masks + new_self_critical + backward):
            # its log-probs at the drawn tokens come back in the graph's "picked" buffers
            b = {"att_feats": att_feats, "seq_saic": seq_s, "seq_naic": seq_n, "sc_saic": sc_s, "sc_naic": sc_n,
                 "picked_saic": torch.zeros(N, S, device=dev), "picked_naic": torch.zeros(N, S, device=dev)}
            b.update(prep)
            loss, m1, m2 = self._rl_replay(b, sample_n)
            st = self._rl_static
            with torch.no_grad():
                g_s, g_n = (st["picked_saic"] - drawn_s)[mask_s], (st["picked_naic"] - drawn_n)[mask_n]
                gap = max(float(g_s.abs().max()) if g_s.numel() else 0.0, float(g_n.abs().max()) if g_n.numel() else 0.0)
        self._last_rl = {"saic_tokens": (seq_s > 0).float().sum(1).mean(), "naic_tokens": (seq_n > 0).float().sum(1).mean(),
                         "active_share": min(S, int(out["phrase_num"].max()) + 1) / S, "reference_gap": gap, "training_forwards": passes + 1,
                         "gradient_pass_replayed": bool(replayed),
                         "seq_saic": seq_s, "seq_naic": seq_n, "phrase_length_saic": out["phrase_length"], "phrase_syn_saic": out["phrase_syn"]}
        self.reduce_and_step()
        return loss, m1, m2

    def _rl_forward_graphs(self, rows_of, att_feats, N, S, dev, sample_n):
        """(graph A, graph B, static inputs -- words so far, the engine's layout, the NA branch's index tensors --, static features, outputs of A, outputs of B) of the reference-estimator step's tape-free forwards for this
        input signature; captured on first use (after one eager pass of both forms), kept with the trainer's other graphs."""
        model = self.model
        key = ("rlref", tuple(att_feats.shape), att_feats.dtype, N, S, sample_n, model.training, model.train_dtype)
        entry = self._graphs.get(key)
        if entry is not None:
            return entry
        if len(self._graphs) >= self.max_graphs:
            return None
        from . import xe
        cfg = model.cfg
        i64 = lambda: torch.zeros(N, S, dtype=torch.int64, device=dev)
        i32 = lambda: torch.zeros(N, S, dtype=torch.int32, device=dev)
        st_in = {"seq": i64(), "pl": i32(), "psyn": i64(), "na_syn": i64(), "na_klen": i32()}
        st_in["pl"][:, 0] = 1; st_in["na_klen"].fill_(1)          # (the warm-up passes see a one-word layout)
        st_feats = att_feats.clone()

        def rows_dev(shared):
            prep = xe.rl_prepare_saic_device(cfg, st_in["seq"], st_in["pl"], st_in["psyn"])
            prep["na_syn"], prep["na_klen"] = st_in["na_syn"], st_in["na_klen"]
            return rows_of(prep, shared, st_feats)

        with torch.no_grad():
            warm: dict = {}
            rows_dev(warm); rows_dev(warm)                          # both forms once outside any capture
            torch.cuda.synchronize()
            shared: dict = {}
            g_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_a):
                out_a = rows_dev(shared)
            g_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_b, pool=g_a.pool()):
                out_b = rows_dev(shared)
        entry = self._graphs[key] = (g_a, g_b, st_in, st_feats, out_a, out_b)
        self._rl_ref_shared = shared                            # (keeps graph A's memory / cross K|V alive: graph B reads them)
        return entry

    def _rl_forward_backward(self, b, att_masks, sample_n):
        """zero-grad, differentiable re-forward of the sampled captions, new_self_critical for both modes, backward (tensors in,
        device scalars out: what a captured self-critical step replays)."""
        from . import xe
        model = self.model
        self.bucket.zero_grad()
        step_word = getattr(self, "_step_word", None)
        base = int(getattr(model.opt, "seed", 0)) << 32
        armed = self.ops is not None and model.train_dtype == torch.bfloat16
        if armed:
            if not self._capturing():
                self.ops.refresh_if_stale()
            self.ops.launch_transposes()
            xe._WEIGHTS["provider"] = self.ops
        if self.grouped_dw:
            xe._DEFER["list"] = []
        try:
            lp_saic, lp_naic = xe.sampled_logprobs_prepared(xe.Params(model), model.cfg, b["att_feats"], att_masks, b, sample_n=sample_n,
                                                            training=model.training,
                                                            seed=base if step_word is not None else base + self._fwd_calls,
                                                            compute_dtype=model.train_dtype, step_word=step_word)
            if "picked_saic" in b:                             # (the reference-estimator step checks the drawn rows against these: the gradient pass's own log-probs at the drawn tokens)
                with torch.no_grad():
                    b["picked_saic"].copy_(lp_saic.detach().float().gather(2, b["seq_saic"][..., None]).squeeze(2))
                    b["picked_naic"].copy_(lp_naic.detach().float().gather(2, b["seq_naic"][..., None]).squeeze(2))
            l1, r1 = xe.new_self_critical(lp_saic, b["seq_saic"], b["sc_saic"], sample_n)
            l2, r2 = xe.new_self_critical(lp_naic, b["seq_naic"], b["sc_naic"], sample_n)
            loss = l1 + l2
            if self.rl_kl:
                loss = loss + xe.rl_kl_term(lp_naic, lp_saic, b["seq_saic"])
            loss.backward()
            xe.flush_weight_grads()
        finally:
            xe._DEFER["list"] = None
            if armed:
                xe._WEIGHTS["provider"] = None
                self.ops.end_step()
        return loss.detach(), r1.mean(), r2.mean()

    def _rl_replay(self, b, sample_n):
        """The gradient pass of the self-critical step as one hipGraph per input signature (shapes are fixed by images x samples)."""
        keys = sorted(b)
        key = ("rl", tuple((k, tuple(b[k].shape), b[k].dtype) for k in keys), sample_n, self.model.training, self.model.train_dtype, self.rl_kl)
        entry = self._graphs.get(key)
        if entry is None and len(self._graphs) >= self.max_graphs:
            self._rl_static = b                                    # (graph cache full: the eager pass fills the caller's own "picked_*" buffers)
            return self._rl_forward_backward(b, None, sample_n)
        if entry is None:
            static = {k: b[k].clone() for k in keys}
            self._rl_forward_backward(static, None, sample_n)      # warm-up outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._rl_forward_backward(static, None, sample_n)
            entry = self._graphs[key] = (g, static, out)
        g, static, out = entry
        pairs = [(static[k], b[k]) for k in keys if static[k].data_ptr() != b[k].data_ptr()]
        if pairs:
            torch._foreach_copy_([d for d, _ in pairs], [s_ for _, s_ in pairs], non_blocking=True)
        g.replay()
        self._rl_static = static                               # (the graph's input / output buffers: "picked_*" are read from here)
        return out

    # ------------------------------------------------------------------ checkpoint (optimizer.pth of misc.py:87-102)
    def state_dict(self):
        """The optimiser state in the REFERENCE's layout: what ``NoamOpt.state_dict()`` returns (captioning/utils/misc.py:195-198),
        i.e. ``torch.optim.Adam(model.parameters()).state_dict()`` plus ``_step``:
        ``{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..P-1]}], '_step': n}`` with i the
        index of the parameter in ``model.parameters()`` order and tensors in the parameter's own shape.  Parameters that never
        received a gradient (the unused length_attn / ff copies) have no entry, as in torch (their ``.grad`` stays None).
        ``_bofi`` (ignored by torch's loader) carries the dropout counters so that a resumed run does not replay the masks of
        step 1."""
        b = self.bucket
        off = {id(p): o for p, o in zip(b.params, b.offsets)}
        params = list(self.model.parameters())
        m, v = self.m.detach().cpu(), self.v.detach().cpu()
        state = {}
        for i, (name, p) in enumerate(self.model.named_parameters()):
            if name.startswith(b.DEAD_PREFIXES) or self._step == 0:
                continue
            o, n = off[id(p)], p.numel()
            state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": m[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": v[o:o + n].view(p.shape).clone()}
        template = torch.optim.Adam([torch.nn.Parameter(torch.empty(0)) for _ in params], lr=0.0, betas=(self.beta1, self.beta2),
                                    eps=self.eps).state_dict()["param_groups"][0]
        group = dict(template, lr=self.rate() if self._step else 0.0, params=list(range(len(params))))
        return {"state": state, "param_groups": [group], "_step": self._step,
                "_bofi": {"fwd_calls": self._fwd_calls, "model_step": getattr(self.model, "_step", 0),
                          "sample_calls": getattr(self.model, "_sample_calls", 0)}}

    def load_state_dict(self, sd):
        """Accepts the reference's ``optimizer.pth`` (and this class's own): per-parameter Adam moments scattered into the flat
        m / v streams through the bucket's offsets.  ``step`` may be a tensor (torch >= 1.12) or an int (the reference's torch 1.7)."""
        if "state" not in sd or "param_groups" not in sd:
            raise hip.BofiHipError("optimizer.pth: expected torch.optim.Adam's state_dict layout ({'state', 'param_groups'[, '_step']})")
        b = self.bucket
        off = {id(p): o for p, o in zip(b.params, b.offsets)}
        params = list(self.model.parameters())
        order = sd["param_groups"][0]["params"]
        if len(order) != len(params):
            raise hip.BofiHipError(f"optimizer.pth holds {len(order)} parameters, the model has {len(params)}")
        self.m.zero_(); self.v.zero_()
        steps = set()
        for pos, key in enumerate(order):
            st = sd["state"].get(key)
            if st is None:
                continue
            p = params[pos]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise hip.BofiHipError(f"optimizer.pth: parameter {pos} has moments of shape {tuple(st['exp_avg'].shape)}, the model wants {tuple(p.shape)}")
            o, n = off[id(p)], p.numel()
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise hip.BofiHipError(f"optimizer.pth: parameters at different Adam steps {sorted(steps)} (the fused kernel keeps one step count)")
        adam_step = steps.pop() if steps else 0
        self._step = int(sd.get("_step", adam_step))             # NoamOpt's own counter (misc.py:195-204); plain Adam: its step
        extra = sd.get("_bofi", {})
        self._fwd_calls = int(extra.get("fwd_calls", self._step))
        self.model._step = int(extra.get("model_step", self._step))
        self.model._sample_calls = int(extra.get("sample_calls", 0))
