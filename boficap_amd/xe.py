"""XE training forward / backward of the UIC model on the HIP ops.

Replaces, for train_mode 'UIC', ``TransformerModel._forward`` (TransformerModel.py:1713-1724,1759-1775) ->
``EncoderDecoder_UIC.forward`` (:413-468) and the autograd graph torch builds under it (tools/train.py:212-227).
Every arithmetic node is a kernel of libboficap_hip.so wrapped in a ``torch.autograd.Function``; torch only owns
the buffers, the tape and the parameter tensors.  Differences in STRUCTURE from the reference (same results):

  * the image memory is encoded once per image; the seq_per_img captions of an image attend it through the
    attention kernels' ``kdiv`` (the reference repeats the features and encodes every copy, :1703-1707);
  * the teacher-forced bound passes (:476-513 SA, :532-565 NA: one predictor pass per phrase index, the mask
    grown per caption) run as ONE batched pass over N x Pmax virtual [LEN] queries with per-query key counts,
    because only row 0 of each pass is ever read (:375);
  * every mask on this path is a key prefix per query row, so masks travel as int32 key counts;
  * with the collate's row lists (HINTS) the decoder runs over the captions' real positions only and the SA and NA branch
    share one bound pass and one decoder pass (_fill_unpadded, _forward_paired).

float32 is the parity mode.  In bf16 mode (GEMM operands bf16, everything else float32) the module keeps a few per-step
side tables, all cleared by the next forward:

  _STEP_CACHE    ("op"/"tr", ptr, M, N, dtype) -> GEMM operand made from a float32 tensor (cast / transposed once per step);
                 also the bf16 "shadow" a producer wrote next to (or instead of) its float32 output;
                 ("prod", ptr) / ("act", ptr)  -> what a linear's output went through (epilogue dropout / relu), for the
                 node that will compute the gradient w.r.t. it;  ("gop"/"gopr", ptr) -> that gradient, already masked and
                 in bf16, made by its producer (LayerNorm backward, bofi_linear_masked);
  _SHADOW_ONLY   data pointers of float32 PLACEHOLDERS: tensors autograd passes around whose storage is never filled because
                 every consumer reads the bf16 copy (LayerNorm / attention outputs, handed-over gradients).  Any float32
                 reader of a placeholder raises (_real, _operand); placeholders are only created where the tensor has one
                 consumer, and the entries keep them alive so that a pointer names one tensor for the whole step;
  _WEIGHTS       the trainer's per-step bf16 weight operands (trainer.WeightOperands);
  _DEFER         weight-gradient GEMMs put off to one grouped launch after backward (flush_weight_grads).
"""
from __future__ import annotations

import ctypes as C
import contextlib
from typing import Optional

import torch
from torch.autograd import Function

from . import hip

F32 = hip.DT_F32


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = hip.lib()
    return _LIB


def _chk(rc, what):
    if rc:
        hip.check(rc, what)


def _empty(ref: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=ref.device)


def _zeros(ref: torch.Tensor, *shape) -> torch.Tensor:
    return torch.zeros(*shape, dtype=torch.float32, device=ref.device)


def _need(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise hip.BofiHipError(f"{what}: float32 tensor on the HIP device expected, got {t.dtype} on {t.device}")
    return t.contiguous()


# GEMM compute dtype of the training path: float32 (parity) or bfloat16 operands with float32 accumulation, outputs,
# master weights and gradients (BASELINE config 3).  Set per forward pass by forward_uic.
_COMPUTE = {"dtype": torch.float32}

# GEMM operands made during the current step: (kind, data_ptr, shape, dtype) -> (source kept alive, operand).  A weight
# that serves two passes (the decoder runs for the SA and the NA branch) and the image memory that feeds every cross
# K/V projection are cast / transposed once per step instead of once per use.  Cleared by forward_uic.
_STEP_CACHE: dict = {}

# bf16 mode keeps activations that only ever feed a GEMM / attention kernel as bf16 "shadows" made by their producer
# (GEMM second output, LayerNorm / attention writing bf16 directly).  The float32 tensor autograd sees is then either
# written as well (GEMM outputs) or a PLACEHOLDER whose storage is never filled (LayerNorm / attention outputs); the
# data pointers of placeholders are listed here and every kernel wrapper that reads float32 input refuses them.
_SHADOW_ONLY: set = set()

# bf16 operands of the WEIGHTS kept per step by a trainer (boficap_amd.trainer.WeightOperands); set for the duration of one
# forward + backward.  None: every weight is cast / transposed on first use in the step (the _STEP_CACHE path).
_WEIGHTS = {"provider": None}

# Weight-gradient GEMMs put off to grouped launches (flush_weight_grads): a list of
# (dz operand, ld, x operand, ld, target, ldc, M, N, K, bias target) while a trainer collects them, else None.
# "side" / "every": with a side stream set, LinearFn.backward hands the list over every `every` problems and the grouped launch runs on
# that stream BESIDE the rest of the backward (the dX / LayerNorm / attention chain is a sequence of latency-bound launches that leave most
# of the chip idle; the weight gradients are throughput work nothing in the backward waits for).  "held": the handed-over operands, kept
# alive until the streams are joined (join_weight_grads) -- the allocator may hand a block freed on the main stream to the next main-stream
# kernel while the side stream still reads it.
_DEFER = {"list": None, "side": None, "every": 0, "held": []}


def flush_weight_grads():
    """Run the deferred weight-gradient GEMMs (bofi_gemm_tn_grouped) on the current stream and forget them."""
    todo = _DEFER["list"]
    if not todo:
        return 0
    n = len(todo)
    vp, ci = C.c_void_p * n, C.c_int * n
    col = lambda k: [t[k] for t in todo]
    rc = _lib().bofi_gemm_tn_grouped(
        n, vp(*[hip.ptr(t) for t in col(0)]), ci(*col(1)), ci(*col(1)), vp(*[hip.ptr(t) for t in col(2)]), ci(*col(3)), ci(*col(3)),
        vp(*[hip.ptr(t) for t in col(4)]), ci(*col(5)), ci(*col(6)), ci(*col(7)), ci(*col(8)),
        vp(*[hip.ptr(t) for t in col(9)]), hip.stream_ptr())
    _DEFER["list"] = []
    _chk(rc, "bofi_gemm_tn_grouped")
    return n


def flush_weight_grads_aside(final: bool = False):
    """The collected weight-gradient GEMMs as one grouped launch on the side stream, ordered behind everything the current stream has
    enqueued so far (their operands are complete there); a no-op below `every` problems unless ``final``.  Without a side stream the
    final call is flush_weight_grads()."""
    side, todo = _DEFER["side"], _DEFER["list"]
    if side is None:
        return flush_weight_grads() if final else 0
    if not todo or (not final and len(todo) < _DEFER["every"]):
        return 0
    side.wait_stream(torch.cuda.current_stream())
    _DEFER["held"].append(todo)
    with torch.cuda.stream(side):
        return flush_weight_grads()


def join_weight_grads():
    """The current stream waits for the side stream's weight-gradient launches; the operands they read may go after this point."""
    side = _DEFER["side"]
    if side is not None and _DEFER["held"]:
        torch.cuda.current_stream().wait_stream(side)
    _DEFER["held"] = []


def _register_shadow(t, shadow, only=False):
    M, N = t.shape
    _STEP_CACHE[("op", t.data_ptr(), M, N, torch.bfloat16)] = (t, shadow)
    if only:
        _SHADOW_ONLY.add(t.data_ptr())


def _shadow(t):
    hit = _STEP_CACHE.get(("op", t.data_ptr(), t.shape[0], t.shape[1], torch.bfloat16)) if t.dim() == 2 else None
    return hit[1] if hit is not None and hit[0] is t else None


def _real(t, what):
    if t.data_ptr() in _SHADOW_ONLY and _shadow(t) is not None:
        raise hip.BofiHipError(f"{what}: this float32 tensor is a placeholder for a bf16 activation; only GEMM operands may consume it")
    return t


def _gemm(x, ldx, w, bias, residual, y, M, N, K, relu=0, row_len=None, rpg=0, drop=None, y2=None):
    code = hip.dtype_code(x)
    if drop is not None or y2 is not None:
        if drop is not None and row_len is not None:
            raise hip.BofiHipError("dropout in the epilogue is not combined with row_len")
        _chk(_lib().bofi_linear_ex(hip.ptr(x), code, ldx, hip.ptr(w), code, hip.ptr(bias), hip.ptr(residual), N if residual is not None else 0,
                                   hip.ptr(y), F32, N, M, N, K, relu, hip.ptr(row_len), rpg, drop[0] if drop else 0.0, drop[1] if drop else 0,
                                   hip.ptr(drop[2]) if drop else None, hip.ptr(y2), N, hip.stream_ptr()), "bofi_linear_ex")
        return
    _chk(_lib().bofi_linear(hip.ptr(x), code, ldx, hip.ptr(w), code, hip.ptr(bias), hip.ptr(residual), N if residual is not None else 0,
                            hip.ptr(y), F32, N, M, N, K, relu, hip.ptr(row_len), rpg, hip.stream_ptr()), "bofi_linear")


def _granule(dt) -> int:
    return 32 if dt == torch.float32 else 64


def _operand(x, M, N, dt, cache=True, colsum=None, relu_y=None, drop=None):
    """[M, N] float32 -> GEMM operand [M, pad(N)] in the compute dtype (zero padded to the kernel's K granule);
    with ``colsum`` (bf16 only) the column sums of x are added into it on the way."""
    g = _granule(dt)
    Np = (N + g - 1) // g * g
    if dt == torch.float32 and Np == N:
        return x, Np
    key = ("op", x.data_ptr(), M, N, dt)
    hit = _STEP_CACHE.get(key) if (colsum is None and (cache or (relu_y is None and drop is None))) else None
    if hit is not None and (cache or hit[0] is x):             # (a gradient's bf16 copy made by its producer: same tensor object only)
        return hit[1], Np
    if relu_y is not None and colsum is None and not cache and dt == torch.bfloat16 and Np == N:
        g = _STEP_CACHE.get(("gopr", x.data_ptr()))             # dz made by the consumer's input-gradient GEMM (bofi_linear_masked)
        if g is not None and g[0] is x and g[2].data_ptr() == relu_y.data_ptr():     # (a saved OUTPUT comes back as a new tensor object)
            return g[1], Np
        if g is not None and g[0] is x:
            raise hip.BofiHipError("masked gradient hand-over: the placeholder reached a different node than the one it was made for")
    if drop is not None and relu_y is None and colsum is None and not cache and dt == torch.bfloat16 and Np == N:
        g = _STEP_CACHE.get(("gop", x.data_ptr()))              # dz made by the producer of this gradient, mask already applied
        if g is not None and g[0] is x and g[2][0] == drop[0] and g[2][1] == drop[1] and g[2][2] is drop[2]:
            return g[1], Np
    if _WEIGHTS["provider"] is not None and dt == torch.bfloat16 and cache and colsum is None and relu_y is None and drop is None:
        y = _WEIGHTS["provider"].operand(x, M, N)
        if y is not None:
            return y, Np
    if x.data_ptr() in _SHADOW_ONLY:
        raise hip.BofiHipError("placeholder activation without its bf16 shadow")
    if dt == torch.float32:
        y = _zeros(x, M, Np)
        y[:, :N] = x
    else:
        y = torch.empty(M, Np, dtype=torch.bfloat16, device=x.device)
        _chk(_lib().bofi_cast_bf16(hip.ptr(x), N, hip.ptr(y), Np, M, N, hip.ptr(colsum), hip.ptr(relu_y),
                                   drop[0] if drop else 0.0, drop[1] if drop else 0, hip.ptr(drop[2]) if drop else None, hip.stream_ptr()),
             "bofi_cast_bf16")
    if cache and colsum is None:
        _STEP_CACHE[key] = (x, y)
    return y, Np


def _transposed(x, M, N, dt, colsum=None, cache=True):
    """[M, N] float32 -> [N, pad(M)] in the compute dtype; optionally adds the column sums of x into ``colsum``."""
    g = _granule(dt)
    Mp = (M + g - 1) // g * g
    key = ("tr", x.data_ptr(), M, N, dt)
    hit = _STEP_CACHE.get(key) if (cache and colsum is None) else None
    if hit is not None:
        return hit[1], Mp
    if _WEIGHTS["provider"] is not None and dt == torch.bfloat16 and cache and colsum is None:
        xt = _WEIGHTS["provider"].transposed(x, M, N)
        if xt is not None:
            return xt, Mp
    xt = torch.empty(N, Mp, dtype=dt, device=x.device)
    _chk(_lib().bofi_transpose_pad(hip.ptr(x), N, hip.ptr(xt), F32 if dt == torch.float32 else hip.DT_BF16, M, N, Mp, hip.ptr(colsum),
                                   hip.stream_ptr()), "bofi_transpose_pad")
    if cache and colsum is None:
        _STEP_CACHE[key] = (x, xt)
    return xt, Mp


def _pad_k(n: int) -> int:
    return (n + 63) // 64 * 64


class LinearFn(Function):
    """y = act(x w^T + b) [+ residual]  (bofi_linear).  Backward: dx = dz w, dw = dz^T x, db = colsum(dz), all on the
    same MFMA GEMM kernel; the weight-gradient GEMM contracts over rows, so both operands are transposed (and
    zero-padded to the kernel's K granule) first, and the bias gradient falls out of the transpose of dz.
    ``gw`` / ``gb``: gradient buffers (views of the trainer's flat bucket) to ACCUMULATE into -- the weight-gradient GEMM
    then adds in its epilogue (residual = output = gw) and autograd sees no gradient for w / b at all."""

    @staticmethod
    def forward(ctx, x, w, b, residual, relu, row_len, rpg, gw, gb, drop, shadow, masked_grad=False):
        x, w = _need(x, "linear x"), _need(w, "linear w")
        M, K = x.shape
        N = w.shape[0]
        dt = _COMPUTE["dtype"]
        if relu and residual is not None:
            raise hip.BofiHipError("relu with a residual is not a node of this model")
        if row_len is not None and not relu:
            raise hip.BofiHipError("row_len is only used with relu (att_embed)")
        if gw is not None and (gw.shape != w.shape or not gw.is_contiguous()):
            raise hip.BofiHipError("weight gradient buffer must match the weight")
        if drop is not None and dt != torch.bfloat16:
            raise hip.BofiHipError("dropout in the GEMM epilogue is the bf16 path; float32 composes DropoutFn")
        y = _empty(x, M, N)
        if M:
            xo, Kp = _operand(x, M, K, dt)
            wo, _ = _operand(w, N, K, dt)
            y2 = None
            if shadow and dt == torch.bfloat16 and N % 64 == 0:      # the consumer is a GEMM / attention kernel: hand it bf16 directly
                y2 = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
            _gemm(xo, Kp, wo, b, residual, y, M, N, Kp, 1 if relu else 0, row_len, rpg, drop, y2)
            if y2 is not None:
                _register_shadow(y, y2)
            if relu and masked_grad and dt == torch.bfloat16:
                # this activation has ONE consumer, a linear: its input-gradient GEMM can mask by (y > 0) and scale by 1 / (1 - p)
                # in its epilogue and hand this node its dz in bf16 (bofi_linear_masked) instead of a float32 dL/dy to mask and cast
                _STEP_CACHE[("act", y.data_ptr())] = (y, 1.0 / (1.0 - drop[0]) if drop is not None else 1.0)
            if drop is not None and residual is not None and not relu:
                # y = residual + dropout(x w^T + b): whoever computes dL/dy next (the LayerNorm that reads y) can hand this node
                # its dz = mask o dL/dy in bf16 right away (LayerNormFn.backward)
                _STEP_CACHE[("prod", y.data_ptr())] = (y, drop)
        ctx.relu, ctx.dt, ctx.drop = bool(relu), dt, drop
        ctx.has_b, ctx.has_r = b is not None, residual is not None
        ctx.gw, ctx.gb = gw, gb
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dt = ctx.dt
        dy = _need(dy, "linear dy")
        L, st = _lib(), hip.stream_ptr()
        dz = dy
        if ctx.relu and dt != torch.bfloat16:          # zeroed rows (row_len) have y == 0 and drop out here too
            dz = torch.empty_like(dy)
            _chk(L.bofi_relu_bwd(hip.ptr(y), hip.ptr(dy), hip.ptr(dz), dy.numel(), st), "bofi_relu_bwd")
        dx = dw = db = None
        tail = (dy if ctx.has_r else None, None, None, None, None, None, None, None, None)
        if M == 0:
            return (torch.zeros_like(x) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(w) if ctx.needs_input_grad[1] and ctx.gw is None else None,
                    _zeros(x, N) if ctx.has_b and ctx.gb is None else None) + tail
        want_w = ctx.gw is not None or ctx.needs_input_grad[1]
        want_b = ctx.has_b and (ctx.gb is not None or ctx.needs_input_grad[2])
        if want_b and ctx.gb is None:
            db = _zeros(x, N)
        bsum = (ctx.gb if ctx.gb is not None else db) if want_b else None
        if dt == torch.bfloat16:
            # one bf16 copy of dz serves both products (its column sums = the bias gradient, taken during the cast);
            # dW = dz^T x runs on the transposing-read GEMM straight from the row-major operands
            # ReLU and dropout masks are applied inside the cast (a dropped ReLU unit has y == 0: the ReLU test covers it)
            # the bias gradient (column sums of dz) is taken by the weight-gradient GEMM from its A tiles
            cast_sum = None if want_w else bsum
            if cast_sum is not None:                               # (frozen weight, trainable bias: rare)
                made = _shadow(dy)                                 # dy may be a placeholder for a bf16 gradient its producer wrote
                if made is None:                                   # (attention backward, bofi_linear_masked): column sums from that copy
                    g_ = _STEP_CACHE.get(("gopr", dy.data_ptr()))
                    made = g_[1] if (g_ is not None and g_[0] is dy) else None
                if made is not None:
                    cast_sum.add_(made.float().sum(0)[:N])
                    cast_sum = None
            dzo, Np = _operand(dy, M, N, dt, cache=False, colsum=cast_sum, relu_y=y if ctx.relu else None, drop=ctx.drop)
            if ctx.needs_input_grad[0]:
                wt, _ = _transposed(w, N, K, dt)       # [K, Np], once per weight per step
                dx = _empty(x, M, K)
                act = _STEP_CACHE.get(("act", x.data_ptr()))
                if act is not None and act[0] is x and K % 64 == 0:
                    # x = relu(..) (+ dropout) of the previous linear, which is its only producer-consumer pair: dx stays an unfilled
                    # placeholder, the masked gradient goes out in bf16
                    dzb = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
                    _chk(L.bofi_linear_masked(hip.ptr(dzo), hip.DT_BF16, Np, hip.ptr(wt), hip.DT_BF16, hip.ptr(x), K, act[1], hip.ptr(dzb),
                                              hip.DT_BF16, K, M, K, Np, st), "bofi_linear_masked")
                    _STEP_CACHE[("gopr", dx.data_ptr())] = (dx, dzb, x)
                    _SHADOW_ONLY.add(dx.data_ptr())                # a placeholder: a cast of it must never happen silently
                else:
                    _gemm(dzo, Np, wt, None, None, dx, M, K, Np)
            if want_w:
                xo, Kp = _operand(x, M, K, dt)         # the forward's operand, still in the step cache
                target = ctx.gw
                if target is None:
                    dw = target = _zeros(x, N, K)
                if _DEFER["list"] is not None and ctx.gw is not None:
                    _DEFER["list"].append((dzo, Np, xo, Kp, target, K, M, N, K, bsum))      # with the step's other weight gradients
                    flush_weight_grads_aside()
                else:
                    _chk(L.bofi_gemm_tn_acc(hip.ptr(dzo), Np, Np, hip.ptr(xo), Kp, Kp, hip.ptr(target), K, M, N, K, hip.ptr(bsum), st), "bofi_gemm_tn_acc")
            return (dx, dw, db) + tail
        if ctx.needs_input_grad[0]:
            dzo, Np = _operand(dz, M, N, dt, cache=False)
            wt, _ = _transposed(w, N, K, dt)           # [K, Np]
            dx = _empty(x, M, K)
            _gemm(dzo, Np, wt, None, None, dx, M, K, Np)
        if want_w:
            dzt, Mp = _transposed(dz, M, N, dt, colsum=bsum, cache=False)     # [N, Mp] (+ bias gradient)
            xt, _ = _transposed(x, M, K, dt)                                   # [K, Mp]
            if ctx.gw is not None:
                _gemm(dzt, Mp, xt, None, ctx.gw, ctx.gw, N, K, Mp)             # gw += dz^T x
            else:
                dw = _empty(x, N, K)
                _gemm(dzt, Mp, xt, None, None, dw, N, K, Mp)
        elif want_b:
            _chk(L.bofi_colsum_add(hip.ptr(dz), hip.ptr(bsum), M, N, st), "bofi_colsum_add")
        return (dx, dw, db) + tail


def linear(x, w, b=None, residual=None, relu=False, row_len=None, rpg=0, gw=None, gb=None, drop=None, shadow=False, masked_grad=False):
    """``drop``: (p, seed, step word or None) -> y = residual + dropout(act(x w^T + b)), mask made in the GEMM epilogue (bf16 path).
    ``shadow``: the output feeds a GEMM or an attention kernel -> also emit it in bf16 from the epilogue (bf16 path)."""
    return LinearFn.apply(x, w, b, residual, relu, row_len, rpg, gw, gb, drop, shadow, masked_grad)


class LayerNormFn(Function):
    """a_2 (x - mean) / (std + eps) + b_2 with the unbiased std (TransformerModel.py:1346-1349).
    ``gg`` / ``gb``: gradient buffers of gain / bias to accumulate into (see LinearFn).
    Returns (x, y): the pre-norm sublayers use x twice -- normalised, and as the residual (SublayerConnection, :1361-1363).
    Taking the residual from the FIRST output routes both gradients of x through this node, and the backward kernel adds them
    on the fly instead of autograd launching an add."""

    @staticmethod
    def forward(ctx, x, gain, bias, gg, gb, gemm_only):
        x, gain, bias = _real(_need(x, "ln x"), "ln x"), _need(gain, "ln gain"), _need(bias, "ln bias")
        rows, d = x.shape
        y = torch.empty_like(x)
        if gemm_only and _COMPUTE["dtype"] == torch.bfloat16 and d % 64 == 0:
            # every consumer is a GEMM: write the normalised rows as bf16 only; y stays an unfilled placeholder
            yb = torch.empty(rows, d, dtype=torch.bfloat16, device=x.device)
            _chk(_lib().bofi_layernorm(hip.ptr(x), hip.ptr(gain), hip.ptr(bias), hip.ptr(yb), hip.DT_BF16, rows, d, hip.stream_ptr()), "bofi_layernorm")
            _register_shadow(y, yb, only=True)
        else:
            _chk(_lib().bofi_layernorm(hip.ptr(x), hip.ptr(gain), hip.ptr(bias), hip.ptr(y), F32, rows, d, hip.stream_ptr()), "bofi_layernorm")
        ctx.gg, ctx.gb = gg, gb
        prod = _STEP_CACHE.get(("prod", x.data_ptr()))
        ctx.prod_drop = prod[1] if (prod is not None and prod[0] is x and _COMPUTE["dtype"] == torch.bfloat16 and d in (512, 128)) else None
        ctx.set_materialize_grads(False)                       # an unused output gets None, not a zero tensor
        ctx.save_for_backward(x, gain)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_pass, dy):
        x, gain = ctx.saved_tensors
        rows, d = x.shape
        if dy is None:                                         # only the pass-through was used
            return g_pass, None, None, None, None, None
        dy = _need(dy, "ln dy")
        add = None if g_pass is None else _need(g_pass, "ln residual gradient")
        direct = ctx.gg is not None and ctx.gb is not None
        dx = torch.empty_like(x)
        dg, db = (ctx.gg, ctx.gb) if direct else (_zeros(x, d), _zeros(x, d))
        pd = ctx.prod_drop
        if pd is not None:
            # x came out of a linear with epilogue dropout: also write that linear's dz (mask applied, bf16)
            dz = torch.empty(rows, d, dtype=torch.bfloat16, device=x.device)
            _chk(_lib().bofi_layernorm_bwd_ex(hip.ptr(x), hip.ptr(gain), hip.ptr(dy), hip.ptr(add), hip.ptr(dx), hip.ptr(dg), hip.ptr(db), rows, d,
                                              hip.ptr(dz), pd[0], pd[1], hip.ptr(pd[2]), hip.stream_ptr()), "bofi_layernorm_bwd_ex")
            _STEP_CACHE[("gop", dx.data_ptr())] = (dx, dz, pd)
        else:
            _chk(_lib().bofi_layernorm_bwd(hip.ptr(x), hip.ptr(gain), hip.ptr(dy), hip.ptr(add), hip.ptr(dx), hip.ptr(dg), hip.ptr(db), rows, d,
                                           hip.stream_ptr()), "bofi_layernorm_bwd")
        return (dx, None, None, None, None, None) if direct else (dx, dg, db, None, None, None)


def layer_norm(x, gain, bias, gg=None, gb=None, gemm_only=False):
    """``gemm_only``: the output is consumed by GEMMs only (bf16 mode then skips the float32 copy)."""
    return LayerNormFn.apply(x, gain, bias, gg, gb, gemm_only)[1]


def layer_norm_res(x, gain, bias, gg=None, gb=None, gemm_only=False):
    """(x for the residual connection, LayerNorm(x)): see LayerNormFn."""
    return LayerNormFn.apply(x, gain, bias, gg, gb, gemm_only)


def _off(t: torch.Tensor, col: int) -> C.c_void_p:
    return C.c_void_p(t.data_ptr() + t.element_size() * col)


def _rows_alloc(shape, dtype, device, ragged: bool, tail):
    """Output buffer of an attention over unpadded rows: the kernels write the captions' rows only, the rows behind the last
    caption must still be finite (row-wise ops and weight-gradient GEMMs read them).  ``tail``: an upper bound, known on the
    host, of how many such rows there are at the end of the list (None: unknown, clear everything)."""
    if not ragged:
        return torch.empty(shape, dtype=dtype, device=device)
    if tail is None or tail >= shape[0]:
        return torch.zeros(shape, dtype=dtype, device=device)
    t = torch.empty(shape, dtype=dtype, device=device)
    t[shape[0] - tail:].zero_()
    return t


class AttentionFn(Function):
    """softmax(q k^T / 8, keys < klen) v per head (TransformerModel.py:1421-1432) on column slices of packed
    projection buffers: q = qbuf[:, qoff:qoff+d], k = kvbuf[:, koff:...], v = kvbuf[:, voff:...].
    qbuf may be the same tensor as kvbuf (packed q|k|v of a self-attention)."""

    @staticmethod
    def forward(ctx, qbuf, kvbuf, qoff, koff, voff, B, H, Lq, Lk, kdiv, klen, klen_sb, klen_sq, klen_bias, drop, seg, grad_bf16=False):
        qbuf, kvbuf = _need(qbuf, "attention q"), _need(kvbuf, "attention kv")
        # seg = (q_start int32 [B], q_count int32 [B], k_ragged): unpadded rows (see bofi_attention_ex); sizes are then the caller's business
        if seg is None and (qbuf.shape[0] != B * Lq or kvbuf.shape[0] * kdiv != B * Lk):
            raise hip.BofiHipError(f"attention operand rows {qbuf.shape[0]}, {kvbuf.shape[0]} do not match B={B} Lq={Lq} Lk={Lk} kdiv={kdiv}")
        if seg is None and klen is not None and (klen.dtype != torch.int32 or klen.numel() < (B - 1) * klen_sb + (Lq - 1) * klen_sq + 1):
            raise hip.BofiHipError("klen must be int32 and cover every (b, query) it is indexed with")
        if seg is not None and ((klen is not None and (klen.dtype != torch.int32 or klen.numel() < qbuf.shape[0])) or seg[0].dtype != torch.int32
                                or seg[1].dtype != torch.int32 or seg[0].numel() != B or seg[1].numel() != B):
            raise hip.BofiHipError("unpadded attention needs int32 q_start / q_count of B entries and, if any, one klen per query row")
        d = H * 64
        # rows outside every segment (padding of the row list) are never written by the kernels: keep them finite
        sp = (hip.ptr(seg[0]), hip.ptr(seg[1]), int(bool(seg[2]))) if seg is not None else (None, None, 0)
        tail = seg[3] if seg is not None and len(seg) > 3 else None
        ldq, ldk = qbuf.shape[1], kvbuf.shape[1]
        bf16 = _COMPUTE["dtype"] == torch.bfloat16
        qs, kvs = (_shadow(qbuf), _shadow(kvbuf)) if bf16 else (None, None)
        ctx.shadows = qs is not None and kvs is not None and Lk <= 64
        if ctx.shadows:
            # bf16 projections from the GEMM epilogue in, bf16 context out (the operand of the output projection):
            # ``out`` stays an unfilled placeholder
            out = _empty(qbuf, qbuf.shape[0], d)              # placeholder: consumers read the bf16 copy
            ob = torch.empty(qbuf.shape[0], d, dtype=torch.bfloat16, device=qbuf.device)      # (unpadded rows: the kernel clears the trailing rows)
            _chk(_lib().bofi_attention_ex(_off(qs, qoff), ldq, _off(kvs, koff), ldk, _off(kvs, voff), ldk, hip.ptr(ob), d, hip.DT_BF16, B, H,
                                          Lq, Lk, kdiv, hip.ptr(klen), klen_sb, klen_sq, klen_bias, drop[0] if drop else 0.0,
                                          drop[1] if drop else 0, hip.ptr(drop[2]) if drop else None, sp[0], sp[1], sp[2],
                                          qbuf.shape[0] if seg is not None else 0, hip.stream_ptr()),
                 "bofi_attention_ex")
            _register_shadow(out, ob, only=True)
            ctx.save_for_backward(qs, kvs)
        else:
            _real(qbuf, "attention q"), _real(kvbuf, "attention kv")
            out = _rows_alloc((qbuf.shape[0], d), torch.float32, qbuf.device, seg is not None, tail)
            drop = None                                        # dropout(p_attn) is built into the bf16 kernels only
            _chk(_lib().bofi_attention_ex(_off(qbuf, qoff), ldq, _off(kvbuf, koff), ldk, _off(kvbuf, voff), ldk, hip.ptr(out), d, F32, B, H,
                                          Lq, Lk, kdiv, hip.ptr(klen), klen_sb, klen_sq, klen_bias, 0.0, 0, None, sp[0], sp[1], sp[2], 0,
                                          hip.stream_ptr()), "bofi_attention_ex")
            ctx.save_for_backward(qbuf, kvbuf)
        ctx.drop = drop
        ctx.meta = (qoff, koff, voff, B, H, Lq, Lk, kdiv, klen_sb, klen_sq, klen_bias)
        # bf16 mode: backward on the matrix cores -- for up to 64 keys (and 64 query rows, or a row list); beyond that (up to 128 keys:
        # max_boxes = 100 regions) the float32 kernel, which walks the query rows in chunks
        ctx.mfma = bf16 and Lk <= 64 and (Lq <= 64 or (seg is not None and not seg[2]))
        ctx.seg = seg
        ctx.grad_bf16 = bool(grad_bf16)
        ctx.same = qbuf.data_ptr() == kvbuf.data_ptr()
        ctx.klen = klen
        return out

    @staticmethod
    def backward(ctx, dout):
        qbuf, kvbuf = ctx.saved_tensors                        # float32 buffers, or their bf16 shadows
        qoff, koff, voff, B, H, Lq, Lk, kdiv, sb, sq, bias = ctx.meta
        dout = _real(_need(dout, "attention dout"), "attention dout")
        ldq, ldk = qbuf.shape[1], kvbuf.shape[1]
        # the MFMA kernel writes every element of the q / k / v slices (no atomics); the VALU kernel accumulates shared keys
        covered = ctx.seg is None and ctx.mfma and ((ctx.same and ldq == 3 * H * 64) or (not ctx.same and ldq == H * 64 and ldk == 2 * H * 64))
        sp = (hip.ptr(ctx.seg[0]), hip.ptr(ctx.seg[1]), int(bool(ctx.seg[2]))) if ctx.seg is not None else (None, None, 0)
        alloc = torch.empty if covered else torch.zeros         # (the float32 kernel adds dk / dv with atomics when keys are shared or the query rows are chunked)
        tail = ctx.seg[3] if ctx.seg is not None and len(ctx.seg) > 3 else None
        fits = (ctx.same and ldq == 3 * H * 64) or (not ctx.same and ldq == H * 64 and ldk == 2 * H * 64)
        # grad_bf16: q (and, for a packed self-attention, k and v) come straight from a projection GEMM whose backward wants
        # its dz in bf16 anyway -- the MFMA kernel writes that, and autograd gets an unfilled float32 placeholder carrying it
        qb16 = ctx.mfma and ctx.grad_bf16 and fits
        if qb16:
            dq = torch.empty(qbuf.shape, dtype=torch.float32, device=qbuf.device)
            dq_out = torch.empty(qbuf.shape, dtype=torch.bfloat16, device=qbuf.device)
            _register_shadow(dq, dq_out, only=True)            # (listed as a placeholder: any float32 reader of it raises)
            dkv = dq if ctx.same else torch.empty(kvbuf.shape, dtype=torch.float32, device=kvbuf.device)
        elif ctx.seg is not None and ctx.mfma and fits:
            # unpadded rows: the kernel writes every caption's rows and clears the rows behind the last caption (q_rows)
            dq = dq_out = torch.empty(qbuf.shape, dtype=torch.float32, device=qbuf.device)
            dkv = dq if ctx.same else torch.empty(kvbuf.shape, dtype=torch.float32, device=kvbuf.device)
        else:
            dq = dq_out = alloc(qbuf.shape, dtype=torch.float32, device=qbuf.device)
            dkv = dq if ctx.same else alloc(kvbuf.shape, dtype=torch.float32, device=kvbuf.device)
        if ctx.mfma:
            F32c, B16c = F32, hip.DT_BF16
            kv_out = dq_out if ctx.same else dkv
            _chk(_lib().bofi_attention_bwd_mfma(_off(qbuf, qoff), ldq, _off(kvbuf, koff), ldk, _off(kvbuf, voff), ldk, hip.dtype_code(qbuf),
                                                hip.ptr(dout), H * 64, _off(dq_out, qoff), ldq, _off(kv_out, koff), _off(kv_out, voff), ldk,
                                                B16c if qb16 else F32c, B16c if (qb16 and ctx.same) else F32c, B, H, Lq, Lk,
                                                kdiv, hip.ptr(ctx.klen), sb, sq, bias, ctx.drop[0] if ctx.drop else 0.0,
                                                ctx.drop[1] if ctx.drop else 0, hip.ptr(ctx.drop[2]) if ctx.drop else None, sp[0], sp[1], sp[2],
                                                qbuf.shape[0] if (ctx.seg is not None and fits) else 0, hip.stream_ptr()), "bofi_attention_bwd_mfma")
            return (dq, None if ctx.same else dkv) + (None,) * 15
        _chk(_lib().bofi_attention_bwd(_off(qbuf, qoff), ldq, _off(kvbuf, koff), ldk, _off(kvbuf, voff), ldk, hip.ptr(dout), H * 64,
                                       _off(dq, qoff), _off(dkv, koff), _off(dkv, voff), B, H, Lq, Lk, kdiv, hip.ptr(ctx.klen), sb, sq, bias,
                                       sp[0], sp[1], sp[2], hip.stream_ptr()), "bofi_attention_bwd")
        return (dq, None if ctx.same else dkv) + (None,) * 15


def attention(qbuf, kvbuf, qoff, koff, voff, B, H, Lq, Lk, kdiv=1, klen=None, klen_sb=0, klen_sq=0, klen_bias=0, drop=None, seg=None,
              grad_bf16=False):
    """``drop``: (p, seed, step word) dropout on the attention probabilities (TransformerModel.py:1430-1431); applied by the bf16
    kernels (bf16 training mode with bf16 projections at hand), ignored by the float32 parity kernels."""
    return AttentionFn.apply(qbuf, kvbuf, qoff, koff, voff, B, H, Lq, Lk, kdiv, klen, klen_sb, klen_sq, klen_bias, drop, seg, grad_bf16)


class EmbedFn(Function):
    """pos_embed(tgt_embed(tok) [+ syn_embed(syn)]) (TransformerModel.py:1484-1511); ids int64 [rows], position = row % L.
    ``g_tok`` / ``g_syn``: gradient buffers of the tables to scatter-add into (see LinearFn)."""

    @staticmethod
    def forward(ctx, lut_tok, lut_syn, pe, tok, syn, L, g_tok, g_syn, pos=None):
        ref = lut_tok if lut_tok is not None else lut_syn
        d = ref.shape[1]
        ids = tok if tok is not None else syn
        rows = ids.numel()
        for t, lut in ((tok, lut_tok), (syn, lut_syn)):
            if t is not None and (t.dtype != torch.int64 or not t.is_contiguous() or lut is None):
                raise hip.BofiHipError("embedding ids must be contiguous int64 with their table")
        x = _empty(ref, rows, d)
        _chk(_lib().bofi_embed_rows(hip.ptr(lut_tok if tok is not None else None), hip.ptr(lut_syn if syn is not None else None), hip.ptr(pe),
                                    hip.ptr(tok), hip.ptr(syn), hip.ptr(pos), rows, L, d, hip.ptr(x), hip.stream_ptr()), "bofi_embed_rows")
        ctx.tok, ctx.syn, ctx.g = tok, syn, (g_tok, g_syn)
        ctx.shapes = (None if lut_tok is None else lut_tok.shape, None if lut_syn is None else lut_syn.shape)
        return x

    @staticmethod
    def backward(ctx, dx):
        dx = _need(dx, "embed dx")
        rows, d = dx.shape
        out = []
        for ids, shape, direct in ((ctx.tok, ctx.shapes[0], ctx.g[0]), (ctx.syn, ctx.shapes[1], ctx.g[1])):
            if ids is None or shape is None:
                out.append(None)
                continue
            g = direct if direct is not None else _zeros(dx, *shape)
            _chk(_lib().bofi_embed_bwd(hip.ptr(dx), hip.ptr(ids), hip.ptr(g), rows, d, float(d) ** 0.5, hip.stream_ptr()), "bofi_embed_bwd")
            out.append(None if direct is not None else g)
        return out[0], out[1], None, None, None, None, None, None, None


def embed(lut_tok, lut_syn, pe, tok, syn, L, g_tok=None, g_syn=None, pos=None):
    """``pos``: explicit position per row (int64) instead of row % L (unpadded row lists)."""
    return EmbedFn.apply(lut_tok, lut_syn, pe, tok, syn, L, g_tok, g_syn, pos)


class LogSoftmaxFn(Function):
    """F.log_softmax over the last dim, in place on the logits (bofi_vocab_finalize); also leaves the greedy ids."""

    @staticmethod
    def forward(ctx, logits):
        logits = _need(logits, "log_softmax input")
        rows, V = logits.shape
        ids = torch.empty(rows, dtype=torch.int64, device=logits.device)
        if rows:
            _chk(_lib().bofi_vocab_finalize(hip.ptr(logits), rows, V, 1, 1, None, 0, hip.ptr(ids), hip.stream_ptr()), "bofi_vocab_finalize")
        ctx.mark_dirty(logits)
        ctx.save_for_backward(logits)
        return logits

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _need(dy, "log_softmax dy")
        dx = torch.empty_like(y)
        _chk(_lib().bofi_logsoftmax_bwd(hip.ptr(y), hip.ptr(dy), hip.ptr(dx), y.shape[0], y.shape[1], hip.stream_ptr()), "bofi_logsoftmax_bwd")
        return dx


def log_softmax(logits):
    return LogSoftmaxFn.apply(logits)


class PickLogSoftmaxFn(Function):
    """(log_softmax(logits) in place, its entry at labels[r] per row): the token NLL's gather fused with the log-softmax.  When only
    the picked values carry a gradient -- the training step -- the backward is ONE pass over the log-probs (bofi_nll_bwd) instead
    of a zero-filled dense dL/dy, a scatter into it and the log-softmax backward reading it back."""

    @staticmethod
    def forward(ctx, logits, labels, grad_bf16=False):
        logits = _need(logits, "log_softmax input")
        ctx.grad_bf16 = bool(grad_bf16)
        rows, V = logits.shape
        labels = labels.to(torch.int64).contiguous()
        ids = torch.empty(rows, dtype=torch.int64, device=logits.device)
        if rows:
            _chk(_lib().bofi_vocab_finalize(hip.ptr(logits), rows, V, 1, 1, None, 0, hip.ptr(ids), hip.stream_ptr()), "bofi_vocab_finalize")
        picked = logits.gather(1, labels.unsqueeze(1)).squeeze(1)
        ctx.mark_dirty(logits)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(logits, labels)
        return logits, picked

    @staticmethod
    def backward(ctx, g_logp, g_picked):
        y, labels = ctx.saved_tensors
        if g_logp is None and g_picked is None:
            return None, None, None
        dx = torch.empty_like(y)
        if g_logp is None:
            gp = _need(g_picked, "picked gradient")
            if ctx.grad_bf16 and _COMPUTE["dtype"] == torch.bfloat16:
                # the logits have one consumer-producer pair, the vocabulary projection: hand it its dz in bf16 (K-granule padded);
                # dx stays an unfilled placeholder
                Vp = (y.shape[1] + 63) // 64 * 64
                dzb = torch.empty(y.shape[0], Vp, dtype=torch.bfloat16, device=y.device)
                _chk(_lib().bofi_nll_bwd(hip.ptr(y), hip.ptr(labels), hip.ptr(gp), hip.ptr(dzb), hip.DT_BF16, Vp, y.shape[0], y.shape[1],
                                         hip.stream_ptr()), "bofi_nll_bwd")
                _register_shadow(dx, dzb, only=True)
                return dx, None, None
            _chk(_lib().bofi_nll_bwd(hip.ptr(y), hip.ptr(labels), hip.ptr(gp), hip.ptr(dx), F32, y.shape[1], y.shape[0], y.shape[1],
                                     hip.stream_ptr()), "bofi_nll_bwd")
            return dx, None, None
        dy = _need(g_logp, "log_softmax dy")
        if g_picked is not None:                               # both outputs used: fold the picked gradient into the dense one
            dy = dy.clone()
            dy.scatter_add_(1, labels.unsqueeze(1), g_picked.unsqueeze(1))
        _chk(_lib().bofi_logsoftmax_bwd(hip.ptr(y), hip.ptr(dy), hip.ptr(dx), y.shape[0], y.shape[1], hip.stream_ptr()), "bofi_logsoftmax_bwd")
        return dx, None, None


class UicCriterionFn(Function):
    """LanguageModelCriterion_UIC (captioning/modules/losses.py:319-369) of the paired training forward in one launch each way
    (bofi_uic_criterion): the four slot outputs [N, Pm, .], the loader's phrase tensors as labels / mask, the picked token
    log-probs with the SA / NA row weights -> (loss [1], the six parts [6], reported only)."""

    @staticmethod
    def forward(ctx, sa_len, sa_syn, na_len, na_syn, picked, phrase_num, phrase_length, phrase_syn, w_sa, w_na):
        ts = [_need(t, "criterion input") for t in (sa_len, sa_syn, na_len, na_syn, picked, w_sa, w_na)]
        N, Pm, c_len = ts[0].shape
        c_syn, L, T = ts[1].shape[2], phrase_length.shape[1], ts[4].numel()
        if ts[1].shape[:2] != (N, Pm) or ts[2].shape != ts[0].shape or ts[3].shape != ts[1].shape or ts[5].numel() != T or ts[6].numel() != T \
                or phrase_num.numel() != N or phrase_length.shape != (N, L) or phrase_syn.shape != (N, L) or Pm + 1 > L:
            raise hip.BofiHipError("criterion: inconsistent shapes")
        ints = [t.to(torch.int64).contiguous() for t in (phrase_num, phrase_length, phrase_syn)]
        out = torch.empty(8, dtype=torch.float32, device=ts[0].device)
        _chk(_lib().bofi_uic_criterion(hip.ptr(ts[0]), hip.ptr(ts[1]), hip.ptr(ts[2]), hip.ptr(ts[3]), N, Pm, c_len, c_syn, hip.ptr(ints[0]),
                                       hip.ptr(ints[1]), hip.ptr(ints[2]), L, hip.ptr(ts[4]), hip.ptr(ts[5]), hip.ptr(ts[6]), T, hip.ptr(out),
                                       hip.stream_ptr()), "bofi_uic_criterion")
        ctx.save_for_backward(*ts, *ints, out)
        ctx.dims = (N, Pm, c_len, c_syn, L, T)
        parts = out[:6]
        ctx.mark_non_differentiable(parts)
        return out[6:7].clone(), parts

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        sa_len, sa_syn, na_len, na_syn, picked, w_sa, w_na, pnum, plen, psyn, out = ctx.saved_tensors
        N, Pm, c_len, c_syn, L, T = ctx.dims
        g = _need(g_loss, "criterion gradient")
        d = [torch.empty_like(t) for t in (sa_len, sa_syn, na_len, na_syn, picked)]
        _chk(_lib().bofi_uic_criterion_bwd(hip.ptr(sa_len), hip.ptr(sa_syn), hip.ptr(na_len), hip.ptr(na_syn), N, Pm, c_len, c_syn, hip.ptr(pnum),
                                           hip.ptr(plen), hip.ptr(psyn), L, hip.ptr(picked), hip.ptr(w_sa), hip.ptr(w_na), T, hip.ptr(g), hip.ptr(out),
                                           hip.ptr(d[0]), hip.ptr(d[1]), hip.ptr(d[2]), hip.ptr(d[3]), hip.ptr(d[4]), hip.stream_ptr()),
             "bofi_uic_criterion_bwd")
        return (*d, None, None, None, None, None)


def log_softmax_pick(logits, labels, grad_bf16=False):
    """(log-probs [T, V], log-probs at labels [T]); see PickLogSoftmaxFn.  ``grad_bf16``: the logits come straight out of a
    linear (their only other use) -- its backward gets the gradient in bf16."""
    return PickLogSoftmaxFn.apply(logits, labels, grad_bf16)


def greedy_ids(logits):
    """argmax over the last dim (first maximal index), no tape; the logits are consumed."""
    rows, V = logits.shape
    ids = torch.empty(rows, dtype=torch.int64, device=logits.device)
    _chk(_lib().bofi_vocab_finalize(hip.ptr(logits), rows, V, 1, 0, None, 0, hip.ptr(ids), hip.stream_ptr()), "bofi_vocab_finalize")
    return ids


class DropoutFn(Function):
    """residual + dropout(x): the mask is a counter hash of (seed, element), regenerated in the backward."""

    @staticmethod
    def forward(ctx, x, residual, p, seed, step_word=None):
        x = _real(_need(x, "dropout x"), "dropout x")
        y = torch.empty_like(x)
        _chk(_lib().bofi_dropout(hip.ptr(x), hip.ptr(residual), hip.ptr(y), x.numel(), p, seed, hip.ptr(step_word), hip.stream_ptr()), "bofi_dropout")
        ctx.p, ctx.seed, ctx.has_r, ctx.step_word = p, seed, residual is not None, step_word
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _need(dy, "dropout dy")
        dx = torch.empty_like(dy)
        _chk(_lib().bofi_dropout(hip.ptr(dy), None, hip.ptr(dx), dy.numel(), ctx.p, ctx.seed, hip.ptr(ctx.step_word), hip.stream_ptr()), "bofi_dropout")
        return dx, (dy if ctx.has_r else None), None, None, None


class _Drop:
    """Dropout sites of one forward pass.  Site k draws from stream ``seed_k + step``: seed_k is a per-site constant mixed
    from the base seed, ``step`` either folded in on the host (eager) or read by the kernels from a DEVICE word
    (``step_word``: what lets a captured graph draw fresh masks on every replay)."""

    def __init__(self, p: float, p_att: float, seed: Optional[int], step_word: Optional[torch.Tensor] = None):
        self.p, self.p_att, self.on = p, p_att, seed is not None
        self.seed, self.k, self.step_word = (seed or 0), 0, step_word

    def _next(self):
        self.k += 1
        return (self.seed * 0x100000001B3 + self.k * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    def site(self, p=None):
        """(p, seed, step word) for a kernel with built-in dropout."""
        return (self.p if p is None else p, self._next(), self.step_word)

    def attn(self):
        """Dropout site of an attention core (None when dropout is off)."""
        return self.site() if self.on and self.p > 0.0 else None

    def __call__(self, x, residual=None, p=None):
        p = self.p if p is None else p
        if not self.on or p <= 0.0:
            return x if residual is None else None          # None: caller fuses the residual into the GEMM instead
        return DropoutFn.apply(x, residual, p, self._next(), self.step_word)


# ------------------------------------------------------------------------------------------------ the model
class Params:
    """Where the forward pass gets its weights -- and, when the model's parameters live in a trainer's flat bucket
    (boficap_amd.trainer.FlatBucket), the gradient buffer each kernel accumulates into directly.

    Without a bucket every entry is the module's own nn.Parameter, packed projections are ``torch.cat``s and the
    gradients travel through autograd as usual (what the parity tests exercise).  With a bucket, q|k|v weights of an
    attention block are adjacent in HBM, so the packed [3d, d] operand is a free view, and its gradient view is handed
    to the weight-gradient GEMM as the accumulation target."""

    def __init__(self, module):
        self.t = dict(module.named_parameters())
        self.t["model.pos_embed.pe"] = module.model.pos_embed.pe
        self.bucket = getattr(module, "_bucket", None)
        self._packed = {}

    def __getitem__(self, name):
        return self.t[name]

    def g(self, name):
        """Gradient accumulation buffer of a parameter (None: let autograd carry the gradient)."""
        return self.t[name].grad if self.bucket is not None else None

    def packed(self, prefix, idx, what):
        """(linears[idx...].what stacked on dim 0, its gradient buffer or None)."""
        key = (prefix, idx, what)
        if key not in self._packed:
            parts = [self.t[f"{prefix}.linears.{i}.{what}"] for i in idx]
            view = self.bucket.span(parts) if self.bucket is not None else None
            self._packed[key] = view if view is not None else (torch.cat(parts, 0), None)
        return self._packed[key]

    # ---- the model's recurring nodes
    def lin(self, x, wname, residual=None, relu=False, row_len=None, rpg=0, drop=None, shadow=False, masked_grad=False):
        w, b = wname + ".weight", wname + ".bias"
        return linear(x, self.t[w], self.t[b], residual, relu, row_len, rpg, self.g(w), self.g(b), drop, shadow, masked_grad)

    def lin_packed(self, x, prefix, idx):
        """Fused q|k|v (or k|v) projection; its output feeds an attention kernel."""
        (w, gw), (b, gb) = self.packed(prefix, idx, "weight"), self.packed(prefix, idx, "bias")
        return linear(x, w, b, gw=gw, gb=gb, shadow=True)

    def ln(self, x, prefix, gemm_only=True):
        a, b = prefix + ".a_2", prefix + ".b_2"
        return layer_norm(x, self.t[a], self.t[b], self.g(a), self.g(b), gemm_only)

    def ln_res(self, x, prefix, gemm_only=True):
        """(x to use as the sublayer's residual, LayerNorm(x))."""
        a, b = prefix + ".a_2", prefix + ".b_2"
        return layer_norm_res(x, self.t[a], self.t[b], self.g(a), self.g(b), gemm_only)


def _sublayer_linear(P, drop, x_in, wname, residual):
    """residual + dropout(x_in w^T + b): the residual rides in the GEMM epilogue when dropout is off."""
    if not drop.on or drop.p <= 0.0:
        return P.lin(x_in, wname, residual=residual)
    if _COMPUTE["dtype"] == torch.bfloat16:
        return P.lin(x_in, wname, residual=residual, drop=drop.site())
    return drop(P.lin(x_in, wname), residual)


def _ffn(P, pre, drop, n, x):
    fused = drop.on and drop.p > 0.0 and _COMPUTE["dtype"] == torch.bfloat16
    separate = drop.on and drop.p > 0.0 and not fused
    h = P.lin(n, pre + ".w_1", relu=True, drop=drop.site() if fused else None, shadow=True, masked_grad=not separate)   # h feeds w_2 only
    if separate:
        h = drop(h)
    return _sublayer_linear(P, drop, h, pre + ".w_2", x)


def encode_memory(P, cfg, att_feats, att_len, drop):
    """att_embed (TransformerModel.py:1642-1645 + pack_wrapper AttModel.py:36-44) and the encoder stack
    (:1366-1383): returns memory [B*R, d]."""
    B, R, Fdim = att_feats.shape
    d = cfg.d_model
    x = P.lin(att_feats.reshape(B * R, Fdim), "att_embed.0", relu=True, row_len=att_len, rpg=R)
    if drop.on:
        x = drop(x, None, drop.p_att)
    sb = 1 if att_len is not None else 0
    for l in range(cfg.N_enc):
        p = f"model.encoder.layers.{l}"
        xr, n = P.ln_res(x, p + ".sublayer.0.norm")
        qkv = P.lin_packed(n, p + ".self_attn", (0, 1, 2))
        ctx = attention(qkv, qkv, 0, d, 2 * d, B, cfg.h, R, R, 1, att_len, sb, 0, 0, drop.attn(), None, True)
        x = _sublayer_linear(P, drop, ctx, p + ".self_attn.linears.3", xr)
        xr, n = P.ln_res(x, p + ".sublayer.1.norm")
        x = _ffn(P, p + ".feed_forward", drop, n, xr)
    return P.ln(x, "model.encoder.norm")


def _cross(P, pre, cfg, drop, n, x, memory, kv_cache, B, Lq, R, spi, att_len_cap, seg=None, seg_img=None, grad_bf16=True):
    """x + src_attn(n, memory, memory): the image's keys are shared by its captions (kdiv) and, because they depend
    on the memory and the layer only, by the SA and the NA pass of the same layer (kv_cache).  ``seg``: unpadded query rows
    (then ``att_len_cap`` holds one key count per ROW)."""
    d = cfg.d_model
    q = P.lin(n, pre + ".linears.0", shadow=True)
    if pre not in kv_cache:
        kv_cache[pre] = P.lin_packed(memory, pre, (1, 2))
    if seg is not None and seg_img is not None and _COMPUTE["dtype"] == torch.bfloat16:
        # unpadded rows, bf16 kernels: one item per IMAGE (its captions' rows are one contiguous run) instead of one per caption --
        # fuller 16-row tiles in the forward, and the MFMA backward walks runs in chunks anyway
        ctx = attention(q, kv_cache[pre], 0, 0, d, B // spi, cfg.h, seg_img[2], R, 1, att_len_cap, 0, 1, 0, drop.attn(),
                        (seg_img[0], seg_img[1], False) + tuple(seg[2:3]), True)
        return _sublayer_linear(P, drop, ctx, pre + ".linears.3", x)
    sb = 0 if seg is not None else (1 if att_len_cap is not None else 0)
    ctx = attention(q, kv_cache[pre], 0, 0, d, B, cfg.h, Lq, R, spi, att_len_cap, sb, (1 if seg is not None else 0), 0, drop.attn(),
                    None if seg is None else (seg[0], seg[1], False) + tuple(seg[2:3]), grad_bf16)
    return _sublayer_linear(P, drop, ctx, pre + ".linears.3", x)


def decode_rows(P, cfg, drop, x, memory, kv_cache, N, S, R, spi, klen_self, att_len_cap, out_gemm_only=True, seg=None, seg_img=None):
    """Decoder stack + final norm (TransformerModel.py:1386-1413) over N captions x S positions; self-attention
    row (n, i) sees keys < klen_self[n, i].  ``out_gemm_only`` False: the output is also read outside a GEMM (row gather).

    ``seg`` = (row_start int32 [N], row_count int32 [N]): ``x`` holds the captions' REAL positions only, caption n in rows
    row_start[n] .. +row_count[n] (rows past the last caption are padding nobody attends to); klen_self / att_len_cap are then
    per row.  Every other op of the stack is row-wise, so only the two attentions know about the layout.
    ``seg_img`` = (row_start int32 [images], row_count int32 [images], max rows of an image): the same rows grouped per image,
    for the cross-attention (see _cross)."""
    d = cfg.d_model
    for l in range(cfg.N_dec):
        p = f"model.decoder.layers.{l}"
        xr, n_ = P.ln_res(x, p + ".sublayer.0.norm")
        qkv = P.lin_packed(n_, p + ".self_attn", (0, 1, 2))
        if seg is None:
            ctx = attention(qkv, qkv, 0, d, 2 * d, N, cfg.h, S, S, 1, klen_self, S, 1, 0, drop.attn(), None, True)
        else:
            ctx = attention(qkv, qkv, 0, d, 2 * d, N, cfg.h, S, S, 1, klen_self, 0, 1, 0, drop.attn(), (seg[0], seg[1], True) + tuple(seg[2:3]), True)
        x = _sublayer_linear(P, drop, ctx, p + ".self_attn.linears.3", xr)
        xr, n_ = P.ln_res(x, p + ".sublayer.1.norm")
        x = _cross(P, p + ".src_attn", cfg, drop, n_, xr, memory, kv_cache, N, S, R, spi, att_len_cap, seg, seg_img)
        xr, n_ = P.ln_res(x, p + ".sublayer.2.norm")
        x = _ffn(P, p + ".feed_forward", drop, n_, xr)
    return P.ln(x, "model.decoder.norm", gemm_only=out_gemm_only)


def bound_teacher_forced(P, cfg, drop, x_in, memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap):
    """The teacher-forced bound passes of one branch as one batch.

    x_in [N*L, d] is the embedded bound input; klen_pass int32 [N, Pmax] the number of keys the [LEN] row of
    caption n sees in pass i.  Returns (len_logp [N, Pmax, 20], syn_logp [N, Pmax, 10]).  Follows
    LengthPredictorLayer (TransformerModel.py:1025-1029) + LengthPredictor_UIC.forward (:367-383), row 0 only."""
    d, H = cfg.d_model, cfg.h
    Pm = klen_pass.shape[1]
    lp = "model.length_predictor"
    p = lp + ".LengthPredictor.0"
    n_all = P.ln(x_in, p + ".sublayer.0.norm", gemm_only=False)         # row 0 is also read as float32 below
    kv = P.lin_packed(n_all, p + ".self_attn", (1, 2))
    x0 = x_in.view(N, L, d)[:, 0, :]
    n0 = n_all.view(N, L, d)[:, 0, :].contiguous()
    q0 = P.lin(n0, p + ".self_attn.linears.0", shadow=True)
    qv = q0.unsqueeze(1).expand(N, Pm, d).reshape(N * Pm, d)
    if _shadow(q0) is not None:                                 # bf16 mode: the virtual queries' bf16 copy, so that the bf16 / MFMA attention kernels run
        _register_shadow(qv, _shadow(q0).unsqueeze(1).expand(N, Pm, d).reshape(N * Pm, d))
    xv = x0.unsqueeze(1).expand(N, Pm, d).reshape(N * Pm, d).contiguous()     # (one pass: reshape alone would hand out the strided row-0 view)
    ctx = attention(qv, kv, 0, 0, d, N, H, Pm, L, 1, klen_pass, Pm, 1, 0, drop.attn())
    x = _sublayer_linear(P, drop, ctx, p + ".self_attn.linears.3", xv)
    xr, n_ = P.ln_res(x, p + ".sublayer.1.norm")
    x = _cross(P, p + ".src_attn", cfg, drop, n_, xr, memory, kv_cache, N, Pm, R, spi, att_len_cap)
    xr, n_ = P.ln_res(x, p + ".sublayer.2.norm")
    x = _ffn(P, p + ".ff", drop, n_, xr)
    o = P.ln(x, lp + ".norm")
    len_lp, syn_lp = _bound_heads(P, drop, o)
    return len_lp.view(N, Pm, -1), syn_lp.view(N, Pm, -1)


_ZERO_CONST: dict = {}


def _zero_const(ref: torch.Tensor, *shape) -> torch.Tensor:
    """A constant block of zeros (padding of a concatenation; never written, no gradient): made once per shape, not filled by a launch of its own
    in every step."""
    key = (ref.device, ref.dtype, shape)
    z = _ZERO_CONST.get(key)
    if z is None:
        z = _ZERO_CONST[key] = torch.zeros(*shape, dtype=ref.dtype, device=ref.device)
    return z


def _bound_heads(P, drop, o):
    """Length / syntactic classifiers on the normalised [LEN] rows ``o`` [M, d] (LengthPredictor_UIC.forward TransformerModel.py:376-379)."""
    lp = "model.length_predictor"
    d = o.shape[1]
    # heads: hidden 100 is not a multiple of the GEMM K granule -> both first layers side by side in one padded GEMM
    # (small tensors; their gradients go through autograd's cat/slice nodes)
    w1l, w1s = P[lp + ".Length_classifier1.weight"], P[lp + ".Syntactic_classifier1.weight"]
    hh = w1l.shape[0]
    Hp = _pad_k(2 * hh)
    w1 = torch.cat([w1l, w1s, _zero_const(w1l, Hp - 2 * hh, d)], 0)
    b1 = torch.cat([P[lp + ".Length_classifier1.bias"], P[lp + ".Syntactic_classifier1.bias"], _zero_const(w1l, Hp - 2 * hh)], 0)
    hid = linear(o, w1, b1, relu=True)
    if drop.on and drop.p > 0.0:
        hid = drop(hid)
    w2l, w2s = P[lp + ".Length_classifier2.weight"], P[lp + ".Syntactic_classifier2.weight"]
    w2l_p = torch.cat([w2l, _zero_const(w2l, w2l.shape[0], Hp - hh)], 1)
    w2s_p = torch.cat([_zero_const(w2s, w2s.shape[0], hh), w2s, _zero_const(w2s, w2s.shape[0], Hp - 2 * hh)], 1)
    len_lp = log_softmax(linear(hid, w2l_p, P[lp + ".Length_classifier2.bias"]))
    syn_lp = log_softmax(linear(hid, w2s_p, P[lp + ".Syntactic_classifier2.bias"]))
    return len_lp, syn_lp


def bound_teacher_forced_dense(P, cfg, drop, x_in, memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap):
    """The teacher-forced bound passes for a bounding network of N_len >= 2 layers (configs/uic_sd_N2.yml): the upper layers read the
    lower layers' outputs of EVERY visible row, so a pass is the whole L-row sequence through every layer under that pass's tgt_mask
    (LengthPredictor_UIC.forward TransformerModel.py:367-375 as called from :476-513 / :532-565), not a row-0 query.  The Pmax passes
    of a caption run as Pmax sequences of one batch: same input rows, per-pass key counts.

    tgt_mask of pass i (:478-506) as key-prefix counts: row 0 sees keys < cum[i]; row r >= 1 sees through the end of the laid-out block
    it starts in or before -- cum[min(i, j(r))], j(r) = #{k : cum[k] <= r} -- and at least key 0 (:486)."""
    d, H = cfg.d_model, cfg.h
    Pm = klen_pass.shape[1]
    dev = x_in.device
    cum = klen_pass.long()                                     # [N, Pm]: cum[n, i] = 1 + sum of the first i phrase lengths
    r = torch.arange(L, device=dev)
    j = (cum.unsqueeze(2) <= r.view(1, 1, L)).sum(1)           # [N, L]
    kstar = torch.minimum(torch.arange(Pm, device=dev).view(1, Pm, 1), j.view(N, 1, L).clamp(max=Pm - 1))
    klen = cum.gather(1, kstar.reshape(N, Pm * L)).view(N, Pm, L)
    klen[:, :, 0] = cum
    klen = klen.clamp(max=L).to(torch.int32).reshape(N * Pm, L).contiguous()
    M = N * Pm
    x = x_in.view(N, 1, L, d).expand(N, Pm, L, d).reshape(M * L, d).contiguous()
    cap = None if att_len_cap is None else att_len_cap.repeat_interleave(Pm).contiguous()
    lp = "model.length_predictor"
    for l in range(cfg.N_len):
        p = f"{lp}.LengthPredictor.{l}"
        xr, n_ = P.ln_res(x, p + ".sublayer.0.norm")
        qkv = P.lin_packed(n_, p + ".self_attn", (0, 1, 2))
        ctx = attention(qkv, qkv, 0, d, 2 * d, M, H, L, L, 1, klen, L, 1, 0, drop.attn(), None, True)
        x = _sublayer_linear(P, drop, ctx, p + ".self_attn.linears.3", xr)
        xr, n_ = P.ln_res(x, p + ".sublayer.1.norm")
        x = _cross(P, p + ".src_attn", cfg, drop, n_, xr, memory, kv_cache, M, L, R, spi * Pm, cap)
        xr, n_ = P.ln_res(x, p + ".sublayer.2.norm")
        x = _ffn(P, p + ".ff", drop, n_, xr)
    o_all = P.ln(x, lp + ".norm", gemm_only=False)
    o = o_all.view(M, L, d)[:, 0, :].contiguous()
    len_lp, syn_lp = _bound_heads(P, drop, o)
    return len_lp.view(N, Pm, -1), syn_lp.view(N, Pm, -1)


# Host-side facts about the batch the caller already knows (set by the trainer from the collate's output) so that the
# forward pass needs no device->host read: {"max_phrase_num": int, "max_tokens": int}.  Consumed (cleared) by the next
# forward_uic.  "max_tokens" (the longest caption of the batch) additionally lets the two decoder passes and the vocabulary
# projection run over that many positions instead of seq_length: positions past a caption's last token are neither attended
# by earlier ones nor counted by the criterion, so the loss and every gradient are unchanged (dynamic padding); the two token
# tensors then come back as [N, max_tokens, V].  "token_rows" (int64 device tensor of flat indices n * max_tokens + t of the
# real caption tokens, zero-padded to a fixed length) goes one step further for the vocabulary projection: only those rows are
# projected, the two token tensors come back as [len(token_rows), V] (criterion_uic_compact is their criterion).
HINTS: dict = {}


def _scoped_compute_dtype(fn):
    """The GEMM compute dtype is a module global while a forward pass builds its graph (the Functions record what they need
    for their backward); restore the previous value on the way out so that nothing leaks into later op-level calls."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        prev = _COMPUTE["dtype"]
        try:
            return fn(*args, **kwargs)
        finally:
            _COMPUTE["dtype"] = prev
    return wrapper


def bound_pass_klen(phrase_num: torch.Tensor, phrase_length: torch.Tensor, max_phrase_num: Optional[int] = None):
    """Key count of the [LEN] row per (caption, pass), TransformerModel.py:493-511: pass 0 sees 1 key; pass i >= 1 sees
    1 + sum(phrase_length[n, 1..min(i, phrase_num[n]-1)]).  Also the final ``last`` per caption (:562-564)."""
    N, L = phrase_length.shape
    Pm = int(max_phrase_num) if max_phrase_num else int(phrase_num.max())     # .max() is a device->host sync
    idx = torch.arange(L, device=phrase_length.device).unsqueeze(0)
    pl = torch.where((idx >= 1) & (idx < phrase_num.unsqueeze(1)), phrase_length, torch.zeros_like(phrase_length))
    cum = 1 + pl.cumsum(1)                                     # cum[n, i] = 1 + sum_{t<=i} pl[n, t]
    return cum[:, :Pm].to(torch.int32).contiguous(), cum[:, -1].to(torch.int32).contiguous(), Pm


@_scoped_compute_dtype
def forward_uic(P, cfg, att_feats, labels, att_masks, phrase_num, phrase_length, phrase_syn, extend_phrase_syn_seq, extend_phrase_seq,
                extend_phrase_seq_mask, *, glat_p: float = -1.0, training: bool = False, seed: Optional[int] = None,
                compute_dtype: torch.dtype = torch.float32, step_word: Optional[torch.Tensor] = None):
    """The six log-prob tensors of EncoderDecoder_UIC.forward (TransformerModel.py:413-468):
    (sa_len [N,S+1,20], sa_syn [N,S+1,10], sa_tok [N,S,V], na_len, na_syn, na_tok).  ``P``: a ``Params``."""
    dev = att_feats.device
    S, L, d = cfg.seq_length, cfg.seq_length + 2, cfg.d_model
    dense_bound = cfg.N_len != 1         # a deeper bounding network: whole-sequence passes (bound_teacher_forced_dense), the plain form of the step
    if compute_dtype not in (torch.float32, torch.bfloat16):
        raise hip.BofiHipError(f"training compute dtype {compute_dtype}: float32 or bfloat16")
    _COMPUTE["dtype"] = compute_dtype
    _STEP_CACHE.clear()
    _SHADOW_ONLY.clear()
    if labels.dim() == 3:
        labels = labels.reshape(-1, labels.shape[2])
        phrase_num = phrase_num.reshape(-1)
        phrase_length = phrase_length.reshape(-1, phrase_length.shape[2])
        extend_phrase_syn_seq = extend_phrase_syn_seq.reshape(-1, extend_phrase_syn_seq.shape[2])
        extend_phrase_seq = extend_phrase_seq.reshape(-1, extend_phrase_seq.shape[2])
        extend_phrase_seq_mask = extend_phrase_seq_mask.reshape(-1, extend_phrase_seq.shape[1], extend_phrase_seq.shape[1])
    att_len = None
    if att_masks is not None:                                  # clip_att, AttModel.py:113-120
        max_len = int(att_masks.long().sum(1).max())
        att_feats, att_masks = att_feats[:, :max_len].contiguous(), att_masks[:, :max_len]
        att_len = att_masks.long().sum(1).to(torch.int32).contiguous()
    att_feats = _need(att_feats.float() if att_feats.dtype != torch.float32 else att_feats, "att_feats")
    B, R, _ = att_feats.shape
    N = labels.shape[0]
    if N % B:
        raise hip.BofiHipError(f"{N} captions for {B} images")
    spi = N // B
    drop = _Drop(cfg.dropout, cfg.drop_prob_lm, seed if training else None, step_word)
    memory = encode_memory(P, cfg, att_feats, att_len, drop)
    split = HINTS.pop("split_memory", None)
    if split is not None:
        # two-stage backward (XETrainer, data-parallel): everything behind the encoder hangs off a DETACHED copy of its output, so that
        # loss.backward() stops there with every decoder-side gradient final -- their exchange starts while the encoder's backward
        # (memory.backward(memory_detached.grad)) still runs
        sh = _shadow(memory)
        split["memory"] = memory
        memory = memory.detach().requires_grad_(True)
        if sh is not None:
            _register_shadow(memory, sh)
        split["memory_detached"] = memory
    att_len_cap = None if att_len is None else att_len.repeat_interleave(spi).contiguous()
    kv_cache: dict = {}

    labels, phrase_num, phrase_length = labels.to(dev).long(), phrase_num.to(dev).long(), phrase_length.to(dev).long()
    ext_syn = extend_phrase_syn_seq.to(dev).long().contiguous()
    ext_seq = extend_phrase_seq.to(dev).long().contiguous()
    prepared = HINTS.pop("paired_inputs", None)                # the paired step's index tensors, made by the collate on the host
    if prepared is not None and (glat_p >= 0 or att_len is not None or HINTS.get("paired") is None):
        prepared = None                                        # (the glancing pass and ragged region masks take the device-side prep)
    if prepared is None:
        klen_pass, last, Pm = bound_pass_klen(phrase_num, phrase_length, HINTS.pop("max_phrase_num", None))
    else:
        HINTS.pop("max_phrase_num", None)
        klen_pass, last, Pm = None, None, int(prepared["klen_b"].shape[1])
    Sd = HINTS.pop("max_tokens", None)
    Sd = S if not Sd else max(1, min(S, int(Sd)))               # decoder positions actually computed
    token_rows = HINTS.pop("token_rows", None)
    unpadded = HINTS.pop("unpadded", None)
    streams = HINTS.pop("streams", None)
    paired = HINTS.pop("paired", None)
    pick_labels = HINTS.pop("pick_labels", None)
    bound_fn = bound_teacher_forced
    if dense_bound:
        if prepared is not None:
            klen_pass, last, Pm = bound_pass_klen(phrase_num, phrase_length, Pm)
        prepared = token_rows = unpadded = streams = paired = pick_labels = None
        bound_fn = bound_teacher_forced_dense
    tname, sname = "model.tgt_embed.lut.weight", "model.syn_embed.lut.weight"
    pe = P["model.pos_embed.pe"]

    def emb(tok, syn, Lp, pos=None):
        x = embed(P[tname] if tok is not None else None, P[sname] if syn is not None else None, pe, tok, syn, Lp,
                  P.g(tname) if tok is not None else None, P.g(sname) if syn is not None else None, pos)
        return drop(x) if drop.on and drop.p > 0.0 else x

    def pad_slots(t):                                          # pass i lands in slot i of the returned [:, 1:] view
        out = t.new_zeros(N, L - 1, t.shape[2])
        out[:, :Pm] = t
        return out

    def vocab(x):
        return P.lin(x, "model.generator.proj")

    def token_logprobs(x):
        if token_rows is None:
            return log_softmax(vocab(x)).view(N, Sd, -1)
        return log_softmax(vocab(x.index_select(0, token_rows)))          # [len(token_rows), V]: the real tokens' rows only

    if prepared is not None:
        return _forward_paired(P, cfg, drop, emb, vocab, pad_slots, unpadded, paired, labels, None, phrase_length,
                               ext_syn, ext_seq, None, None, memory, kv_cache, N, L, Sd, R, spi, None, None, glat_p, pick_labels, prepared)
    # --- semi-autoregressive branch (TransformerModel.py:476-530)
    word_seq = labels.clone()
    word_seq[:, 0] = cfg.len_idx
    sa_bound = lambda: bound_fn(P, cfg, drop, emb(word_seq.contiguous(), None, L), memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap)
    if unpadded is not None and paired is not None:
        return _forward_paired(P, cfg, drop, emb, vocab, pad_slots, unpadded, paired, labels, word_seq, phrase_length, ext_syn, ext_seq,
                               extend_phrase_seq_mask.to(dev), last, memory, kv_cache, N, L, Sd, R, spi, att_len_cap, klen_pass, glat_p,
                               pick_labels)
    if unpadded is not None:
        na_bound = lambda: bound_teacher_forced(P, cfg, drop, emb(None, ext_syn, L), memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap)
        (sa_len, sa_syn), sa_tok, (na_len, na_syn), na_tok = _fill_unpadded(
            P, cfg, drop, emb, vocab, unpadded, labels, phrase_length, ext_syn, ext_seq, extend_phrase_seq_mask.to(dev), last, memory,
            kv_cache, N, Sd, R, spi, att_len_cap, glat_p, sa_bound, na_bound, streams)
        return pad_slots(sa_len), pad_slots(sa_syn), sa_tok, pad_slots(na_len), pad_slots(na_syn), na_tok
    sa_len, sa_syn = sa_bound()
    syn_mid = ext_syn[:, 1:1 + Sd].contiguous()
    ext_seq = ext_seq[:, :Sd].contiguous()
    klen_sa = extend_phrase_seq_mask.to(dev).long().sum(-1)[:, :Sd].to(torch.int32).contiguous()  # prefix masks (dataloader.py:414)
    x = decode_rows(P, cfg, drop, emb(ext_seq, syn_mid, Sd), memory, kv_cache, N, Sd, R, spi, klen_sa, att_len_cap, token_rows is None)
    sa_tok = token_logprobs(x)

    # --- non-autoregressive branch (:532-587)
    na_len, na_syn = bound_fn(P, cfg, drop, emb(None, ext_syn, L), memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap)
    klen_na = (last - 1).unsqueeze(1).expand(N, Sd).contiguous()
    fill_in = torch.full((N, Sd), cfg.bos_idx, dtype=torch.int64, device=dev)
    if glat_p >= 0:                                            # glancing input (:437-463): reveal a share of the true tokens
        with torch.no_grad():
            x = decode_rows(P, cfg, drop, emb(fill_in, syn_mid, Sd), memory, dict(kv_cache), N, Sd, R, spi, klen_na, att_len_cap)
            pred = greedy_ids(vocab(x)).view(N, Sd)
            real = labels[:, 1:1 + Sd]
            ntok = phrase_length.sum(1) - 1
            tok_mask = torch.arange(Sd, device=dev).unsqueeze(0) < ntok.unsqueeze(1)
            same = ((pred == real) & tok_mask).sum(1)
            keep_prob = ((ntok - same) / ntok * glat_p).unsqueeze(-1) * tok_mask.float()
            keep = _glance_draws(N, cfg.seq_length, dev)[:, :Sd] < keep_prob
            fill_in = torch.where(keep, real, fill_in).contiguous()
    x = decode_rows(P, cfg, drop, emb(fill_in, syn_mid, Sd), memory, kv_cache, N, Sd, R, spi, klen_na, att_len_cap, token_rows is None)
    na_tok = token_logprobs(x)
    return pad_slots(sa_len), pad_slots(sa_syn), sa_tok, pad_slots(na_len), pad_slots(na_syn), na_tok


@_scoped_compute_dtype
def forward_uic_ss(P, cfg, att_feats, labels, att_masks, phrase_num, phrase_length, phrase_syn, extend_phrase_syn_seq, *, ss_prob: float,
                   draw=None, training: bool = False, seed: Optional[int] = None, compute_dtype: torch.dtype = torch.float32,
                   step_word: Optional[torch.Tensor] = None):
    """The six log-prob tensors of TransformerModel._forward with scheduled sampling on (ss_prob > 0, TransformerModel.py:1760-1766):
    the SA branch is ss_SAIC (:1988-2121), the NA branch the teacher-forced one without a glancing pass.

    ss_SAIC is a per-phrase loop whose every iteration runs a bounding step and a full decode_SA pass WITH the tape.  What it
    differentiates, though, is a function of the loop's final inputs only: the bounding step of iteration i reads the words
    emitted before it (its [LEN] row sees keys < phrase_last), and the decoder rows of phrase i see keys < its own end, all of
    them final once the phrase is placed -- later iterations neither change those inputs nor those rows' masks.  So the loop runs
    here WITHOUT the tape (decisions: greedy slots and tokens, the reference's bookkeeping on the host, ``draw()`` standing for
    its ``random()`` calls in their order), and the tape then sees ONE batched bounding pass over (caption, iteration) queries
    and ONE decoder pass on the final inputs; rows the loop never wrote are zero, as in the reference's zero-initialised
    buffers.  Same outputs and gradients (tests/golden/tiny_ss.npz).  With dropout on, the decisions are taken without dropout
    and the differentiated pass applies it (the reference does both in one stochastic pass)."""
    import random as _random
    import numpy as np
    dev = att_feats.device
    S, L, d = cfg.seq_length, cfg.seq_length + 2, cfg.d_model
    if cfg.N_len != 1:
        raise NotImplementedError("training with N_len >= 2 (see forward_uic)")
    draw = draw if draw is not None else _random.random
    for k in ("max_phrase_num", "max_tokens", "token_rows", "unpadded", "streams", "paired", "pick_labels", "paired_inputs"):
        HINTS.pop(k, None)                                     # the row-list fast paths describe the ground-truth layout: not this pass
    _COMPUTE["dtype"] = compute_dtype
    _STEP_CACHE.clear()
    _SHADOW_ONLY.clear()
    if labels.dim() == 3:
        labels = labels.reshape(-1, labels.shape[2])
        phrase_num = phrase_num.reshape(-1)
        phrase_length = phrase_length.reshape(-1, phrase_length.shape[2])
        phrase_syn = phrase_syn.reshape(-1, phrase_syn.shape[2])
        extend_phrase_syn_seq = extend_phrase_syn_seq.reshape(-1, extend_phrase_syn_seq.shape[2])
    att_len = None
    if att_masks is not None:
        max_len = int(att_masks.long().sum(1).max())
        att_feats, att_masks = att_feats[:, :max_len].contiguous(), att_masks[:, :max_len]
        att_len = att_masks.long().sum(1).to(torch.int32).contiguous()
    att_feats = _need(att_feats.float() if att_feats.dtype != torch.float32 else att_feats, "att_feats")
    B, R, _ = att_feats.shape
    N = labels.shape[0]
    spi = N // B
    att_len_cap = None if att_len is None else att_len.repeat_interleave(spi).contiguous()
    tname, sname = "model.tgt_embed.lut.weight", "model.syn_embed.lut.weight"
    pe = P["model.pos_embed.pe"]
    lab_h = labels.cpu().numpy().astype(np.int64)
    plen_h = phrase_length.cpu().numpy().astype(np.int64)
    psyn_h = phrase_syn.cpu().numpy().astype(np.int64)

    def make_emb(dr):
        def emb(tok, syn, Lp):
            x = embed(P[tname] if tok is not None else None, P[sname] if syn is not None else None, pe, tok, syn, Lp,
                      P.g(tname) if tok is not None else None, P.g(sname) if syn is not None else None, None)
            return dr(x) if dr.on and dr.p > 0.0 else x
        return emb

    t = lambda a, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)

    # ---- the loop, no tape, no dropout (TransformerModel.py:2011-2119)
    ppl = np.zeros((N, L), np.int64); pps = np.full((N, L), cfg.pad_idx, np.int64)
    seq = np.full((N, L), cfg.pad_idx, np.int64)
    ext_len = np.full((N, L), cfg.pad_idx, np.int64); ext_phrase = np.full((N, L), cfg.pad_idx, np.int64)
    ext_syn = np.full((N, L), cfg.pad_idx, np.int64)
    row_end = np.zeros((N, L), np.int64)                       # keys (a prefix) row r of phrase_mask sees
    written = np.zeros((N, L), bool)
    finished = np.zeros(N, bool)
    label_last = np.zeros(N, np.int64); seq_last = np.zeros(N, np.int64); phrase_last = np.ones(N, np.int64)
    seq[:, 0], ppl[:, 0], ext_len[:, 0] = cfg.bos_idx, 1, cfg.len_idx
    klen_iters = []

    def stretch(dst, pl, cur, src, s0, prev):                  # the position-wise copy (:2054-2067, :2079-2092)
        if cur <= prev:
            dst[pl:pl + cur] = src[s0 + prev - cur: s0 + prev]
        else:
            pre_less, times, copied = prev - (cur % prev), cur // prev, 0
            for k in range(prev):
                n = times if k < pre_less else times + 1
                dst[pl + copied: pl + copied + n] = src[s0 + k]
                copied += n

    off = _Drop(cfg.dropout, cfg.drop_prob_lm, None, None)
    emb0 = make_emb(off)
    with torch.no_grad():
        memory0 = encode_memory(P, cfg, att_feats, att_len, off)
        kv0: dict = {}
        for i in range(1, L):
            klen_iters.append(phrase_last.copy())
            len_lp, syn_lp = bound_teacher_forced(P, cfg, off, emb0(t(ext_len), None, L), memory0, kv0, N, L, R, spi,
                                                  t(phrase_last.reshape(N, 1), torch.int32), att_len_cap)
            len_n = len_lp.view(N, -1).cpu().numpy().argmax(1)           # first maximal index, as torch.max on the CPU
            syn_n = syn_lp.view(N, -1).cpu().numpy().argmax(1)
            for j in range(N):
                if finished[j]:
                    continue
                ln, sn, pl = int(len_n[j]), int(syn_n[j]), int(phrase_last[j])
                if ln == 0 or sn < 4 or sn > 6 or plen_h[j, i] == 0:     # SA_SYN_LOWER / SA_SYN_UPPER
                    finished[j] = True
                    continue
                if ln + pl >= L - 1:
                    ln = L - 1 - pl
                    finished[j] = True
                ppl[j, i], pps[j, i] = ln, sn
            for j in range(N):
                if ppl[j, i] == 0:
                    continue
                pl = int(phrase_last[j])
                if draw() < ss_prob:
                    cur = int(ppl[j, i])
                    ext_syn[j, pl:pl + cur] = pps[j, i]
                    if draw() < 0.5:
                        stretch(ext_phrase[j], pl, cur, seq[j], int(seq_last[j]), int(ppl[j, i - 1]))
                    else:
                        ext_phrase[j, pl:pl + cur] = cfg.bos_idx
                else:
                    cur = min(int(plen_h[j, i]), L - 1 - pl)
                    ppl[j, i] = cur
                    ext_syn[j, pl:pl + cur] = psyn_h[j, i]
                    stretch(ext_phrase[j], pl, cur, lab_h[j], int(label_last[j]), int(plen_h[j, i - 1]))
                row_end[j, pl:] = pl + cur
            klen_dec = np.maximum(row_end[:, 1:1 + S] - 1, 0)
            if (klen_dec == 0).any():
                raise FloatingPointError("ss_SAIC: a caption without any key -- NaN log-probs; the reference returns a malformed tuple here "
                                         "(TransformerModel.py:2103-2105)")
            x = decode_rows(P, cfg, off, emb0(t(ext_phrase[:, 1:-1]), t(ext_syn[:, 1:-1]), S), memory0, kv0, N, S, R, spi,
                            t(klen_dec, torch.int32), att_len_cap)
            tok = greedy_ids(P.lin(x, "model.generator.proj")).view(N, S).cpu().numpy()
            for j in range(N):
                cur = int(ppl[j, i])
                if cur == 0:
                    continue
                pl = int(phrase_last[j])
                seq[j, pl:pl + cur] = tok[j, pl - 1: pl - 1 + cur]
                ext_len[j, pl:pl + cur] = tok[j, pl - 1: pl - 1 + cur]
                written[j, pl:pl + cur] = True
                phrase_last[j] = pl + cur
                seq_last[j] += int(ppl[j, i - 1])
                label_last[j] += int(plen_h[j, i - 1])
            if finished.all():
                break
    n_it = len(klen_iters)

    # ---- the differentiated passes
    _STEP_CACHE.clear()
    _SHADOW_ONLY.clear()
    P._packed.clear()                                          # (stacked q|k|v operands made without the tape above carry no gradient)
    drop = _Drop(cfg.dropout, cfg.drop_prob_lm, seed if training else None, step_word)
    emb = make_emb(drop)
    memory = encode_memory(P, cfg, att_feats, att_len, drop)
    kv_cache: dict = {}

    def pad_slots(x, Pm):
        out = x.new_zeros(N, L - 1, x.shape[2])
        out[:, :Pm] = x
        return out

    klen_pass = t(np.stack(klen_iters, 1), torch.int32)                     # [N, iterations]: keys of the [LEN] row per iteration
    sa_len, sa_syn = bound_teacher_forced(P, cfg, drop, emb(t(ext_len), None, L), memory, kv_cache, N, L, R, spi, klen_pass, att_len_cap)
    klen_dec = t(np.maximum(row_end[:, 1:1 + S] - 1, 0), torch.int32)
    x = decode_rows(P, cfg, drop, emb(t(ext_phrase[:, 1:-1]), t(ext_syn[:, 1:-1]), S), memory, kv_cache, N, S, R, spi, klen_dec, att_len_cap)
    sa_tok = log_softmax(P.lin(x, "model.generator.proj")).view(N, S, -1)
    sa_tok = torch.where(t(written[:, 1:1 + S], torch.bool).unsqueeze(-1), sa_tok, torch.zeros_like(sa_tok))

    # ---- non-autoregressive branch, teacher-forced (:1764-1766)
    labels_d, phrase_num_d, phrase_length_d = labels.to(dev).long(), phrase_num.to(dev).long(), phrase_length.to(dev).long()
    ext_syn_gt = extend_phrase_syn_seq.to(dev).long().contiguous()
    klen_gt, last, Pm = bound_pass_klen(phrase_num_d, phrase_length_d)
    na_len, na_syn = bound_teacher_forced(P, cfg, drop, emb(None, ext_syn_gt, L), memory, kv_cache, N, L, R, spi, klen_gt, att_len_cap)
    klen_na = (last - 1).unsqueeze(1).expand(N, S).contiguous()
    fill_in = torch.full((N, S), cfg.bos_idx, dtype=torch.int64, device=dev)
    x = decode_rows(P, cfg, drop, emb(fill_in, ext_syn_gt[:, 1:1 + S].contiguous(), S), memory, kv_cache, N, S, R, spi, klen_na, att_len_cap)
    na_tok = log_softmax(P.lin(x, "model.generator.proj")).view(N, S, -1)
    HINTS["ss_trace"] = dict(iters=n_it, emitted=seq[:, 1:-1].copy(), predict_phrase_length=ppl.copy())
    return pad_slots(sa_len, n_it), pad_slots(sa_syn, n_it), sa_tok, pad_slots(na_len, Pm), pad_slots(na_syn, Pm), na_tok


def _glance_draws(N: int, S: int, dev) -> torch.Tensor:
    """The uniform draws of the glancing pass, one per (caption, position) as the reference takes them (torch.rand over
    [N, S], TransformerModel.py:455).  HINTS["glat_uniform"] injects them (parity tests feed both sides the same numbers)."""
    u = HINTS.pop("glat_uniform", None)
    if u is None:
        return torch.rand(N, S, device=dev)
    u = u.to(dev).float().reshape(N, S)
    return u


_ORDER_CACHE: dict = {}


def _paired_order(N, spi, dev):
    """Index tensors of the image-major paired caption order for N captions, spi per image: (caption of paired slot p [2N],
    is-NA flag [2N, 1], slot of caption n's SA copy [N], of its NA copy [N]).  Functions of the batch SHAPE only: made once and
    kept (inside a captured step they would be a dozen tiny launches per replay)."""
    key = (N, spi, str(dev))
    hit = _ORDER_CACHE.get(key)
    if hit is None:
        pn = torch.arange(2 * N, device=dev)
        j = pn % (2 * spi)
        cap_n = (pn // (2 * spi)) * spi + j % spi
        cap_na = (j >= spi).unsqueeze(1)
        n_all = torch.arange(N, device=dev)
        at_sa = (n_all // spi) * (2 * spi) + n_all % spi
        hit = _ORDER_CACHE[key] = (cap_n, cap_na, at_sa, at_sa + spi)
        if len(_ORDER_CACHE) > 64:
            _ORDER_CACHE.pop(next(iter(_ORDER_CACHE)))
    return hit


def _forward_paired(P, cfg, drop, emb, vocab, pad_slots, unpadded, paired, labels, word_seq, phrase_length, ext_syn, ext_seq, ext_mask,
                    last, memory, kv_cache, N, L, Sd, R, spi, att_len_cap, klen_pass, glat_p, pick_labels=None, prepared=None):
    """forward_uic with the SA and the NA branch as ONE batch: one bound pass over 2N captions and one decoder pass over both
    branches' rows, instead of two of each.

    At these sizes (a few thousand rows, d_model 512) every GEMM / LayerNorm / attention launch is bound by its own latency
    chain, not by its rows: a 2 560-row projection takes as long as a 5 120-row one (measured, 15.4 us both).  The two branches
    run the same layers with the same weights on different inputs, so their rows share every launch: half the launches of the
    decoder and bound stacks, forward and backward.  Captions are ordered image-major -- image i's ``spi`` SA copies, then its
    ``spi`` NA copies -- so that the 2*spi captions sharing an image's cross-attention keys stay adjacent (kdiv = 2 spi).

    ``unpadded`` as in _fill_unpadded (the single-branch row list: the glancing pass runs on it); ``paired`` =
    (pair_start int32 [2N], pair_count int32 [2N], pair_src int64 [T2], pair_na bool [T2] [, tail]): row r of the paired list is
    row pair_src[r] of the single list, in the NA branch iff pair_na[r]; T2 padded like T.  Returns the six tensors of
    forward_uic with BOTH token entries being the paired log-probs [T2, V] (criterion_uic_compact with a pair of weights) and the
    four bound outputs as [N, Pm, .] -- Pm passes, not padded out to seq_length + 1 slots."""
    dev = labels.device
    row_start, row_count, row_cap, row_pos = unpadded[:4]
    pair_start, pair_count, pair_src, pair_na = paired[:4]
    if prepared is not None:
        # every index tensor comes from the collate (XETrainer.add_token_rows, numpy): nothing to derive on the device
        cap_n, cap_na, at_sa, at_na = _paired_order(N, spi, dev)
        for pre in [f"model.decoder.layers.{l}.src_attn" for l in range(cfg.N_dec)]:
            if pre not in kv_cache:
                kv_cache[pre] = P.lin_packed(memory, pre, (1, 2))
        len_lp, syn_lp = bound_teacher_forced(P, cfg, drop, emb(prepared["tok_b"], prepared["syn_b"], L), memory, kv_cache, 2 * N, L, R, 2 * spi,
                                              prepared["klen_b"], None)
        sa_len, sa_syn = len_lp.index_select(0, at_sa), syn_lp.index_select(0, at_sa)
        na_len, na_syn = len_lp.index_select(0, at_na), syn_lp.index_select(0, at_na)
        seg2 = (pair_start, pair_count) + tuple(paired[4:5])
        img = (prepared["img_start"], prepared["img_count"], 2 * spi * Sd)
        x = decode_rows(P, cfg, drop, emb(prepared["tok2"], prepared["syn2"], Sd, prepared["pos2"]), memory, kv_cache, 2 * N, Sd, R, 2 * spi,
                        prepared["klen2"], None, True, seg2, img)
        if pick_labels is not None:
            tok_all, picked = log_softmax_pick(vocab(x), pick_labels, True)
            tok_all._bofi_picked = (picked, pick_labels)
        else:
            tok_all = log_softmax(vocab(x))
        return sa_len, sa_syn, tok_all, na_len, na_syn, tok_all
    with torch.no_grad():
        at = row_cap * L + row_pos
        syn_c = ext_syn.reshape(-1)[at + 1]
        seq_c = ext_seq.reshape(-1)[row_cap * ext_seq.shape[1] + row_pos]
        klen_full = ext_mask.long().sum(-1)
        klen_sa = klen_full.reshape(-1)[row_cap * klen_full.shape[1] + row_pos].to(torch.int32)
        klen_na = (last - 1)[row_cap].to(torch.int32)
        cross_len = None if att_len_cap is None else att_len_cap[row_cap]
        T = row_cap.numel()
        cap_n, cap_na, at_sa, at_na = _paired_order(N, spi, dev)  # captions in paired order (constants of the batch shape)
        tok_b, syn_b = word_seq[cap_n], ext_syn[cap_n]
        none = torch.full_like(tok_b, -1)                         # the SA bound input has no syntactic term, the NA one no token term
        tok_b = torch.where(cap_na, none, tok_b).contiguous()
        syn_b = torch.where(cap_na, syn_b, none).contiguous()
        klen_b = klen_pass[cap_n].contiguous()
        att_b = None if att_len_cap is None else att_len_cap[cap_n].contiguous()
    # the image's cross-attention K/V of every layer that reads it, once, with the tape (the glancing pass below has none)
    for pre in [f"model.decoder.layers.{l}.src_attn" for l in range(cfg.N_dec)]:
        if pre not in kv_cache:
            kv_cache[pre] = P.lin_packed(memory, pre, (1, 2))
    len_lp, syn_lp = bound_teacher_forced(P, cfg, drop, emb(tok_b, syn_b, L), memory, kv_cache, 2 * N, L, R, 2 * spi, klen_b, att_b)
    sa_len, sa_syn = len_lp.index_select(0, at_sa), syn_lp.index_select(0, at_sa)
    na_len, na_syn = len_lp.index_select(0, at_na), syn_lp.index_select(0, at_na)

    fill_in = torch.full((T,), cfg.bos_idx, dtype=torch.int64, device=dev)
    if glat_p >= 0:                                               # glancing input (TransformerModel.py:437-463), NA rows only
        with torch.no_grad():
            seg1 = (row_start, row_count) + tuple(unpadded[4:5])
            x = decode_rows(P, cfg, drop, emb(fill_in, syn_c.contiguous(), Sd, row_pos), memory, dict(kv_cache), N, Sd, R, spi,
                            klen_na.contiguous(), None if cross_len is None else cross_len.contiguous(), True, seg1)
            pred = greedy_ids(vocab(x)).view(T)
            real = labels.reshape(-1)[at + 1]
            ntok = phrase_length.sum(1) - 1
            in_cap = torch.arange(T, device=dev) < (row_start[-1] + row_count[-1])
            same = torch.zeros(N, dtype=torch.int64, device=dev).index_add_(0, row_cap, ((pred == real) & in_cap).long())
            keep_prob = ((ntok - same) / ntok * glat_p)[row_cap] * in_cap.float()
            keep = _glance_draws(N, cfg.seq_length, dev)[row_cap, row_pos] < keep_prob
            fill_in = torch.where(keep, real, fill_in)
    with torch.no_grad():
        tok2 = torch.where(pair_na, fill_in[pair_src], seq_c[pair_src]).contiguous()
        syn2 = syn_c[pair_src].contiguous()
        pos2 = row_pos[pair_src].contiguous()
        klen2 = torch.where(pair_na, klen_na[pair_src], klen_sa[pair_src]).contiguous()
        cross2 = None if cross_len is None else cross_len[pair_src].contiguous()
    seg2 = (pair_start, pair_count) + tuple(paired[4:5])
    with torch.no_grad():                                         # the same rows per image: its 2 spi captions are adjacent
        img = (pair_start.view(-1, 2 * spi)[:, 0].contiguous(), pair_count.view(-1, 2 * spi).sum(1).to(torch.int32).contiguous(), 2 * spi * Sd)
    x = decode_rows(P, cfg, drop, emb(tok2, syn2, Sd, pos2), memory, kv_cache, 2 * N, Sd, R, 2 * spi, klen2, cross2, True, seg2, img)
    if pick_labels is not None:                                   # the criterion's token labels are known: pick while the row is at hand
        tok_all, picked = log_softmax_pick(vocab(x), pick_labels, True)
        tok_all._bofi_picked = (picked, pick_labels)
    else:
        tok_all = log_softmax(vocab(x))
    # the four bound outputs stay [N, Pm, .] (Pm >= the batch's largest phrase count: the slots past it carry no loss weight;
    # criterion_uic_compact reads as many label slots as there are outputs)
    return sa_len, sa_syn, tok_all, na_len, na_syn, tok_all


class _Fork:
    """The four branches of the training forward (SA bound, SA fill, NA bound, glancing + NA fill) on HIP streams of their own.

    They only share the image memory and its cross-attention K/V; most of their kernels are far too small to fill 256 CUs, so
    running them side by side is what keeps the chip busy.  Autograd runs each node's backward on the stream of its forward and
    orders the streams where gradients meet, so the backward forks the same way; inside a captured step the streams become
    parallel branches of the hipGraph.  Parameter gradients are accumulated with float atomics into the flat bucket, which is
    what makes concurrent branches over the SAME weights (the decoder serves both fills) legal."""

    def __init__(self, streams):
        self.main = torch.cuda.current_stream()
        self.side = list(streams)

    def share(self, tensors):
        """Tensors made on the main stream that the side streams read (and their bf16 shadows): tell the allocator."""
        for t in tensors:
            if t is None:
                continue
            sh = _shadow(t)
            for s in self.side:
                t.record_stream(s)
                if sh is not None:
                    sh.record_stream(s)

    def begin(self):
        for s in self.side:
            s.wait_stream(self.main)

    def on(self, i):
        return torch.cuda.stream(self.side[i])

    def join(self, outs):
        for s in self.side:
            self.main.wait_stream(s)
        for t in outs:
            t.record_stream(self.main)
            sh = _shadow(t)
            if sh is not None:
                sh.record_stream(self.main)


class _NoFork:
    def share(self, tensors): pass
    def begin(self): pass
    def on(self, i): return contextlib.nullcontext()
    def join(self, outs): pass


def _fill_unpadded(P, cfg, drop, emb, vocab, unpadded, labels, phrase_length, ext_syn, ext_seq, ext_mask, last, memory, kv_cache, N, Sd, R,
                   spi, att_len_cap, glat_p, sa_bound, na_bound, streams=None):
    """The two bound passes and decode_SA / decode_NA (+ the glancing pass) of forward_uic, the decoder over the captions' real
    positions only.

    ``unpadded`` = (row_start int32 [N], row_count int32 [N], row_cap int64 [T], row_pos int64 [T] [, tail]): row r of the decoder
    batch is position row_pos[r] of caption row_cap[r]; T is padded (to a multiple the caller chooses) with rows that belong
    to no caption.  A padded [N, Sd] batch spends most of its rows on positions past the captions' ends (synthetic
    COCO-like captions: ~45% real); their outputs carry zero loss weight and nothing attends to them, so leaving them out
    changes neither the loss nor any gradient.  Returns (sa_bound(), sa_tok [T, V], na_bound(), na_tok [T, V]); the op order
    (SA bound, SA fill, NA bound, glancing, NA fill) is the padded path's, so the dropout streams agree with it site by site.
    ``streams``: three side streams -> the branches run concurrently (see _Fork)."""
    dev = labels.device
    row_start, row_count, row_cap, row_pos = unpadded[:4]
    seg = (row_start, row_count) + tuple(unpadded[4:5])        # (+ the host's bound on the number of padding rows, if given)
    L = labels.shape[1]
    with torch.no_grad():
        at = row_cap * L + row_pos                                # (caption, position) in the [N, L] loader arrays
        syn_c = ext_syn.reshape(-1)[at + 1].contiguous()
        seq_c = ext_seq.reshape(-1)[row_cap * ext_seq.shape[1] + row_pos].contiguous()
        klen_full = ext_mask.long().sum(-1)                       # [N, S'] prefix masks (dataloader.py:414)
        klen_sa = klen_full.reshape(-1)[row_cap * klen_full.shape[1] + row_pos].to(torch.int32).contiguous()
        klen_na = (last - 1)[row_cap].to(torch.int32).contiguous()
        cross_len = None if att_len_cap is None else att_len_cap[row_cap].contiguous()
        real = labels.reshape(-1)[at + 1]
        ntok = phrase_length.sum(1) - 1
        T = row_cap.numel()
        in_cap = torch.arange(T, device=dev) < (row_start[-1] + row_count[-1])
    fork = _NoFork()
    if streams:
        if len(streams) < 3:
            raise hip.BofiHipError("three side streams expected")
        fork = _Fork(streams)
        # what every branch reads is made here, on the main stream, before the streams part: the image's cross-attention K/V of
        # the decoder layers and of the bound layer
        for pre in [f"model.decoder.layers.{l}.src_attn" for l in range(cfg.N_dec)] + ["model.length_predictor.LengthPredictor.0.src_attn"]:
            if pre not in kv_cache:
                kv_cache[pre] = P.lin_packed(memory, pre, (1, 2))
        fork.share([memory, syn_c, seq_c, klen_sa, klen_na, cross_len, real, ntok, in_cap] + list(kv_cache.values()))
        fork.begin()
    bound_sa = sa_bound()
    with fork.on(0):
        x = decode_rows(P, cfg, drop, emb(seq_c, syn_c, Sd, row_pos), memory, kv_cache, N, Sd, R, spi, klen_sa, cross_len, True, seg)
        sa_tok = log_softmax(vocab(x))
    with fork.on(1):
        bound_na = na_bound()
    with fork.on(2):
        fill_in = torch.full((T,), cfg.bos_idx, dtype=torch.int64, device=dev)
        if glat_p >= 0:                                           # glancing input (TransformerModel.py:437-463)
            with torch.no_grad():
                x = decode_rows(P, cfg, drop, emb(fill_in, syn_c, Sd, row_pos), memory, dict(kv_cache), N, Sd, R, spi, klen_na, cross_len, True, seg)
                pred = greedy_ids(vocab(x)).view(T)
                same = torch.zeros(N, dtype=torch.int64, device=dev).index_add_(0, row_cap, ((pred == real) & in_cap).long())
                keep_prob = ((ntok - same) / ntok * glat_p)[row_cap] * in_cap.float()
                keep = _glance_draws(N, cfg.seq_length, dev)[row_cap, row_pos] < keep_prob
                fill_in = torch.where(keep, real, fill_in).contiguous()
        x = decode_rows(P, cfg, drop, emb(fill_in, syn_c, Sd, row_pos), memory, kv_cache, N, Sd, R, spi, klen_na, cross_len, True, seg)
        na_tok = log_softmax(vocab(x))
    fork.join([sa_tok, na_tok, bound_na[0], bound_na[1]])
    return bound_sa, sa_tok, bound_na, na_tok


@_scoped_compute_dtype
def criterion_uic_compact(outs, phrase_num, phrase_length, phrase_syn, token_labels, token_weight):
    """criterion_uic for token tensors that hold the real tokens' rows only (HINTS["token_rows"]): ``token_labels`` int64 and
    ``token_weight`` float32 [len(token_rows)] are the labels of those rows and 1 / 0 for real / padding entries.  Same value and
    gradients as criterion_uic on the full tensors (the rows left out carry zero weight there)."""
    sa_len, sa_syn, sa_tok, na_len, na_syn, na_tok = outs
    dev = sa_tok.device
    if phrase_length.dim() == 3:
        phrase_num = phrase_num.reshape(-1)
        phrase_length = phrase_length.reshape(-1, phrase_length.shape[2])
        phrase_syn = phrase_syn.reshape(-1, phrase_syn.shape[2])
    phrase_num, phrase_length, phrase_syn = phrase_num.to(dev).long(), phrase_length.to(dev).long(), phrase_syn.to(dev).long()
    pair = isinstance(token_weight, (tuple, list))             # both branches in one tensor (_forward_paired): (SA weights, NA weights)
    if pair:
        if sa_tok is not na_tok:
            raise hip.BofiHipError("a pair of token weights goes with the paired log-probs of _forward_paired")
        made = getattr(sa_tok, "_bofi_picked", None)              # picked inside the forward (HINTS["pick_labels"]): fused backward
        if made is not None and made[1] is token_labels and sa_len.dim() == 3 and sa_len.shape[1] + 1 <= phrase_length.shape[1]:
            # everything the criterion needs is at hand as small tensors: one launch (and one in the backward); the slot masks and the
            # token count of the general form below are not even made
            loss, parts = UicCriterionFn.apply(sa_len, sa_syn, na_len, na_syn, made[0], phrase_num, phrase_length, phrase_syn,
                                               token_weight[0], token_weight[1])
            return loss.squeeze(0), list(parts.unbind(0))
    slot = torch.arange(phrase_length.shape[1] - 1, device=dev).unsqueeze(0)
    slot_mask = slot < phrase_num.unsqueeze(1)
    len_lab, syn_lab = phrase_length[:, 1:], phrase_syn[:, 1:]
    denom = token_weight[0].sum() if pair else token_weight.sum()

    def nll(lp, lab, mask):
        P_ = lp.shape[1]                                        # [N, Pm, .] from the paired forward, [N, L - 1, .] otherwise
        return (-lp.gather(2, lab[:, :P_].unsqueeze(2)).squeeze(2) * mask[:, :P_]).sum() / denom

    def tok(lp):
        return (-lp.gather(1, token_labels.unsqueeze(1)).squeeze(1) * token_weight).sum() / denom

    if pair:
        if made is not None and made[1] is token_labels:
            picked = -made[0]
        else:
            picked = -sa_tok.gather(1, token_labels.unsqueeze(1)).squeeze(1)      # one gather (one scatter in backward) for both branches
        tok_sa, tok_na = (picked * token_weight[0]).sum() / denom, (picked * token_weight[1]).sum() / denom
    else:
        tok_sa, tok_na = tok(sa_tok), tok(na_tok)
    parts = [nll(sa_len, len_lab, slot_mask), tok_sa, nll(sa_syn, syn_lab, slot_mask),
             nll(na_len, len_lab, slot_mask), tok_na, nll(na_syn, syn_lab, slot_mask)]
    return sum(parts), parts


def rl_prepare(cfg, saic=None, naic=None, *, sample_n: int = 1, strict_q1: bool = True, device=None):
    """Host half of ``sampled_logprobs``: the sampled captions' slot layouts -> the index tensors the decoder passes read
    (the loader's collate on the samples, boficap_amd.collate.phrase_collate).  Returns a dict of device tensors:
    ``sa_seq`` / ``sa_syn`` int64 [N, S], ``sa_klen`` int32 [N, S] (decode_SA inputs and key counts), ``na_syn`` int64 [N, S],
    ``na_klen`` int32 [N, S] (decode_NA; quirk Q1's fill mask when ``strict_q1``) -- for the modes given."""
    import numpy as np
    from .collate import phrase_collate
    S = cfg.seq_length
    out = {}

    def layout(r):
        seq = r["seq"].detach().cpu().numpy().astype(np.int64)
        N = seq.shape[0]
        if seq.shape[1] != S or N % sample_n:
            raise hip.BofiHipError(f"sampled captions {seq.shape} for {sample_n} samples per image")
        labels = np.zeros((N, S + 2), np.int64)
        labels[:, 0] = cfg.bos_idx            # core_SAIC starts from seq[:, 0] = BOS (TransformerModel.py:1900), as the loader's labels do (dataloader.py:298)
        labels[:, 1:S + 1] = seq
        plen = r["phrase_length"].detach().cpu().numpy().astype(np.int64)
        psyn = np.where(plen > 0, r["phrase_syn"].detach().cpu().numpy().astype(np.int64), 0)
        return phrase_collate(labels, plen, psyn, pad_idx=cfg.pad_idx, bos_idx=cfg.bos_idx, eos_idx=cfg.eos_idx, len_idx=cfg.len_idx)

    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if saic is not None:
        c = layout(saic)
        out["sa_syn"] = to(c["extend_phrase_syn_seq"][:, 1:-1])
        out["sa_seq"] = to(c["extend_phrase_seq"])
        out["sa_klen"] = to(c["extend_phrase_seq_mask"].sum(-1).astype(np.int32))
    if naic is not None:
        c = layout(naic)
        N = c["phrase_length"].shape[0]
        out["na_syn"] = to(c["extend_phrase_syn_seq"][:, 1:-1])
        last = c["phrase_length"][:, 1:].sum(1) + 1                       # 1 + tokens laid out (TransformerModel.py:1859-1866)
        fill = np.full(N, last[-1] - 1) if strict_q1 else last - 1       # Q1: every row's fill mask uses the LAST row's length (:1872-1873)
        out["na_klen"] = to(np.repeat(fill[:, None], S, 1).astype(np.int32))
    return out


def rl_prepare_naic_device(cfg, phrase_length: torch.Tensor, phrase_syn: torch.Tensor, *, strict_q1: bool = True):
    """``rl_prepare``'s non-autoregressive half on the device: the slot layout [N, S] -> {na_syn int64 [N, S], na_klen int32 [N, S]} (the labels of the collate -- the
    semi-autoregressive kernel's, its token output unused -- and quirk Q1's fill mask: every row takes the LAST row's token count when ``strict_q1``,
    TransformerModel.py:1859-1873)."""
    N, S = phrase_length.shape
    zeros = torch.zeros(N, S, dtype=torch.int64, device=phrase_length.device)
    na_syn = rl_prepare_saic_device(cfg, zeros, phrase_length, phrase_syn)["sa_syn"]
    ntok = phrase_length.to(torch.int32).sum(1, dtype=torch.int32)
    fill = ntok[-1:].expand(N) if strict_q1 else ntok
    return {"na_syn": na_syn, "na_klen": fill[:, None].expand(N, S).contiguous()}


_COLLATE = {"tensor_ops": False}      # tests: True = rl_prepare_saic_device as tensor operations (the kernel's reference)


def rl_prepare_saic_device(cfg, seq: torch.Tensor, phrase_length: torch.Tensor, phrase_syn: torch.Tensor):
    """``rl_prepare``'s semi-autoregressive half on the device (one launch, bofi_saic_collate; no host round trip, no synchronisation: it can sit inside a captured
    graph; the tensor-operation form below is its reference and what a CPU tensor gets): ``seq`` int64 [N, S] sampled tokens, ``phrase_length`` / ``phrase_syn`` [N, S] as the engine exports them -> {sa_syn, sa_seq int64 [N, S],
    sa_klen int32 [N, S]} -- the loader's collate of those captions (captioning/data/dataloader.py:343-428; the index arithmetic of
    boficap_amd.collate.phrase_collate, checked against it in tests/test_gpu_rl.py)."""
    N, S = seq.shape
    if seq.is_cuda and not _COLLATE["tensor_ops"]:              # one launch (train_ops.hip saic_collate_kernel) instead of ~35 tensor operations
        seq_c, pl_c, ps_c = seq.contiguous(), phrase_length.to(torch.int32).contiguous(), phrase_syn.to(torch.int64).contiguous()
        out = {"sa_syn": torch.empty(N, S, dtype=torch.int64, device=seq.device), "sa_seq": torch.empty(N, S, dtype=torch.int64, device=seq.device),
               "sa_klen": torch.empty(N, S, dtype=torch.int32, device=seq.device)}
        hip.check(hip.lib().bofi_saic_collate(hip.ptr(seq_c), hip.ptr(pl_c), hip.ptr(ps_c), N, S, cfg.bos_idx, hip.ptr(out["sa_syn"]), hip.ptr(out["sa_seq"]),
                                              hip.ptr(out["sa_klen"]), hip.stream_ptr()), "bofi_saic_collate")
        return out
    L = S + 2
    dev = seq.device
    plen = phrase_length.long()
    psyn = torch.where(plen > 0, phrase_syn.long(), torch.zeros_like(plen))
    labels = torch.zeros(N, L, dtype=torch.int64, device=dev)
    labels[:, 0] = cfg.bos_idx
    labels[:, 1:S + 1] = seq
    ntok = plen.sum(1)
    ends = plen.cumsum(1)
    t = torch.arange(S, device=dev)[None, :]
    valid = t < ntok[:, None]
    pid = (ends[:, None, :] <= t[:, :, None]).sum(2).clamp(max=S - 1)      # phrase of token position t
    zero = torch.zeros_like(pid)
    sa_syn = torch.where(valid, psyn.gather(1, pid), zero)
    start = ends - plen
    cur = plen.gather(1, pid)
    k = t - start.gather(1, pid)
    first = pid == 0
    pm1 = (pid - 1).clamp(min=0)
    prev = torch.where(first, torch.ones_like(pid), plen.gather(1, pm1))
    prev_start = torch.where(first, zero, 1 + start.gather(1, pm1))
    cur_s, prev_s = cur.clamp(min=1), prev.clamp(min=1)
    times = torch.div(cur_s, prev_s, rounding_mode="floor")
    pre_less = prev_s - cur_s % prev_s
    stretched = torch.where(k < pre_less * times, torch.div(k, times.clamp(min=1), rounding_mode="floor"),
                            pre_less + torch.div(k - pre_less * times, times + 1, rounding_mode="floor"))
    src = torch.where(cur <= prev, prev - cur + k, stretched)
    sa_seq = torch.where(valid, labels.gather(1, (prev_start + src).clamp(0, L - 1)), zero)
    sa_klen = torch.where(valid, ends.gather(1, pid), ntok[:, None].expand(N, S)).to(torch.int32)
    return {"sa_syn": sa_syn.contiguous(), "sa_seq": sa_seq.contiguous(), "sa_klen": sa_klen.contiguous()}


@_scoped_compute_dtype
def sampled_logprobs_prepared(P, cfg, att_feats, att_masks, prep, *, sample_n: int = 1, training: bool = False, seed: Optional[int] = None,
                              compute_dtype: torch.dtype = torch.float32, step_word: Optional[torch.Tensor] = None, reuse: Optional[dict] = None):
    out = _sampled_logprobs_prepared(P, cfg, att_feats, att_masks, prep, sample_n=sample_n, training=training, seed=seed, compute_dtype=compute_dtype,
                                     step_word=step_word, reuse=reuse)
    if reuse is not None and "shadows" not in reuse:            # the bf16 operands the cross K|V were made as (kept by their producers in the step cache, which
        reuse["shadows"] = [(t, _shadow(t), t.data_ptr() in _SHADOW_ONLY) for t in reuse["kv_cache"].values()]      # every call clears): registered again by the next calls
    return out


def _sampled_logprobs_prepared(P, cfg, att_feats, att_masks, prep, *, sample_n: int = 1, training: bool = False, seed: Optional[int] = None,
                               compute_dtype: torch.dtype = torch.float32, step_word: Optional[torch.Tensor] = None, reuse: Optional[dict] = None):
    """Device half of ``sampled_logprobs``: tensors in, log-probs out, no host work (``att_masks`` None) -- what a captured
    self-critical step replays.  ``prep``: rl_prepare's dict.  ``reuse`` (tape-free callers only: a dict kept across calls on the SAME inputs, weights and
    seed): the encoder's memory and the decoder layers' cross K|V do not depend on the captions -- under the counter-based dropout masks they are the same
    tensors in every call -- so the per-phrase forwards of the reference-estimator step compute them once."""
    dev = att_feats.device
    S, d = cfg.seq_length, cfg.d_model
    if compute_dtype not in (torch.float32, torch.bfloat16):
        raise hip.BofiHipError(f"training compute dtype {compute_dtype}: float32 or bfloat16")
    _COMPUTE["dtype"] = compute_dtype
    _STEP_CACHE.clear()
    _SHADOW_ONLY.clear()
    att_len = None
    if att_masks is not None:
        max_len = int(att_masks.long().sum(1).max())
        att_feats, att_masks = att_feats[:, :max_len].contiguous(), att_masks[:, :max_len]
        att_len = att_masks.long().sum(1).to(torch.int32).contiguous()
    att_feats = _need(att_feats.float() if att_feats.dtype != torch.float32 else att_feats, "att_feats")
    B, R, _ = att_feats.shape
    N = B * sample_n
    drop = _Drop(cfg.dropout, cfg.drop_prob_lm, seed if training else None, step_word)
    if reuse is not None and torch.is_grad_enabled():
        raise hip.BofiHipError("reuse: tape-free calls only (the gradient pass needs its own graph through the encoder)")
    if reuse is not None and "memory" in reuse:
        memory, kv_cache = reuse["memory"], reuse["kv_cache"]
        drop.k = reuse["drop_sites"]                          # (the decoder's dropout sites keep the numbers they have behind the encoder's: the same masks)
        for t, sh, only in reuse.get("shadows", ()):
            if sh is not None:
                _register_shadow(t, sh, only)
    else:
        memory = encode_memory(P, cfg, att_feats, att_len, drop)
        kv_cache = {}
        if reuse is not None:
            reuse["memory"], reuse["kv_cache"], reuse["drop_sites"] = memory, kv_cache, drop.k
    att_len_cap = None if att_len is None else att_len.repeat_interleave(sample_n).contiguous()
    tname, sname, pe = "model.tgt_embed.lut.weight", "model.syn_embed.lut.weight", P["model.pos_embed.pe"]

    def emb(tok, syn):
        x = embed(P[tname], P[sname], pe, tok, syn, S, P.g(tname), P.g(sname))
        return drop(x) if drop.on and drop.p > 0.0 else x

    def tokens(x):
        return log_softmax(P.lin(x, "model.generator.proj")).view(N, S, -1)

    def check(t, dtype):
        if t.shape != (N, S) or t.dtype != dtype:
            raise hip.BofiHipError(f"prepared tensor {tuple(t.shape)} {t.dtype}: {(N, S)} {dtype} expected ({B} images x {sample_n} samples)")
        return t.contiguous()

    out = [None, None]
    if "sa_seq" in prep and "na_syn" in prep:
        # both modes: one decoder pass over 2N captions (same layers, same weights; every launch is latency-bound at these sizes,
        # see _forward_paired).  Image-major order -- image i's SAIC samples, then its NAIC samples -- keeps the 2n captions that
        # share an image's keys adjacent (kdiv = 2n).
        with torch.no_grad():
            pn = torch.arange(2 * N, device=dev)
            j = pn % (2 * sample_n)
            cap = (pn // (2 * sample_n)) * sample_n + j % sample_n
            is_na = (j >= sample_n).unsqueeze(1)
            n_all = torch.arange(N, device=dev)
            at_sa = (n_all // sample_n) * (2 * sample_n) + n_all % sample_n
            tok2 = torch.where(is_na, torch.full_like(prep["sa_seq"][cap], cfg.bos_idx), check(prep["sa_seq"], torch.int64)[cap]).contiguous()
            syn2 = torch.where(is_na, check(prep["na_syn"], torch.int64)[cap], check(prep["sa_syn"], torch.int64)[cap]).contiguous()
            klen2 = torch.where(is_na, check(prep["na_klen"], torch.int32)[cap], check(prep["sa_klen"], torch.int32)[cap]).contiguous()
            len2 = None if att_len_cap is None else att_len_cap[cap].contiguous()
        x = embed(P[tname], P[sname], pe, tok2, syn2, S, P.g(tname), P.g(sname))
        x = drop(x) if drop.on and drop.p > 0.0 else x
        x = decode_rows(P, cfg, drop, x, memory, kv_cache, 2 * N, S, R, 2 * sample_n, klen2, len2)
        lp = log_softmax(P.lin(x, "model.generator.proj")).view(2 * N, S, -1)
        return lp.index_select(0, at_sa), lp.index_select(0, at_sa + sample_n)
    if "sa_seq" in prep:
        out[0] = tokens(decode_rows(P, cfg, drop, emb(check(prep["sa_seq"], torch.int64), check(prep["sa_syn"], torch.int64)), memory, kv_cache,
                                    N, S, R, sample_n, check(prep["sa_klen"], torch.int32), att_len_cap))
    if "na_syn" in prep:
        bos = torch.full((N, S), cfg.bos_idx, dtype=torch.int64, device=dev)
        out[1] = tokens(decode_rows(P, cfg, drop, emb(bos, check(prep["na_syn"], torch.int64)), memory, kv_cache, N, S, R, sample_n,
                                    check(prep["na_klen"], torch.int32), att_len_cap))
    return out[0], out[1]


def sampled_logprobs(P, cfg, att_feats, att_masks, saic=None, naic=None, *, sample_n: int = 1, strict_q1: bool = True, training: bool = False,
                     seed: Optional[int] = None, compute_dtype: torch.dtype = torch.float32, step_word: Optional[torch.Tensor] = None):
    """Token log-probs of SAMPLED captions with the autograd tape: the differentiable half of the self-critical step.

    The reference samples with gradients enabled (loss_wrapper.py:193-209: ``model(..., mode='sample')`` in SAIC and in NAIC
    mode, then ``struc_crit(seq_logprobs, seq, gts)``).  Here the engine samples without a tape and this function recomputes
    the log-probs of what was sampled: given a sampled slot layout the distribution of every position is exactly the
    teacher-forced one -- ``decode_SA`` on the sampled caption (inputs of a phrase = the previous phrase squeezed / stretched,
    block mask; TransformerModel.py:1933-1952 does this per iteration) resp. ``decode_NA`` on the layout's syntactic labels
    (:1870-1875, with quirk Q1's fill mask when ``strict_q1``).  Slot layouts are discrete: no gradient reaches the bound layer,
    as in the reference.

    ``saic`` / ``naic``: dicts with ``seq`` [N, S] int64, ``phrase_length`` [N, S] and ``phrase_syn`` [N, S] as returned by
    ``_sample`` (N = images x sample_n, image b's copies in rows b*n..b*n+n-1).  Returns (saic_logprobs, naic_logprobs), each
    [N, S, V] float32 or None.  One encoder pass serves both.  = rl_prepare (host) + sampled_logprobs_prepared (device)."""
    prep = rl_prepare(cfg, saic, naic, sample_n=sample_n, strict_q1=strict_q1, device=att_feats.device)
    return sampled_logprobs_prepared(P, cfg, att_feats, att_masks, prep, sample_n=sample_n, training=training, seed=seed,
                                     compute_dtype=compute_dtype, step_word=step_word)


def new_self_critical(logprobs, seq, scores, sample_n: int):
    """StructureLosses 'new_self_critical' (captioning/modules/losses.py:37-51, 157-176), reduction 'mean': the reward of a
    sample is its score minus the mean score of the image's OTHER samples; loss = -sum(logp(token) * mask * reward) / sum(mask),
    mask = (seq > 0) shifted right by one with a leading 1.  ``scores``: [N] (any float tensor / array) from the external
    caption scorer.  Returns (loss, rewards [B, n])."""
    seq = seq.to(logprobs.device).long()
    mask = (seq > 0).to(logprobs.dtype)
    mask = torch.cat([mask.new_ones(mask.size(0), 1), mask[:, :-1]], 1)
    sc = torch.as_tensor(scores, dtype=logprobs.dtype, device=logprobs.device).view(-1, sample_n)
    if sample_n < 2:
        raise ValueError("new_self_critical needs at least two samples per image")
    reward = sc - (sc.sum(1, keepdim=True) - sc) / (sample_n - 1)
    picked = logprobs.gather(2, seq.unsqueeze(2)).squeeze(2)
    loss = (-picked * mask * reward.view(-1, 1)).sum() / mask.sum()
    return loss, sc


STRUCTURE_LOSS_TYPES = ("seqnll", "risk", "max_margin", "multi_margin", "softmax_margin", "real_softmax_margin", "new_self_critical")


def structure_loss(loss_type: str, logprobs, seq, scores, sample_n: int, reduction: str = "mean", entropy_reward_weight: float = 0.0):
    """StructureLosses.forward for every ``structure_loss_type`` of the reference (captioning/modules/losses.py:38-179): the sequence-level
    losses of Edunov et al. over the ``sample_n`` sampled captions of an image -- 'seqnll' (:72-79), 'risk' (:81-87), 'max_margin' (:96-106),
    'multi_margin' (:118-128), 'softmax_margin' (:136-144), 'real_softmax_margin' (:146-155), 'new_self_critical' (:157-176) -- with the
    entropy reward (:53-57) and reduction 'none' where the reference has it.  ``logprobs`` [N, S, V] (what the type takes: log-softmax or
    logits), ``scores`` [N] from the caption scorer.  The reference AS SHIPPED raises a NameError for every type but 'new_self_critical'
    (its losses.py never imports ``F``); the formulas are pinned by tests/golden/tiny_structure_losses, recorded from the reference's code
    with that import supplied.  Index bookkeeping on [N, S] tensors; returns (loss, reward [B, n] = the raw scores)."""
    if loss_type not in STRUCTURE_LOSS_TYPES:
        raise ValueError(f"structure_loss_type {loss_type!r}: one of {STRUCTURE_LOSS_TYPES}")
    if reduction not in ("mean", "none") or (reduction == "none" and loss_type in ("risk", "max_margin", "multi_margin")):
        raise ValueError(f"structure_loss_type {loss_type!r} has no reduction {reduction!r} (losses.py:87,105,127)")
    if sample_n < 2 and loss_type == "new_self_critical":
        raise ValueError("new_self_critical needs at least two samples per image")
    Fn = torch.nn.functional
    seq = seq.to(logprobs.device).long()
    mask = (seq > 0).to(logprobs.dtype)
    mask = torch.cat([mask.new_ones(mask.size(0), 1), mask[:, :-1]], 1)
    sc = torch.as_tensor(scores, dtype=logprobs.dtype, device=logprobs.device).view(-1, sample_n)
    reward = sc
    if entropy_reward_weight > 0:
        with torch.no_grad():
            entropy = -(Fn.softmax(logprobs, dim=2) * Fn.log_softmax(logprobs, dim=2)).sum(2)
            entropy = (entropy * mask).sum(1) / mask.sum(1)
        sc = sc + entropy_reward_weight * entropy.view(-1, sample_n)
    costs = -sc
    if loss_type in ("risk", "softmax_margin"):
        costs = costs - costs.min(1, keepdim=True)[0]
        costs = costs / costs.max(1, keepdim=True)[0]
    picked = logprobs.gather(2, seq.unsqueeze(2)).squeeze(2)
    if loss_type == "new_self_critical":
        adv = sc - (sc.sum(1, keepdim=True) - sc) / (sample_n - 1)
        out = -picked * mask * adv.view(-1, 1)
        return (out.sum(1) / mask.sum(1) if reduction == "none" else out.sum() / mask.sum()), reward
    picked = picked * mask
    if loss_type == "risk":
        cap = picked.sum(1).view(-1, sample_n)
        return (Fn.softmax(cap.exp(), dim=1) * costs).sum(1).mean(), reward
    cap = (picked.sum(1) / mask.sum(1)).view(-1, sample_n)
    if loss_type == "seqnll":
        return Fn.cross_entropy(cap, costs.min(1)[1], reduction=reduction), reward
    if loss_type in ("max_margin", "multi_margin"):
        best, idx = costs.min(1, keepdim=True)
        hinge = Fn.relu(costs - best - cap.gather(1, idx) + cap)
        return ((hinge.max(1)[0] / 2).mean() if loss_type == "max_margin" else hinge.mean()), reward
    return Fn.cross_entropy(cap + costs, costs.min(1)[1], reduction=reduction), reward


def rl_kl_term(naic_logprobs, saic_logprobs, saic_seq):
    """The KL term of LossWrapper's UIC struc_flag branch under ``rl_kl`` (captioning/modules/loss_wrapper.py:216-222):
    sum over the SAIC caption's tokens (seq > 0) of KL(exp(SAIC log-probs), detached || NAIC) / (count + 1e-6), with
    nn.KLDivLoss(reduction='none') = target * (log target - input) and 0 where target == 0."""
    mask = (saic_seq.to(naic_logprobs.device) > 0).unsqueeze(2)
    target = torch.exp(saic_logprobs).detach()
    kl = torch.where(target > 0, target * (torch.log(target) - naic_logprobs), torch.zeros_like(target))
    return torch.sum(kl * mask) / (torch.sum(mask) + 1e-6)


def criterion_uic(outs, phrase_num, phrase_length, phrase_syn, labels, reduction: str = "mean", self_dis: bool = False):
    """LanguageModelCriterion_UIC.forward (captioning/modules/losses.py:319-369): six masked NLL sums.

    reduction 'mean' (:357-365): each sum divided by the number of real caption tokens of the batch; returns (loss, [6 parts]).
    With ``self_dis`` (:336-339, 366-368; configs uic_sd*, opts.py:58) the self-distillation term is added:
    KLDivLoss(NA token log-probs, exp(SA token log-probs).detach()) over the real token positions, summed over the vocabulary
    and divided by the same token count -- the gradient reaches the NA branch only.
    reduction 'none' (:357-361, drop_worst: tools/train.py:216-220): one value per caption, the caption's six sums divided by ITS
    token count; returns (loss [N], None) as the reference does (its parts are None there, the KL term is not part of it).
    Index bookkeeping only (gathers and masks on tensors of N x S elements) next to the dense KL when asked for."""
    sa_len, sa_syn, sa_tok, na_len, na_syn, na_tok = outs
    dev = sa_tok.device
    if reduction not in ("mean", "none"):
        raise hip.BofiHipError(f"LanguageModelCriterion_UIC: reduction {reduction!r} (the reference has 'mean' and 'none')")
    if sa_tok.dim() != 3:
        raise hip.BofiHipError("criterion_uic takes the dense [N, S, V] token log-probs (criterion_uic_compact has the row-list form)")
    if phrase_length.dim() == 3:
        phrase_num = phrase_num.reshape(-1)
        phrase_length = phrase_length.reshape(-1, phrase_length.shape[2])
        phrase_syn = phrase_syn.reshape(-1, phrase_syn.shape[2])
        labels = labels.reshape(-1, labels.shape[2])
    phrase_num, phrase_length = phrase_num.to(dev).long(), phrase_length.to(dev).long()
    phrase_syn, labels = phrase_syn.to(dev).long(), labels.to(dev).long()
    real = labels[:, 1:-1][:, :sa_tok.shape[1]]                # the token tensors may cover max_tokens positions only (HINTS)
    pos = torch.arange(real.shape[1], device=dev).unsqueeze(0)
    tok_mask = pos < (phrase_length.sum(1, keepdim=True) - 1)
    slot = torch.arange(phrase_length.shape[1] - 1, device=dev).unsqueeze(0)
    slot_mask = slot < phrase_num.unsqueeze(1)
    len_lab, syn_lab = phrase_length[:, 1:], phrase_syn[:, 1:]

    def rows(lp, lab, mask):                                   # masked NLL per caption
        return (-lp.gather(2, lab.unsqueeze(2)).squeeze(2) * mask).sum(1)

    per_cap = [rows(sa_len, len_lab, slot_mask), rows(sa_tok, real, tok_mask), rows(sa_syn, syn_lab, slot_mask),
               rows(na_len, len_lab, slot_mask), rows(na_tok, real, tok_mask), rows(na_syn, syn_lab, slot_mask)]
    if reduction == "none":
        return sum(per_cap) / tok_mask.sum(1), None
    denom = tok_mask.sum()
    parts = [p.sum() / denom for p in per_cap]
    loss = sum(parts)
    if self_dis:
        target_logp = sa_tok.detach()
        kl = torch.where(target_logp > -float("inf"), target_logp.exp() * (target_logp - na_tok), torch.zeros_like(na_tok))     # xlogy: 0 where the target is 0
        loss = loss + (kl.sum(2) * tok_mask).sum() / denom
    return loss, parts
