"""ctypes binding of libboficap_hip.so (C ABI: include/boficap_hip.h).

The HIP library is the product path.  There is no CPU fallback: importing this module without the
built library raises, and every non-zero status from the library raises ``BofiHipError``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOFI_LIB_PATH") or os.path.join(HERE, "libboficap_hip.so")     # (the override is a developer knob for A/B runs)

DT_F32, DT_BF16 = 0, 1
FLAG_STRICT_Q1, FLAG_RAW_LOGITS, FLAG_GRAPH = 1, 2, 4
FLAG_REFINE_SHIFT = 8
FLAG_SAMPLE = 16
FLAG_PHASE_ENCODE, FLAG_PHASE_BOUND, FLAG_PHASE_FILL = 32, 64, 128
FLAG_SAIC_LAYOUT_ONLY = 4096
ABI_VERSION = 4


class BofiHipError(RuntimeError):
    pass


class BofiConfigC(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "vocab", "feat", "d_model", "d_ff", "heads", "n_enc", "n_dec", "seq_length",
        "pad_idx", "bos_idx", "eos_idx", "len_idx", "head_hidden", "max_batch", "max_regions", "dtype", "n_len")]


_P, _I, _I64 = C.c_void_p, C.c_int, C.c_int64

# name -> (restype, argtypes); mirrors include/boficap_hip.h one to one
SIGNATURES = {
    "bofi_abi_version": (_I, []),
    "bofi_last_error": (C.c_char_p, []),
    "bofi_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "bofi_linear": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "bofi_attention": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "bofi_vocab_finalize": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "bofi_attention_ex": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, C.c_float, C.c_uint64, _P, _P, _P, _I, _I, _P]),
    "bofi_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "bofi_linear_rows": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "bofi_rowgemm": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "bofi_bound_qattn": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P]),
    "bofi_layernorm_bwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, C.c_float, C.c_uint64, _P, _P]),
    "bofi_attention_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P]),
    "bofi_attention_bwd_mfma": (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, C.c_float, C.c_uint64,
                                     _P, _P, _P, _I, _I, _P]),
    "bofi_logsoftmax_bwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "bofi_rl_take_draws": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "bofi_saic_collate": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "bofi_nll_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bofi_uic_criterion": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    "bofi_uic_criterion_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bofi_colsum_add": (_I, [_P, _P, _I, _I, _P]),
    "bofi_embed_rows": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "bofi_embed_bwd": (_I, [_P, _P, _P, _I, _I, C.c_float, _P]),
    "bofi_transpose_pad": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "bofi_transpose_many": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "bofi_gemm_tn_acc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P]),
    "bofi_gemm_tn_grouped": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bofi_cast_bf16": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, C.c_float, C.c_uint64, _P, _P]),
    "bofi_reload_env": (None, []),
    "bofi_linear_fused": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "bofi_linear_ex": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, C.c_float, C.c_uint64, _P, _P, _I, _P]),
    "bofi_linear_masked": (_I, [_P, _I, _I, _P, _I, _P, _I, C.c_float, _P, _I, _I, _I, _I, _I, _P]),
    "bofi_dropout": (_I, [_P, _P, _P, _I64, C.c_float, C.c_uint64, _P, _P]),
    "bofi_adam_step": (_I, [_P, _P, _P, _P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_float, _I, C.c_float, C.c_float, _P]),
    "bofi_relu_bwd": (_I, [_P, _P, _P, _I64, _P]),
    "bofi_vocab_stats": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "bofi_vocab_sample": (_I, [_P, _I, _I, _I, _I, C.c_float, C.c_uint64, _P, _I, _P, _P]),
    "bofi_engine_logprob": (_P, [_P]),
    "bofi_engine_refresh_device": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_I64), _P]),
    "bofi_engine_set_q1_group": (_I, [_P, _I]),
    "bofi_engine_set_decodes_in_flight": (_I, [_P, _I]),
    "bofi_engine_bound_loop_active": (_I, [_P, _I]),
    "bofi_engine_saic_put_words": (_I, [_P, _P, _I, _P]),
    "bofi_engine_set_sampling": (_I, [_P, C.c_float, C.c_uint64]),
    "bofi_engine_set_saic_range": (_I, [_P, _I, _I]),
    "bofi_engine_set_bound_iter_cap": (_I, [_P, _I]),
    "bofi_engine_set_live_iterations_max": (_I, [_P, _P]),
    "bofi_engine_set_saturation_out": (_I, [_P, _P]),
    "bofi_engine_set_bound_loop": (_I, [_P, _I]),
    "bofi_engine_create": (_I, [C.POINTER(BofiConfigC), C.POINTER(_P)]),
    "bofi_engine_destroy": (None, [_P]),
    "bofi_engine_fork": (_I, [_P, C.POINTER(_P)]),
    "bofi_engine_fork_sized": (_I, [_P, _I, C.POINTER(_P)]),
    "bofi_engine_stream": (_P, [_P]),
    "bofi_engine_set_weight": (_I, [_P, C.c_char_p, _P, _I64]),
    "bofi_engine_finalize": (_I, [_P]),
    "bofi_engine_decode_naic": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bofi_engine_decode_saic": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "bofi_engine_encode": (_I, [_P, _P, _I, _P, _I, _I, _P, _P]),
    "bofi_engine_bound_step": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "bofi_engine_debug_copy": (_I, [_P, C.c_char_p, _P, _I64, _P]),
    "bofi_gemm_flops": (C.c_double, [_I, _P]),
    "bofi_pack_frag": (_I, [_P, _P, _I, _I, _P]),
    "bofi_attn_block": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P]),
    "bofi_linear_block": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bofi_engine_set_row_stats_out": (_I, [_P, _P, _P]),
    "bofi_attn_linear_block": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "bofi_ffn_block": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "bofi_ffn_linear_block": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    "bofi_engine_fill_naic": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _P]),
    "bofi_attn_out_ffn_block": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python -m boficap_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BofiHipError(
                f"{LIB_PATH} is missing: build it with `python -m boficap_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the decode path.")
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be loaded BEFORE this
        # library so that both resolve to ONE HIP runtime; loaded the other way round the process
        # ends up with two runtimes and the second sees no device.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        with open("/proc/self/maps") as f:
            copies = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
        if len(copies) > 1:
            raise BofiHipError(f"two HIP runtimes are loaded ({sorted(copies)}); import torch before boficap_amd.hip")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if l.bofi_abi_version() != ABI_VERSION:
            raise BofiHipError(f"ABI version {l.bofi_abi_version()} != binding version {ABI_VERSION}")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().bofi_last_error()
        raise BofiHipError(f"{what or 'libboficap_hip'}: status {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def dtype_code(t) -> int:
    import torch
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise BofiHipError(f"unsupported dtype {t.dtype}")


_torch = None


def stream_ptr():
    """The current torch HIP stream as a raw pointer (hot: called once per kernel launch)."""
    global _torch
    if _torch is None:
        import torch
        _torch = torch
    return _torch._C._cuda_getCurrentRawStream(_torch.cuda.current_device())


def gemm_flops(reset: bool = False):
    """(GEMM FLOPs enqueued since the last reset, the part of them in early-out launches) -- host-side tally of the library."""
    sk = C.c_double(0.0)
    v = lib().bofi_gemm_flops(1 if reset else 0, C.cast(C.byref(sk), C.c_void_p))
    return float(v), float(sk.value)
