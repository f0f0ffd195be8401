"""State-dict schema of the reference's UIC TransformerModel and a deterministic initialiser.

The 311-entry schema (names, shapes) is the drop-in contract for ``model.pth`` (SURVEY.md §8b;
reference ``captioning/models/TransformerModel.py:1511-1666``).  ``make_state_dict`` fills it from
numpy's PCG64 stream so the same weights can be regenerated bit-for-bit on any host (the GPU box
has no copy of the reference and no checkpoint).
"""
from __future__ import annotations

import hashlib
import math
import os
from collections import OrderedDict

import numpy as np

from .config import BofiConfig, LENGTH_DIM, SYN_DIM


def _attn(prefix: str, d: int):
    out = []
    for i in range(4):                      # 0=Q 1=K 2=V 3=O (TransformerModel.py:1454-1456,1467)
        out.append((f"{prefix}.linears.{i}.weight", (d, d)))
        out.append((f"{prefix}.linears.{i}.bias", (d,)))
    return out


def _ff(prefix: str, d: int, dff: int):
    return [(f"{prefix}.w_1.weight", (dff, d)), (f"{prefix}.w_1.bias", (dff,)),
            (f"{prefix}.w_2.weight", (d, dff)), (f"{prefix}.w_2.bias", (d,))]


def _norm(prefix: str, d: int):
    return [(f"{prefix}.a_2", (d,)), (f"{prefix}.b_2", (d,))]


def schema(cfg: BofiConfig) -> "OrderedDict[str, tuple]":
    """Names and shapes in the order ``nn.Module.state_dict()`` yields them for the reference."""
    d, dff, V = cfg.d_model, cfg.d_ff, cfg.tgt_vocab
    s = [("att_embed.0.weight", (d, cfg.att_feat_size)), ("att_embed.0.bias", (d,))]
    for l in range(cfg.N_enc):
        p = f"model.encoder.layers.{l}"
        s += _attn(f"{p}.self_attn", d) + _ff(f"{p}.feed_forward", d, dff)
        s += _norm(f"{p}.sublayer.0.norm", d) + _norm(f"{p}.sublayer.1.norm", d)
    s += _norm("model.encoder.norm", d)
    for l in range(cfg.N_dec):
        p = f"model.decoder.layers.{l}"
        s += _attn(f"{p}.self_attn", d) + _attn(f"{p}.src_attn", d) + _ff(f"{p}.feed_forward", d, dff)
        for k in range(3):
            s += _norm(f"{p}.sublayer.{k}.norm", d)
    s += _norm("model.decoder.norm", d)
    s += [("model.syn_embed.lut.weight", (SYN_DIM, d)),
          ("model.tgt_embed.lut.weight", (V, d)),
          ("model.pos_embed.pe", (1, cfg.max_pe, d)),
          ("model.generator.proj.weight", (V, d)), ("model.generator.proj.bias", (V,))]
    lp = "model.length_predictor"
    s += _attn(f"{lp}.length_attn", d) + _ff(f"{lp}.ff", d, dff)        # dead weights, kept for schema
    s += _norm(f"{lp}.norm", d)
    hh = cfg.head_hidden
    s += [(f"{lp}.Length_classifier1.weight", (hh, d)), (f"{lp}.Length_classifier1.bias", (hh,)),
          (f"{lp}.Length_classifier2.weight", (LENGTH_DIM, hh)), (f"{lp}.Length_classifier2.bias", (LENGTH_DIM,)),
          (f"{lp}.Syntactic_classifier1.weight", (hh, d)), (f"{lp}.Syntactic_classifier1.bias", (hh,)),
          (f"{lp}.Syntactic_classifier2.weight", (SYN_DIM, hh)), (f"{lp}.Syntactic_classifier2.bias", (SYN_DIM,))]
    for l in range(cfg.N_len):
        p = f"{lp}.LengthPredictor.{l}"
        s += _attn(f"{p}.self_attn", d) + _attn(f"{p}.src_attn", d) + _ff(f"{p}.ff", d, dff)
        for k in range(3):
            s += _norm(f"{p}.sublayer.{k}.norm", d)
    return OrderedDict(s)


def positional_table(max_len: int, d: int) -> np.ndarray:
    """sin/cos table of PositionalEncoding (TransformerModel.py:1496-1502), computed in float32
    with torch so that it is bit-identical to the registered buffer of the reference."""
    import torch
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy()


def make_state_dict(cfg: BofiConfig, seed: int = 0, *, bound_preset: bool = True,
                    preset_name: str | None = None, gen_scale: float = 1.0,
                    randomize_norms: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 weights.

    Matrices: Xavier-uniform (TransformerModel.py:1621-1623).  Biases: U(+-1/sqrt(fan_in)) as
    nn.Linear leaves them.  LayerNorm gains/offsets are perturbed away from (1, 0) when
    ``randomize_norms`` so that a swapped a_2/b_2 cannot pass parity.  ``bound_preset`` replaces the
    two output layers of the bound heads by the calibrated ones in
    ``presets/bound_heads_<preset_name>_seed<seed>.npz`` (written by oracle/calibrate_preset.py) so
    that phrase slots are actually produced -- random weights emit EOS at once, SURVEY.md §8c;
    ``preset_name`` defaults to "full"/"tiny" by d_model (+ "_n<N_len>" for a multi-layer bounding network).  ``gen_scale`` scales
    generator.proj.weight to widen the top-2 logit gap of the greedy argmax.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in schema(cfg).items():
        if name == "model.pos_embed.pe":
            sd[name] = positional_table(cfg.max_pe, cfg.d_model)
        elif name.endswith(".a_2"):
            sd[name] = (rng.uniform(0.6, 1.4, shape) if randomize_norms else np.ones(shape)).astype(np.float32)
        elif name.endswith(".b_2"):
            sd[name] = (rng.uniform(-0.2, 0.2, shape) if randomize_norms else np.zeros(shape)).astype(np.float32)
        elif len(shape) == 2:
            fan_out, fan_in = shape
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            sd[name] = rng.uniform(-bound, bound, shape).astype(np.float32)
        else:                                                   # Linear bias: fan_in of its weight
            w = sd[name[:-len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(w.shape[1])
            sd[name] = rng.uniform(-bound, bound, shape).astype(np.float32)
    if gen_scale != 1.0:
        sd["model.generator.proj.weight"] = (sd["model.generator.proj.weight"] * np.float32(gen_scale)).astype(np.float32)
    if bound_preset:
        name = preset_name or (("full" if cfg.d_model == 512 else "tiny") + ("" if cfg.N_len == 1 else f"_n{cfg.N_len}"))
        if name == "full" and os.environ.get("BOFI_PRESET_FULL"):      # developer knob for A/B runs: "full_r2" = round 2's bimodal preset
            name = os.environ["BOFI_PRESET_FULL"]
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "presets",
                            f"bound_heads_{name}_seed{seed}.npz")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run oracle/calibrate_preset.py (or pass bound_preset=False)")
        with np.load(path) as heads:
            for k in heads.files:
                if k == "attn_gain":
                    apply_attn_gain(sd, float(heads[k]))
                    continue
                if heads[k].shape != sd[k].shape:
                    raise ValueError(f"preset {path}: {k} has shape {heads[k].shape}, model wants {sd[k].shape}")
                sd[k] = heads[k].astype(np.float32)
    return sd


def apply_attn_gain(sd, gain: float) -> None:
    """Scale the two attention output projections of the bound layer (part of the synthetic
    preset: makes the [LEN] row depend more strongly on the slots laid out and on the image)."""
    if gain == 1.0:
        return
    p = "model.length_predictor.LengthPredictor.0."
    for k in (p + "self_attn.linears.3.weight", p + "src_attn.linears.3.weight"):
        sd[k] = (sd[k] * np.float32(gain)).astype(np.float32)


def digest(sd) -> str:
    """SHA-256 over names, shapes and float32 bytes (fixtures store it to check regeneration).
    The sin/cos buffer ``pe`` is left out: it is a fixed function, not a drawn weight, and torch's
    float32 sin/cos differ in the last bit between host CPUs (observed: Xeon vs EPYC)."""
    hsh = hashlib.sha256()
    for k, v in sd.items():
        if k == "model.pos_embed.pe":
            continue
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        hsh.update(k.encode()); hsh.update(str(a.shape).encode()); hsh.update(a.tobytes())
    return hsh.hexdigest()


def synthetic_att_feats(B: int, R: int, F: int, seed: int = 1234) -> np.ndarray:
    """|N(0,1)| region features, float32 [B, R, F]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.abs(rng.standard_normal((B, R, F), dtype=np.float32))


def with_len_row_shared(sd, cfg):
    """A copy of ``sd`` whose word table's [LEN] row equals the syntactic table's.  The synthetic bound-head preset is
    calibrated on the NA bound input (syn_embed); with this one row shared, iteration 1 of the semi-autoregressive bound step
    (tgt_embed input, TransformerModel.py:515-518) equals the NA one, so SAIC decodes emit multi-phrase captions instead of
    stopping (or NaN-ing, :1956-1958) at once.  Used by the SAIC / self-critical fixtures and tests."""
    out = dict(sd)
    lut = np.array(sd["model.tgt_embed.lut.weight"], copy=True)
    lut[cfg.len_idx] = sd["model.syn_embed.lut.weight"][cfg.len_idx]
    out["model.tgt_embed.lut.weight"] = lut
    return out
