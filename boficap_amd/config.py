"""Hyper-parameters of the bound+fill captioner, read from the reference's ``opt`` namespace.

The reference reads its hyper-parameters with ``getattr(opt, name, default)`` in
``captioning/models/AttModel.py:56-79`` and ``captioning/models/TransformerModel.py:1631-1640``;
``BofiConfig.from_opt`` reads the same attribute names with the same defaults so that an ``opt``
built for the reference constructs the same network here.
"""
from __future__ import annotations

import dataclasses
from argparse import Namespace

# class counts of the bound heads (reference TransformerModel.py:329-332)
LENGTH_DIM = 20
SYN_DIM = 10
SYN_LOWER = 4
SYN_UPPER = 6


@dataclasses.dataclass(frozen=True)
class BofiConfig:
    vocab_size: int = 9487          # tgt_vocab = vocab_size + 4 (AttModel.py:79)
    att_feat_size: int = 2048
    d_model: int = 512
    d_ff: int = 2048
    h: int = 8
    N_enc: int = 6
    N_dec: int = 6
    N_len: int = 1
    seq_length: int = 20
    dropout: float = 0.1
    drop_prob_lm: float = 0.5
    pad_idx: int = 0
    bos_idx: int = 1
    eos_idx: int = 2
    len_idx: int = 3
    train_mode: str = "UIC"
    decoder_input_mode: str = "add"
    max_pe: int = 5000              # PositionalEncoding max_len (TransformerModel.py:1491)
    head_hidden: int = 100          # Length/Syntactic_classifier1 width (TransformerModel.py:346-349)

    @property
    def tgt_vocab(self) -> int:
        return self.vocab_size + 4

    @property
    def d_k(self) -> int:
        return self.d_model // self.h

    @property
    def bound_len(self) -> int:     # [LEN] + seq_length + [EOS] (TransformerModel.py:1824-1829)
        return self.seq_length + 2

    @staticmethod
    def from_opt(opt: Namespace) -> "BofiConfig":
        g = lambda name, default: getattr(opt, name, default)
        seq_length = g("max_length", 20) or opt.seq_length            # AttModel.py:62
        cfg = BofiConfig(
            vocab_size=opt.vocab_size,
            att_feat_size=opt.att_feat_size,
            d_model=g("d_model", g("input_encoding_size", 512)),       # TransformerModel.py:1634
            d_ff=g("d_ff", g("rnn_size", 2048)),                       # :1635
            h=g("num_att_heads", 8),                                   # :1636
            N_enc=g("N_enc", g("num_layers", 6)),                      # :1631
            N_dec=g("N_dec", g("num_layers", 6)),                      # :1632
            N_len=g("N_len", 0),                                       # :1633
            seq_length=seq_length,
            dropout=g("dropout", 0.1),
            drop_prob_lm=g("drop_prob_lm", 0.5),
            pad_idx=g("pad_idx", 0), bos_idx=g("bos_idx", 1),
            eos_idx=g("eos_idx", 2), len_idx=g("len_idx", 3),
            train_mode=g("train_mode", "AIC"),
            decoder_input_mode=g("decoder_input_mode", "add"),
        )
        cfg.validate()
        return cfg

    def validate(self) -> None:
        if self.train_mode != "UIC":
            raise NotImplementedError(
                f"train_mode={self.train_mode!r}: only the unified bound+fill model (UIC, "
                "TransformerModel.py:1558-1568) is built; the ablation variants are out of scope")
        if self.decoder_input_mode != "add":
            raise NotImplementedError("decoder_input_mode must be 'add' (TransformerModel.py:576-577)")
        if self.N_len < 1:
            raise NotImplementedError("N_len must be >= 1: the UIC model has a bounding network (TransformerModel.py:1558-1568)")
        if self.d_model % self.h:
            raise ValueError("d_model must be a multiple of num_att_heads")

    def to_opt(self, **extra) -> Namespace:
        """An ``opt`` namespace the reference's ``captioning.models.setup`` accepts."""
        ns = Namespace(
            caption_model="transformer", vocab_size=self.vocab_size,
            input_encoding_size=self.d_model, rnn_size=self.d_ff, num_layers=self.N_enc,
            drop_prob_lm=self.drop_prob_lm, fc_feat_size=2048, att_feat_size=self.att_feat_size,
            att_hid_size=512, vocab={str(i): f"w{i}" for i in range(1, self.vocab_size + 1)},
            max_length=self.seq_length, seq_length=self.seq_length, train_mode=self.train_mode,
            N_enc=self.N_enc, N_dec=self.N_dec, N_len=self.N_len, d_model=self.d_model,
            d_ff=self.d_ff, num_att_heads=self.h, dropout=self.dropout,
            decoder_input_mode=self.decoder_input_mode,
            pad_idx=self.pad_idx, bos_idx=self.bos_idx, eos_idx=self.eos_idx, len_idx=self.len_idx,
        )
        for k, v in extra.items():
            setattr(ns, k, v)
        return ns


# The configuration of configs/uic_sd.yml:23-32 (the one BASELINE.json quotes its metric on).
FULL = BofiConfig()
# A small configuration for golden fixtures whose full tensors fit in the repository.
# d_k stays 64 (d_model / h) because the HIP attention kernel is specialised for it, as every
# reference config is (d_model 512, 8 heads).
TINY = BofiConfig(vocab_size=60, att_feat_size=64, d_model=128, d_ff=256, h=2, N_enc=2, N_dec=2,
                  N_len=1, seq_length=20)
# The small configuration with a two-layer bounding network (configs/uic_sd_N2.yml: N_len 2): the decode engine's dense bounding pass.
TINY_N2 = dataclasses.replace(TINY, N_len=2)
