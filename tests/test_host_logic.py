"""CPU-side checks: state_dict schema, opt handling, C-ABI header/library agreement (no GPU)."""
import os
import re

import pytest
import torch

from boficap_amd import weights as W
from boficap_amd.config import FULL, TINY, BofiConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg,key", [(TINY, "schema_TINY"), (FULL, "schema_FULL")])
def test_schema_matches_reference_state_dict(cfg, key, manifest):
    """manifest[schema_*] was recorded from the reference model's state_dict() (oracle/make_golden.py)."""
    ours = [[k, list(s)] for k, s in W.schema(cfg).items()]
    assert ours == manifest[key]
    assert len(W.schema(FULL)) == 311


def test_module_state_dict_is_drop_in(manifest):
    import captioning.models as models            # the repo's drop-in import path
    m = models.setup(TINY.to_opt())
    sd = m.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == manifest["schema_TINY"]
    ref = {k: torch.from_numpy(v) for k, v in W.make_state_dict(TINY, 0).items()}
    res = m.load_state_dict(ref, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m.state_dict()["model.pos_embed.pe"], ref["model.pos_embed.pe"])
    for attr in ("vocab", "seq_length", "ss_prob", "d_model", "train_mode"):
        assert hasattr(m, attr)


def test_opt_defaults_follow_reference():
    from argparse import Namespace
    opt = Namespace(vocab_size=100, input_encoding_size=128, rnn_size=256, num_layers=2, drop_prob_lm=0.5,
                    fc_feat_size=2048, att_feat_size=64, att_hid_size=512, vocab={"1": "a"}, seq_length=20,
                    train_mode="UIC", N_len=1, num_att_heads=2)
    cfg = BofiConfig.from_opt(opt)
    assert (cfg.d_model, cfg.d_ff, cfg.N_enc, cfg.N_dec, cfg.seq_length) == (128, 256, 2, 2, 20)
    assert cfg.tgt_vocab == 104 and (cfg.pad_idx, cfg.bos_idx, cfg.eos_idx, cfg.len_idx) == (0, 1, 2, 3)
    opt.train_mode = "SAIC"
    with pytest.raises(NotImplementedError):
        BofiConfig.from_opt(opt)


def test_weights_regenerate_bit_identically(manifest):
    assert W.digest(W.make_state_dict(TINY, 0, gen_scale=6.0)) == manifest["tiny_mix"]["digest"]


def test_decode_without_device_fails_loudly():
    """No CPU fallback: without a HIP device the model must raise, never compute on the host."""
    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    import captioning.models as models
    from boficap_amd.hip import BofiHipError
    m = models.setup(TINY.to_opt())
    with pytest.raises(BofiHipError):
        m(torch.zeros(2, 0), torch.zeros(2, 36, 64), None, opt={"train_mode": "NAIC"}, mode="sample")


def test_cabi_library_exports_every_declared_symbol():
    """include/boficap_hip.h, the ctypes table and the built .so must agree (load only, no compute)."""
    from boficap_amd import hip
    header = open(os.path.join(ROOT, "include", "boficap_hip.h")).read()
    declared = set(re.findall(r"\b(bofi_[a-z0-9_]+)\s*\(", header))
    declared -= {"bofi_engine", "bofi_config"}
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    if not os.path.exists(hip.LIB_PATH):
        from boficap_amd.build import build
        build(verbose=False)
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bofi_abi_version() == hip.ABI_VERSION
