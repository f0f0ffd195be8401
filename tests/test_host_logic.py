"""CPU-side checks: state_dict schema, opt handling, C-ABI header/library agreement (no GPU)."""
import os

import numpy as np
import re

import pytest
import torch

from boficap_amd import weights as W
from boficap_amd.config import FULL, TINY, BofiConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from boficap_amd.config import TINY_N2


@pytest.mark.parametrize("cfg,key", [(TINY, "schema_TINY"), (FULL, "schema_FULL"), (TINY_N2, "schema_TINY_N2")])
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
    # ... and on the number of parameters of every entry point (the header is what a maintainer binds against)
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, params in re.findall(r"\b(bofi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", code, flags=re.S):
        if name in hip.SIGNATURES:
            n = 0 if params.strip() in ("", "void") else params.count(",") + 1
            assert n == len(hip.SIGNATURES[name][1]), (name, n, len(hip.SIGNATURES[name][1]))
    if not os.path.exists(hip.LIB_PATH):
        from boficap_amd.build import build
        build(verbose=False)
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bofi_abi_version() == hip.ABI_VERSION


def test_vectorised_phrase_collate_equals_the_loops():
    """boficap_amd.collate.phrase_collate against the loop restatement of dataloader.py:343-428, including phrase
    lengths the synthetic sampler never draws (long phrases after short ones: the stretch branch)."""
    from training_batch import collate_loops, make_training_batch
    from boficap_amd.collate import phrase_collate, synthetic_training_batch
    from boficap_amd.config import FULL, TINY
    for cfg in (TINY, FULL):
        a, b = make_training_batch(cfg, 3, 5, 9), synthetic_training_batch(cfg, 3, 5, 9)
        assert a.keys() == b.keys() and all(a[k].shape == b[k].shape and (a[k] == b[k]).all() for k in a)
    rng = np.random.default_rng(0)
    S = FULL.seq_length
    for _ in range(100):
        N = 5
        labels, plen, psyn, pn = np.zeros((N, S + 2), np.int64), np.zeros((N, S), np.int64), np.zeros((N, S), np.int64), np.zeros(N, np.int64)
        for n in range(N):
            lens = rng.integers(1, 8, int(rng.integers(1, 9)))
            while lens.sum() > S:
                lens = lens[:-1]
            P = len(lens)
            pn[n], plen[n, :P], psyn[n, :P] = P, lens, rng.integers(4, 10, P)
            labels[n, 1:1 + lens.sum()] = rng.integers(7, 9000, lens.sum())
        a, b = collate_loops(FULL, labels, pn, plen, psyn), phrase_collate(labels, plen, psyn, len_idx=FULL.len_idx)
        assert all((a[k] == b[k]).all() for k in a)
    with pytest.raises(ValueError):
        phrase_collate(np.zeros((1, S + 2), np.int64), np.array([[0, 2] + [0] * (S - 2)]), np.zeros((1, S), np.int64))


def test_collate_against_the_reference_loader(manifest):
    """tests/golden/tiny_collate.npz holds what the REFERENCE's Dataset.collate_func (captioning/data/dataloader.py:231-452) returned for sampled
    captions (squeeze, exact stretch and stretch with a remainder all present) and ragged / equal region counts: the vectorised phrase collate, the
    label store's batch and the region collate must reproduce it."""
    from boficap_amd.collate import phrase_collate
    from boficap_amd.config import TINY
    from boficap_amd.data import LabelStore, collate_regions
    from conftest import load_golden
    g = load_golden("tiny_collate")
    man = manifest["tiny_collate"]
    assert man["squeeze"] > 0 and man["stretch_exact"] > 0 and man["stretch_with_remainder"] > 0
    n_img, spi, S = man["n_img"], man["seq_per_img"], TINY.seq_length
    n_cap = n_img * spi
    seqs, pn, pl, ps = g["in_seqs"], g["in_phrase_num"], g["in_phrase_length"], g["in_phrase_syn"]
    labels = np.zeros((n_cap, S + 2), np.int64)
    labels[:, 1:S + 1], labels[:, 0], labels[:, S + 1] = seqs, TINY.bos_idx, TINY.eos_idx
    b = phrase_collate(labels, pl, ps, pad_idx=TINY.pad_idx, bos_idx=TINY.bos_idx, eos_idx=TINY.eos_idx, len_idx=TINY.len_idx)
    for tag in ("ragged", "full"):
        for k, v in b.items():
            ref = g[f"{tag}_{k}"].reshape(v.shape)               # (the loader hands the [S, S] masks over flattened, dataloader.py:440)
            assert (v == ref).all(), (tag, k)
        regions = g[f"{tag}_regions"]
        feats, masks = collate_regions([g[f"{tag}_att_feats"][i, :int(r)] for i, r in enumerate(regions)])
        assert feats.shape == g[f"{tag}_att_feats"].shape and (feats == g[f"{tag}_att_feats"]).all()
        if tag == "full":
            assert masks is None and g["full_att_masks"].size == 0
        else:
            assert (masks == g["ragged_att_masks"]).all()
    # the label store on the same captions as a label file (image i owns captions i*spi .. +spi: the run it draws is the whole range)
    store = LabelStore(dict(labels=seqs, label_start_ix=np.arange(n_img) * spi + 1, label_end_ix=(np.arange(n_img) + 1) * spi, phrase_num=pn,
                            phrase_length=pl, phrase_label=ps), pad_idx=TINY.pad_idx, bos_idx=TINY.bos_idx, eos_idx=TINY.eos_idx, len_idx=TINY.len_idx)
    sb = store.batch(list(range(n_img)), spi, np.random.default_rng(0))
    for k in ("labels", "phrase_num", "phrase_length", "phrase_syn", "extend_phrase_syn_seq", "extend_phrase_seq", "extend_phrase_seq_mask"):
        assert (sb[k].reshape(g[f"full_{k}"].shape) == g[f"full_{k}"]).all(), k
    assert (sb["gts"][0] == g["full_gts_first"]).all()


def test_noam_rate_and_bucket_layout():
    from boficap_amd.trainer import FlatBucket, noam_rate
    assert abs(noam_rate(1, 512, 1.0, 20000) - 512 ** -0.5 * 20000 ** -1.5) < 1e-15          # misc.py:179-185
    assert abs(noam_rate(20000, 512, 1.0, 20000) - 512 ** -0.5 * 20000 ** -0.5) < 1e-12
    assert noam_rate(40000, 512, 1.0, 20000) < noam_rate(20000, 512, 1.0, 20000)
    net = torch.nn.Linear(3, 2)
    w0 = net.weight.detach().clone()
    b = FlatBucket(net)
    assert torch.equal(net.weight, w0) and b.numel == 64 + 64 and net.weight.data_ptr() == b.flat.data_ptr()
    b.flat.zero_()
    assert float(net.weight.detach().abs().sum()) == 0.0                                                # the module sees the bucket


def test_shipped_library_has_no_packed_f32_arithmetic(tmp_path):
    """boficap_amd/build.py compiles without the SLP / loop vectorisers: no v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 in any kernel of
    the shipped code objects.  Round 2 saw wrong sums from SLP-packed float32 chains in the bounding tail beside other kernels' MFMAs;
    round 3 could not reproduce it (dev/exp/pk_fma_repro.py: 0 mismatches in 8 000 concurrent steps of the SLP build, docs/history/r03.md 13.7),
    so the cause stays unestablished -- and beside MFMAs the packed forms cost issue slots anyway (MI355X_MICROARCH.md, cycle constants).
    This pins the build so that a flag change cannot bring them back unnoticed."""
    import glob
    import shutil
    import subprocess
    from boficap_amd import hip
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(hip.LIB_PATH) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    lib = shutil.copy(hip.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = glob.glob(str(tmp_path / "lib.so.*amdgcn*gfx950*"))
    assert objs, "no gfx950 code object in the library"
    packed, mfma = 0, 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        packed += sum(asm.count(op) for op in ("v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"))
        mfma += asm.count("v_mfma_f32_16x16x32_bf16")
    assert mfma > 1000, mfma                                   # (the disassembly is the real thing)
    assert packed == 0, packed


def test_persistent_gemm_statistics_registers_are_never_copied_or_spilled(tmp_path):
    """gemm_pers.hip's loader wavefronts read the LayerNorm statistics with inline-assembly loads whose completion only a hand-counted
    s_waitcnt covers (compiler-tracked loads would drain the slab ring): the compiler believes the destination registers defined at the asm
    statement.  That is safe as long as it never MOVES or SPILLS them between the load and the wait -- a copy would read registers whose data has
    not arrived.  Pinned on the shipped code objects: every load site writes the same registers, and inside the loaders' code no move / accumulator
    copy / scratch / LDS store / lane-permute instruction names one of them (the sums and the loads' own address arithmetic are all that touches them)."""
    import glob
    import re
    import shutil
    import subprocess
    from boficap_amd import hip
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(hip.LIB_PATH) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    lib = shutil.copy(hip.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    checked = 0
    for o in glob.glob(str(tmp_path / "lib.so.*amdgcn*gfx950*")):
        asm = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        if "gemm_pers_kernel" not in asm:
            continue
        funcs = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", asm)
        for f in funcs:
            head = f.split("\n", 1)[0]
            if not re.search(r"gemm_pers_kernelILi[13]E", head):            # FEAT bit 0: the folded LayerNorm
                continue
            lines = [l.split("//")[0].strip() for l in f.split("\n")[1:]]
            # the asm loads: runs of 4 or 8 loads without offset or cache modifiers, nothing but address arithmetic between them (the
            # compiler's own zero-offset loads come singly or in pairs, next to their offset: siblings)
            runs, run = [], []
            for i, l in enumerate(lines):
                m = re.fullmatch(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off", l)
                if m:
                    run.append((i, int(m.group(1)), int(m.group(2))))
                elif l.startswith(("global_", "buffer_", "scratch_", "flat_", "s_barrier", "s_cbranch", "s_branch", "s_endpgm")):      # (a wait the compiler puts between them is safe)
                    if run:
                        runs.append(run)
                    run = []
            # every load site (tile 0's, and the loop's copies) lands in the SAME registers, in the same order -- no copies between sites: the
            # statistics loads are the run that REPEATS (the residual variants' epilogue has a run of compiler loads of its own, once)
            from collections import Counter
            cand = Counter(tuple((a, b) for _, a, b in r) for r in runs if len(r) in (4, 8))
            assert cand and cand.most_common(1)[0][1] >= 2, (head, cand)
            stat_dst = cand.most_common(1)[0][0]
            assert sum(1 for d, n in cand.items() if n >= 2) == 1, (head, cand)      # one repeating run only
            sites = [r for r in runs if tuple((a, b) for _, a, b in r) == stat_dst]
            dsts = [stat_dst]
            regs = {r for a, b in dsts[0] for r in range(a, b + 1)}
            first = sites[0][0][0]
            end = next(i for i in range(first, len(lines)) if lines[i].startswith("s_endpgm"))
            bad = ("v_mov_b32", "v_accvgpr", "scratch_", "buffer_store", "global_store", "ds_write", "ds_store", "v_swap", "v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")

            def names(l):
                out = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
                    out.update(range(int(a), int(b) + 1))
                out.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
                return out
            for l in lines[first:end]:
                if l.startswith(bad) and names(l) & regs:
                    raise AssertionError(f"{head}: {l!r} touches a statistics register")
            checked += 1
    assert checked >= 4, checked


def test_bench_batches_per_launch_rule():
    """bench.py's dynamic batching: 16 batches per launch for the default 320 steps (20 launches, five per stream), 5 for the driver's 20 steps (one launch per stream: a
    region cannot hold more images in flight than it times), and a count that divides whatever else is asked for."""
    import importlib, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    assert bench.choose_coalesce(320, 4) == 16 and bench.choose_coalesce(20, 4) == 5 and bench.choose_coalesce(200, 4) == 10
    assert bench.choose_coalesce(7, 4) == 1 and bench.choose_coalesce(8, 4) == 2 and bench.choose_coalesce(400, 4) == 20
    for k in (20, 40, 64, 100, 320, 480):
        c = bench.choose_coalesce(k, 4)
        assert k % c == 0
