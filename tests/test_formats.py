"""On-disk formats of the reference on the host (no GPU): optimizer.pth layout, infos / histories pickles, yml inheritance,
the label-h5 schema -> loader batch."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_label_store_batches_equal_the_collate_loops():
    """A label file in the schema of scripts/prepro_labels_stanford.py:389-399 -> LabelStore.batch == the loader's collate kept
    as loops (oracle/training_batch.collate_loops restates dataloader.py:343-428)."""
    from boficap_amd.config import TINY as cfg
    from boficap_amd.data import LabelStore
    from training_batch import collate_loops
    rng = np.random.default_rng(0)
    S, N = cfg.seq_length, 7
    ncap = rng.integers(3, 8, N)
    M = int(ncap.sum())
    labels = np.zeros((M, S), np.uint32)
    pnum = np.zeros(M, np.uint32)
    plen = np.zeros((M, S), np.uint32)
    plab = np.zeros((M, S), np.uint32)
    for m in range(M):
        lens = rng.integers(1, 4, int(rng.integers(2, 7)))
        while lens.sum() > S:
            lens = lens[:-1]
        pnum[m] = len(lens)
        plen[m, :len(lens)] = lens
        plab[m, :len(lens)] = rng.integers(4, 7, len(lens))
        labels[m, :lens.sum()] = rng.integers(7, cfg.tgt_vocab, lens.sum())
    end = np.cumsum(ncap).astype(np.uint32)
    f = {"labels": labels, "label_start_ix": (end - ncap + 1).astype(np.uint32), "label_end_ix": end, "label_length": (labels > 0).sum(1).astype(np.uint32),
         "phrase_num": pnum, "phrase_length": plen, "phrase_label": plab}
    store = LabelStore(f)
    assert store.seq_length == S and store.num_images == N
    for spi in (5, 9):                                            # 9 > captions per image: draws with replacement
        b = store.batch([0, 3, 6], spi, np.random.default_rng(1))
        assert b["labels"].shape == (3, spi, S + 2) and (b["labels"][..., 0] == cfg.bos_idx).all() and (b["labels"][..., -1] == cfg.eos_idx).all()
        lab = b["labels"].reshape(-1, S + 2)
        # every sampled caption is one of its image's captions
        for i, ix in enumerate([0, 3, 6]):
            own = {tuple(r) for r in store.gts(ix)}
            assert all(tuple(r[1:-1]) in own for r in b["labels"][i])
        # rebuild with the loops from the same sampled captions
        rows = [int(np.where((labels.astype(np.int64) == r[1:-1]).all(1))[0][0]) for r in lab]
        ref = collate_loops(cfg, lab, pnum[rows].astype(np.int64), plen[rows].astype(np.int64), plab[rows].astype(np.int64))
        for k, v in ref.items():
            assert np.array_equal(b[k].reshape(v.shape), v), k
    with pytest.raises(KeyError):
        LabelStore({k: v for k, v in f.items() if k != "phrase_label"})


def test_infos_histories_and_yaml_base(tmp_path):
    from argparse import Namespace
    from boficap_amd import checkpoint as ck
    base = tmp_path / "base.yml"
    child = tmp_path / "sub" / "child.yml"
    child.parent.mkdir()
    base.write_text("batch_size: 10\nN_enc: 6\nnested: {a: 1, b: 2}\n")
    child.write_text("_BASE_: ../base.yml\nbatch_size: 64\nnested: {b: 3}\nstructure_after: 15\n")
    cfg = ck.load_yaml_with_base(str(child))
    assert cfg == {"batch_size": 64, "N_enc": 6, "nested": {"a": 1, "b": 3}, "structure_after": 15}

    class FakeOpt:
        def state_dict(self):
            return {"state": {}, "param_groups": [{"params": []}], "_step": 3}
    opt = Namespace(checkpoint_path=str(tmp_path / "run"), id="bofi")
    model = torch.nn.Linear(2, 2)
    infos = {"iter": 12, "epoch": 1, "vocab": {"1": "a"}, "opt": opt, "loader_state_dict": None, "best_val_score": 0.5}
    hist = {"loss_history": {10: 2.5}, "lr_history": {10: 1e-4}, "ss_prob_history": {}, "val_result_history": {}}
    ck.save_checkpoint(opt, model, infos, FakeOpt(), hist)
    ck.save_checkpoint(opt, model, infos, FakeOpt(), append="best")
    files = sorted(os.listdir(opt.checkpoint_path))
    assert files == ["histories_bofi.pkl", "infos_bofi-best.pkl", "infos_bofi.pkl", "model-best.pth", "model.pth", "optimizer-best.pth", "optimizer.pth"]
    with open(os.path.join(opt.checkpoint_path, "infos_bofi.pkl"), "rb") as f:
        raw = f.read()
    assert raw[:2] == b"\x80\x02"                                   # pickle protocol 2 (misc.py:41)
    got, gh = ck.load_infos(opt.checkpoint_path, "bofi")
    assert got["iter"] == 12 and got["opt"].id == "bofi" and gh["loss_history"] == {10: 2.5}
    assert ck.load_infos(opt.checkpoint_path, "bofi", "best")[1] == {}
    ck.check_resume_opts(Namespace(caption_model="transformer", rnn_size=2048, num_layers=6), Namespace(caption_model="transformer", rnn_size=2048, num_layers=6))
    with pytest.raises(AssertionError):
        ck.check_resume_opts(Namespace(caption_model="transformer", rnn_size=2048), Namespace(caption_model="transformer", rnn_size=512))


def _reference_resume_lines(start_from, run_id, opt):
    """tools/train.py:55-69,117-128 of the reference, restated: what it executes on a run directory before training resumes."""
    import pickle as pk
    infos = {"iter": 0, "epoch": 0, "loader_state_dict": None, "vocab": None}
    path = os.path.join(start_from, "infos_" + run_id + ".pkl")
    if os.path.isfile(path):
        with open(path, "rb") as f:
            infos = pk.load(f, encoding="latin-1")
        saved_model_opt = infos["opt"]
        for checkme in ["caption_model", "rnn_type", "rnn_size", "num_layers"]:
            assert getattr(saved_model_opt, checkme) == getattr(opt, checkme), checkme      # no default: a missing attribute raises
    histories = {}
    hp = os.path.join(start_from, "histories_" + run_id + ".pkl")
    if os.path.isfile(hp):
        with open(hp, "rb") as f:
            histories.update(pk.load(f, encoding="latin-1"))
    iteration, epoch = infos["iter"], infos["epoch"]
    if "iterators" in infos:
        infos["loader_state_dict"] = {s: {"index_list": infos["split_ix"][s], "iter_counter": infos["iterators"][s]} for s in ["train", "val", "test"]}
    loader_state = infos["loader_state_dict"]                      # unconditional index (train.py:125)
    return iteration, epoch, loader_state, infos.get("best_val_score", None), histories


def test_run_directory_written_as_tools_train_does_resumes_in_the_reference(tmp_path):
    """The writer path of tools/train.py (checkpoint.new_infos / resume_opt / save_checkpoint) -> the reference's resume lines."""
    from boficap_amd import checkpoint as ck
    from boficap_amd.config import TINY
    opt = TINY.to_opt()
    opt.id, opt.checkpoint_path = "bofi", str(tmp_path / "run")
    opt.bofi_train_dtype = torch.bfloat16                             # a knob that must not reach the pickle
    infos = ck.new_infos({"1": "a"})
    infos["iter"], infos["epoch"] = 7, 0
    infos["opt"] = ck.resume_opt(opt)

    class Opt:
        def state_dict(self):
            return {"state": {}, "param_groups": [{"params": []}], "_step": 7}
    ck.save_checkpoint(opt, torch.nn.Linear(2, 2), infos, Opt(), {"loss_history": {7: 1.0}})
    ref_opt = ck.resume_opt(opt)                                      # the reference run's own opt: same model-defining options
    it, ep, loader_state, best, hist = _reference_resume_lines(opt.checkpoint_path, "bofi", ref_opt)
    assert (it, ep, loader_state, best) == (7, 0, None, None) and hist["loss_history"] == {7: 1.0}
    saved = ck.load_infos(opt.checkpoint_path, "bofi")[0]["opt"]
    assert not hasattr(saved, "bofi_train_dtype")
    for k in ck.RESUME_KEYS:
        assert hasattr(saved, k), k
    # an infos dict written WITHOUT the skeleton keys still gets them on disk
    ck.save_checkpoint(opt, torch.nn.Linear(2, 2), {"iter": 1, "epoch": 0, "opt": ck.resume_opt(opt)}, Opt(), append="x")
    assert "loader_state_dict" in ck.load_infos(opt.checkpoint_path, "bofi", "x")[0]


def test_yaml_base_chain_of_three_and_a_loop(tmp_path):
    from boficap_amd import checkpoint as ck
    (tmp_path / "a.yml").write_text("x: 1\ny: {p: 1, q: 2}\n")
    (tmp_path / "b.yml").write_text("_BASE_: a.yml\ny: {q: 3}\nz: 5\n")
    (tmp_path / "c.yml").write_text("_BASE_: b.yml\nx: 9\n")
    assert ck.load_yaml_with_base(str(tmp_path / "c.yml")) == {"x": 9, "y": {"p": 1, "q": 3}, "z": 5}
    (tmp_path / "l1.yml").write_text("_BASE_: l2.yml\n")
    (tmp_path / "l2.yml").write_text("_BASE_: l1.yml\n")
    with pytest.raises(ValueError):
        ck.load_yaml_with_base(str(tmp_path / "l1.yml"))
    (tmp_path / "bad.yml").write_text("_BASE_: a.yml\nx: {k: 1}\n")
    with pytest.raises(AssertionError):
        ck.load_yaml_with_base(str(tmp_path / "bad.yml"))
