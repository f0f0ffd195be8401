"""Checkpoint files of the reference (captioning/utils/misc.py:87-102, tools/train.py:62-69,117-128,292-367):

    model[-append].pth            model.state_dict()                       (311 entries)
    optimizer[-append].pth        NoamOpt.state_dict() = torch Adam's state_dict + '_step'
    infos_<id>[-append].pkl       {'iter', 'epoch', 'loader_state_dict', 'vocab', 'opt', 'best_val_score', ...}, pickle protocol 2
    histories_<id>[-append].pkl   {'val_result_history', 'loss_history', 'lr_history', 'ss_prob_history'}

written and read so that a run of either code base resumes from the other's directory (``--start_from``).
"""
from __future__ import annotations

import os
import pickle

import torch


def pickle_dump(obj, f):
    """misc.py:33-43: protocol 2."""
    return pickle.dump(obj, f, protocol=2)


def pickle_load(f):
    """misc.py:20-30: latin-1 for pickles written by Python 2."""
    return pickle.load(f, encoding="latin-1")


def save_checkpoint(opt, model, infos, optimizer, histories=None, append=""):
    """misc.py:87-102, same arguments; ``optimizer``: anything with the reference-layout ``state_dict()`` (XETrainer)."""
    if len(append) > 0:
        append = "-" + append
    if not os.path.isdir(opt.checkpoint_path):
        os.makedirs(opt.checkpoint_path)
    checkpoint_path = os.path.join(opt.checkpoint_path, "model%s.pth" % append)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, checkpoint_path)
    print("model saved to {}".format(checkpoint_path))
    torch.save(optimizer.state_dict(), os.path.join(opt.checkpoint_path, "optimizer%s.pth" % append))
    with open(os.path.join(opt.checkpoint_path, "infos_" + opt.id + "%s.pkl" % append), "wb") as f:
        pickle_dump(infos, f)
    if histories:
        with open(os.path.join(opt.checkpoint_path, "histories_" + opt.id + "%s.pkl" % append), "wb") as f:
            pickle_dump(histories, f)


def load_infos(start_from: str, run_id: str, append: str = ""):
    """tools/train.py:62-69, 74-77: (infos, histories) of a run directory; empty dicts for files that are not there."""
    if append:
        append = "-" + append
    out = []
    for stem in ("infos_", "histories_"):
        path = os.path.join(start_from, stem + run_id + append + ".pkl")
        if os.path.isfile(path):
            with open(path, "rb") as f:
                out.append(pickle_load(f))
        else:
            out.append({})
    return tuple(out)


def check_resume_opts(saved_opt, opt, need_be_same=("caption_model", "rnn_type", "rnn_size", "num_layers")):
    """tools/train.py:66-68: the model-defining options of a resumed run must equal the saved ones."""
    for k in need_be_same:
        a, b = getattr(saved_opt, k, None), getattr(opt, k, None)
        assert a == b, "Command line argument and saved model disagree on '%s' " % k


def load_yaml_with_base(filename: str) -> dict:
    """The reference's config files inherit through ``_BASE_`` (captioning/utils/config.py:35-95, CfgNode.load_yaml_with_base):
    values of the file overwrite those of its base, recursively for nested dicts; the base path is relative to the file."""
    import yaml
    with open(filename) as f:
        cfg = yaml.safe_load(f) or {}

    def merge(a, b):
        for k, v in a.items():
            if isinstance(v, dict) and k in b:
                assert isinstance(b[k], dict), "Cannot inherit key '{}' from base!".format(k)
                merge(v, b[k])
            else:
                b[k] = v

    if "_BASE_" in cfg:
        base = cfg.pop("_BASE_")
        if base.startswith("~"):
            base = os.path.expanduser(base)
        if not base.startswith("/"):
            base = os.path.join(os.path.dirname(filename), base)
        base_cfg = load_yaml_with_base(base)
        merge(cfg, base_cfg)
        return base_cfg
    return cfg
