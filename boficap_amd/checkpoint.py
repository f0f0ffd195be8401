"""The run directory both code bases agree on (file names and pickle protocol are the on-disk contract:
reference captioning/utils/misc.py:87-102, tools/train.py:55-69,117-128):

    model[-tag].pth            state_dict of the 311-entry schema (boficap_amd/weights.py)
    optimizer[-tag].pth        torch Adam state_dict + NoamOpt's '_step'
    infos_<id>[-tag].pkl       iter / epoch / loader_state_dict / vocab / opt / best_val_score, pickle protocol 2
    histories_<id>[-tag].pkl   val_result_history / loss_history / lr_history / ss_prob_history, pickle protocol 2

Everything here is host glue; the bodies are this build's own.
"""
from __future__ import annotations

import os
import pickle
from argparse import Namespace

import torch

PICKLE_PROTOCOL = 2            # what the reference's loader expects (written by Python 2 era tooling too)

# options a resumed run must share with the saved one, with the defaults the reference's option parser gives them
# (captioning/utils/opts.py:60-67): a saved `opt` always carries all four, so that the reference's resume check finds them
RESUME_KEYS = {"caption_model": "show_tell", "rnn_type": "lstm", "rnn_size": 512, "num_layers": 1}

# keys the reference's training loop indexes without a default when it resumes (tools/train.py:55-60, 125)
INFOS_SKELETON = {"iter": 0, "epoch": 0, "loader_state_dict": None}


def pickle_dump(obj, f):
    pickle.dump(obj, f, protocol=PICKLE_PROTOCOL)


def pickle_load(f):
    # latin-1 also opens pickles that Python 2 wrote
    return pickle.load(f, encoding="latin-1")


def _tag(append: str) -> str:
    return "-" + append if append else ""


def run_files(directory: str, run_id: str, append: str = "") -> dict:
    """Paths of the four files of a run directory, keyed by what they hold."""
    t = _tag(append)
    return {
        "model": os.path.join(directory, f"model{t}.pth"),
        "optimizer": os.path.join(directory, f"optimizer{t}.pth"),
        "infos": os.path.join(directory, f"infos_{run_id}{t}.pkl"),
        "histories": os.path.join(directory, f"histories_{run_id}{t}.pkl"),
    }


def new_infos(vocab=None) -> dict:
    """An infos dict as a fresh run starts with: every key the reference's resume path reads unconditionally is present."""
    infos = dict(INFOS_SKELETON)
    infos["vocab"] = vocab
    return infos


def resume_opt(opt) -> Namespace:
    """`opt` as it is stored in infos['opt']: plain picklable values only, and every RESUME_KEYS entry present."""
    plain = {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, dict, list, tuple, type(None)))}
    for key, default in RESUME_KEYS.items():
        plain.setdefault(key, default)
    return Namespace(**plain)


def save_checkpoint(opt, model, infos, optimizer, histories=None, append=""):
    """Write the run directory `opt.checkpoint_path`.  `optimizer`: anything whose state_dict() has the reference layout
    (XETrainer); `histories` is skipped when empty, as the reference does."""
    paths = run_files(opt.checkpoint_path, opt.id, append)
    os.makedirs(opt.checkpoint_path, exist_ok=True)
    stored = dict(INFOS_SKELETON)
    stored.update(infos)
    writers = [
        ("model", lambda p: torch.save({name: t.detach().cpu() for name, t in model.state_dict().items()}, p)),
        ("optimizer", lambda p: torch.save(optimizer.state_dict(), p)),
        ("infos", lambda p: _dump_to(p, stored)),
    ]
    if histories:
        writers.append(("histories", lambda p: _dump_to(p, histories)))
    for what, write in writers:
        write(paths[what])
    print(f"checkpoint written: {paths['model']} (+ {', '.join(w for w, _ in writers[1:])})")


def _dump_to(path: str, obj) -> None:
    with open(path, "wb") as f:
        pickle_dump(obj, f)


def load_infos(start_from: str, run_id: str, append: str = ""):
    """(infos, histories) of a run directory; a file that is not there gives an empty dict."""
    paths = run_files(start_from, run_id, append)
    loaded = []
    for what in ("infos", "histories"):
        if os.path.isfile(paths[what]):
            with open(paths[what], "rb") as f:
                loaded.append(pickle_load(f))
        else:
            loaded.append({})
    return loaded[0], loaded[1]


def check_resume_opts(saved_opt, opt, need_be_same=tuple(RESUME_KEYS)):
    """A resumed run must be the saved model: every option of `need_be_same` has to agree (an option one side lacks counts
    as its parser default)."""
    differing = [k for k in need_be_same
                 if getattr(saved_opt, k, RESUME_KEYS.get(k)) != getattr(opt, k, RESUME_KEYS.get(k))]
    assert not differing, f"cannot resume: saved run and this run differ in {differing}"


def load_yaml_with_base(filename: str) -> dict:
    """A config file with its `_BASE_` chain applied: the reference's yml files name a base file (relative to themselves, `~`
    allowed) whose values they override, nested mappings key by key."""
    import yaml

    chain, path, seen = [], filename, set()
    while path is not None:
        real = os.path.realpath(path)
        if real in seen:
            raise ValueError(f"_BASE_ chain of {filename} loops back to {path}")
        seen.add(real)
        with open(path) as f:
            layer = yaml.safe_load(f) or {}
        base = layer.pop("_BASE_", None)
        chain.append(layer)
        if base is None:
            path = None
        else:
            base = os.path.expanduser(base)
            path = base if os.path.isabs(base) else os.path.join(os.path.dirname(path), base)

    def overlay(dst: dict, src: dict, where: str) -> None:
        for key, val in src.items():
            if isinstance(val, dict) and key in dst:
                if not isinstance(dst[key], dict):
                    raise AssertionError(f"{where}: '{key}' is a mapping here but a plain value in the base file")
                overlay(dst[key], val, where)
            else:
                dst[key] = val

    merged: dict = {}
    for layer in reversed(chain):            # root base first, the file itself last
        overlay(merged, layer, filename)
    return merged
