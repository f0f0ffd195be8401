"""pytest configuration: the ``gpu`` marker and shared fixture loaders."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def record_parity(name, measured, bar, note=""):
    """Keep the MEASURED maximum behind a tolerance as evidence: appended to gpurun_out/parity_errors.json (merged back from the GPU
    box by gpurun; the round's copy is committed as profiles/rNN_parity_errors.json).  A bar is meant to sit within ~2x of its
    measurement -- the file shows whether it does."""
    path = os.environ.get("BOFI_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity_errors.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[name] = {"measured": float(measured), "bar": float(bar), "bar_over_measured": (float(bar) / float(measured)) if measured else None, "note": note}
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass


TINY_CASES = ["tiny_mix", "tiny_q1_last_shortest", "tiny_q1_last_empty_nan", "tiny_single",
              "tiny_ragged", "tiny_ragged_short", "tiny_saic_multi"]       # the last one: multi-phrase SAIC decodes (patched [LEN] row)


@pytest.fixture(scope="session")
def weight_cache():
    """state dicts are regenerated from seeds (and checked against the fixture digests)."""
    from boficap_amd import weights as W
    from boficap_amd.config import FULL, TINY
    cache = {}

    def get(config_name, seed, gen_scale, digest=None, patch=None):
        key = (config_name, seed, gen_scale, patch)
        if key not in cache:
            from boficap_amd.config import TINY_N2
            cfg = {"TINY": TINY, "FULL": FULL, "TINY_N2": TINY_N2}[config_name]
            sd = W.make_state_dict(cfg, seed=seed, gen_scale=gen_scale)
            if patch == "len_row_shared":
                sd = W.with_len_row_shared(sd, cfg)
            elif patch is not None:
                raise KeyError(patch)
            if digest is not None:
                assert W.digest(sd) == digest, "regenerated weights differ from the ones the fixture was made with"
            cache[key] = (cfg, sd)
        return cache[key]
    return get


@pytest.fixture(autouse=True)
def _reset_training_globals():
    """The training graph keeps its GEMM compute dtype and per-step operand cache in module globals (set per forward
    pass); op-level tests assume the float32 parity mode unless they say otherwise."""
    xe = sys.modules.get("boficap_amd.xe")
    if xe is not None:
        import torch
        xe._COMPUTE["dtype"] = torch.float32
        xe._STEP_CACHE.clear()
        xe._SHADOW_ONLY.clear()
        xe.HINTS.clear()
    yield
