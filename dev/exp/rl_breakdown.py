#!/usr/bin/env python3
"""Where the self-critical step's time goes: the two sampling decodes, host scoring round trip, gradient graph, optimiser."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from types import SimpleNamespace

args = SimpleNamespace(batch=10, seq_per_img=5, dtype="bf16", no_graph=False, streams=0, gpus=1, steps=5, warmup=2)
import boficap_amd.trainer as T
orig_sample = None
from boficap_amd.transformer_model import TransformerModel
times = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); times.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w
_s = TransformerModel._sample
def sample(self, fc, att, masks=None, opt={}):
    return timed("sample_" + opt.get("train_mode", "?"), _s)(self, fc, att, masks, opt)
TransformerModel._sample = sample
if hasattr(TransformerModel, "sample_pair"):
    TransformerModel.sample_pair = timed("sample_pair", TransformerModel.sample_pair)
T.XETrainer._rl_replay = timed("grad_graph", T.XETrainer._rl_replay)
T.XETrainer.reduce_and_step = timed("reduce_and_step", T.XETrainer.reduce_and_step)
import boficap_amd.xe as xe
xe.rl_prepare = timed("rl_prepare", xe.rl_prepare)
sys.argv = ["bench.py", "--mode", "rl", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"]
t0 = time.perf_counter()
bench.main()
import statistics
print({k: round(statistics.median(v[-8:]) * 1e3, 3) for k, v in times.items()}, "ms per call, median of the last 8 steps")
