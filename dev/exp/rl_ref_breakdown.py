#!/usr/bin/env python3
"""Where the reference-estimator self-critical step's time goes (the default of XETrainer.rl_step since round 5): per-phrase engine iteration, host
collate, tape-free training forward, draws, gradient pass, optimiser.  Timers synchronise: the sum is an upper bound of the un-instrumented step."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import boficap_amd.trainer as T
import boficap_amd.xe as xe
from boficap_amd.engine import BofiEngine
times = {}
def timed(name, fn):
    def w(*a, **k):
        if torch.cuda.is_current_stream_capturing() or os.environ.get("RL_TIMERS_OFF"):
            return fn(*a, **k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); times.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w
xe.rl_prepare = timed("rl_prepare (host collate)", xe.rl_prepare)
xe.rl_prepare_saic_device = timed("rl_prepare_saic_device (eager calls only)", xe.rl_prepare_saic_device)
torch.cuda.CUDAGraph.replay = timed("graph replays (13 forwards + the gradient pass + optimiser)", torch.cuda.CUDAGraph.replay)
torch._foreach_copy_ = timed("foreach copies into static buffers", torch._foreach_copy_)
_sm = torch.softmax
torch.softmax = timed("softmax of the draw", _sm)
xe.sampled_logprobs_prepared = timed("training forward (per phrase + gradient pass)", xe.sampled_logprobs_prepared)
xe.new_self_critical = timed("new_self_critical", xe.new_self_critical)
BofiEngine.decode_saic = timed("engine: one SAIC iteration", BofiEngine.decode_saic)
BofiEngine.decode_naic = timed("engine: NAIC decode", BofiEngine.decode_naic)
BofiEngine.saic_put_words = timed("engine: put words", BofiEngine.saic_put_words)
T.XETrainer.reduce_and_step = timed("reduce_and_step", T.XETrainer.reduce_and_step)
_mn = torch.multinomial
torch.multinomial = timed("multinomial", _mn)
T.XETrainer._rl_reference_step = timed("WHOLE reference step", T.XETrainer._rl_reference_step)
_rl_step = T.XETrainer.rl_step
def rl_step(self, *a, **k):
    if getattr(self.model.opt, "bofi_rl_reference_estimator", True) is False:
        os.environ["RL_TIMERS_OFF"] = "1"                        # (the fast-estimator leg of the bench: not what is measured here)
    return _rl_step(self, *a, **k)
T.XETrainer.rl_step = rl_step
sys.argv = ["bench.py", "--mode", "rl", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"]
bench.main()
os.environ["RL_TIMERS_OFF"] = "1"
n_steps = len(times["WHOLE reference step"])
for k, v in times.items():
    print(f"{k:50s} {len(v) / n_steps:6.1f} calls/step  {sum(v) / n_steps * 1e3:8.3f} ms/step  ({statistics.median(v) * 1e3:.3f} ms per call)")
