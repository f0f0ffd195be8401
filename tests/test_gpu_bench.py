"""The driver's contract with bench.py: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line last, with the fields the
driver parses, the roofline and CPU-baseline objects, and a config that names the workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    if env and env.get("BOFI_BENCH_REHEARSAL") == "1":         # the rehearsal's transport (gloo) announces itself on stdout, the ranks' lines interleaved
        lines = [l for l in lines if l.startswith("{")]
    assert len(lines) == 1, lines[:3]                          # progress goes to stderr
    return json.loads(lines[-1])


def test_bench_line_with_the_drivers_arguments():
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-secondary", "--cpu-budget", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]           # value = images per step / time per step
    c = d["config"]
    assert "workload" in c and "batch=64" in c["workload"] and c["images_per_step_per_gpu"] == 64 and not c["nan_in_output"]
    assert c["batches_per_launch"] == 5 and c["decodes_in_flight"] >= 1              # 20 steps = 4 launches of 5 batches
    # round 5: the bounding loop is one persistent kernel that ends by itself (no iterations are enqueued, no budget); were it off, a verified budget
    assert "bound_loop" in c and "knobs" in c and "iteration_budget" in c
    if c["bound_iterations_enqueued"] is None:
        assert "persistent kernel" in c["bound_loop"] and 1 <= c["bound_iterations"] <= 20
    else:
        assert c["bound_iterations"] < c["bound_iterations_enqueued"] <= 20
    # a short region is timed five times back to back and the median reported; every stream is warmed
    assert c["timed_regions"] == 5 and len(c["region_ms"]) == 5 and c["region_ms"] == sorted(c["region_ms"]) and c["warmup_steps_run"] >= 5 * c["decodes_in_flight"]
    assert abs(d["ms_per_step"] * 20 - c["region_ms"][2]) < 0.02 * c["region_ms"][2]
    # roofline_gemm times the kernels the engine RUNS at the measured size: the row-block sublayer kernels at 5 batches per launch
    assert any("rb_ffn" in g["kernel"] for g in d["roofline_gemm"]) and any("rb_gemm" in g["kernel"] for g in d["roofline_gemm"])
    assert all(0 < g["frac"] < 1 and g["us_per_launch"] > 0 for g in d["roofline_gemm"])
    r = d["roofline"]
    assert r["one_at_a_time"]["frac"] > 0 and r["one_at_a_time"]["launch_ms"] > r["launch_ms"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["flops_per_launch"] > 0 and r["launch_ms"] > 0 and "traffic" in r
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "images/sec" and b["cores"] >= 1 and b["value"] > 0 and b["sample"]
    assert d["value"] > 20 * b["value"]


def test_bench_odd_step_counts_and_one_batch_per_launch():
    d = _run("--steps", "7", "--warmup", "3", "--no-secondary", "--no-cpu-baseline", "--no-gemm-roofline")
    assert d["steps"] == 7 and d["config"]["batches_per_launch"] == 1 and d["value"] > 0
    d = _run("--steps", "8", "--warmup", "2", "--inflight", "1", "--coalesce", "1", "--no-secondary", "--no-cpu-baseline", "--no-gemm-roofline", "--ids-only",
             "--iter-budget", "off")
    assert d["config"]["decodes_in_flight"] == 1 and d["config"]["seq_logprob_materialised"] is False and d["config"]["bound_iterations_enqueued"] == 20


def test_bench_two_ranks_walk_the_multi_gpu_path_on_one_device():
    """`python bench.py --gpus 2` as the driver starts it (self-launch through torch.distributed.run, one process per rank), rehearsed on
    the one GPU of the box: both ranks on device 0 over gloo (RCCL refuses two ranks on one device).  Checks what a scaling run needs:
    ONE line from rank 0, n_gpus 2, value = the images of BOTH ranks over the slowest rank's time, and the XE exchange secondary."""
    d = _run("--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-gemm-roofline", "--no-from-host",
             env={"BOFI_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d["config"]
    assert abs(d["value"] - 2 * 64 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]       # whole-job aggregate: 64 images per rank and step
    dp = d["secondary"]["xe_config3_dp"]
    assert "error" not in dp, dp
    assert dp["rccl_ranks"] == 2 and dp["fp32_ring_all_reduce"]["step_ms"] > 0 and dp["bf16_mesh_direct"]["step_ms"] > 0
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["dist_backend"] == "gloo" and "xe_dp_error" not in d["config"] and d["config"]["xe_dp_step_ms"] > 0
    assert dp["fp32_ring_all_reduce"]["exchanged_bytes_per_rank"] == 2 * dp["bf16_mesh_direct"]["exchanged_bytes_per_rank"]


def test_bench_reruns_without_the_iteration_budget_when_a_decode_outruns_it():
    """--iter-budget auto enqueues (live iterations of the probe + 1) bounding iterations per decode and checks the device-side maximum of the
    decodes' live-iteration counts after every leg; with a budget the decodes cannot meet (forced here) the line must come from the re-run
    that enqueues all of them.  (The budget belongs to the five-launch iterations: BOFI_BOUND_LOOP=0; the persistent loop kernel ends by itself.)"""
    d = _run("--steps", "10", "--warmup", "5", "--no-secondary", "--no-cpu-baseline", "--no-gemm-roofline", "--no-from-host",
             env={"BOFI_BENCH_ITER_CAP": "3", "BOFI_BOUND_LOOP": "0"})
    c = d["config"]
    assert c["knobs"].get("BOFI_BOUND_LOOP") == "0" and "five launches" in c["bound_loop"]
    assert c["bound_iterations"] > 3 and c["bound_iterations_enqueued"] == 20 and c["iteration_budget"].startswith("off") and not c["nan_in_output"]
