"""bench.py prints ONE JSON line with the fields the driver reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the `roofline` and `cpu_baseline` objects.  Run here
on a small batch (8 192 instances, T = 200, a few iterations) as a child process, the way the driver starts it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _run(["--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "8192", "--horizon", "200", "--no-full-solves",
              "--no-dense-blocks", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # value = iterations of all instances per second over the timed steps
    assert abs(d["value"] - 8192 * 4 / (d["ms_per_step"] * 4e-3)) <= 0.02 * d["value"]
    # the roofline describes the TIMED iterations (launch trace, round 6): as many iterations traced as timed, the per-iteration
    # wall times of the trace add up to the timed region (to the host-side overhead of a slice boundary), the dominant kernel's
    # launches are the timed ones, and `limited_by` is either absent or read from a committed counter file by name
    assert r["iterations_traced"] == 4 and r["launches_timed"] >= 4
    assert 0.5 * d["ms_per_step"] <= r["iteration_ms"]["mean"] <= 1.05 * d["ms_per_step"], (r["iteration_ms"], d["ms_per_step"])
    assert r["steady_state_ms_per_step"] > 0 and r["factorizations_per_launch"] > 0
    assert sum(v for k, v in r["kernel_ms_per_iteration"].items() if "gate" not in k) >= 0.9 * r["iteration_ms"]["mean"]
    assert r["limited_by"] is None or r["limited_by"]["source"].startswith("profiles/")


def test_bench_steps_form_carries_full_solves_of_the_headline_workload():
    """`--steps K` (the driver's form) times K iterations; the full solves SURVEY section 8(d) defines the metric over ride along as
    `full_solves_T<horizon>` (here on a short horizon and a small batch)."""
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2048", "--horizon", "101", "--no-dense-blocks", "--no-cpu-baseline"])
    f = d["full_solves_T101"]
    for k in ("instances", "seconds", "converged", "converged_fraction", "iterations_median_converged", "iterations_p99_converged",
              "converged_solves_per_sec", "iteration_limit"):
        assert k in f, k
    assert f["instances"] == 2048 and f["converged_fraction"] >= 0.95 and f["converged_solves_per_sec"] > 0


def test_roofline_of_a_run_with_several_slices_is_taken_over_the_first():
    """More than 25 timed iterations (the default run times 1000): the batch is repacked between slices and instances leave, so the
    `roofline` object describes the first slice -- every instance running -- and says so."""
    d = _run(["--gpus", "1", "--steps", "30", "--warmup", "0", "--batch", "8192", "--horizon", "200", "--no-full-solves",
              "--no-dense-blocks", "--no-cpu-baseline"])
    r = d["roofline"]
    assert r["iterations_traced"] == 25 and "first 25 of the 30" in r["window"]
    assert 0.0 < r["frac"] < 1.0 and r["factorizations_per_launch"] >= 0.9
    assert len(d["solve"]["profile"]) >= 2


def test_loop_only_line():
    d = _run(["--loop-only", "--steps", "3", "--warmup", "1", "--batch", "8192", "--horizon", "200"])
    assert d["steps"] == 3 and d["instances_per_gpu"] == 8192 and d["value"] > 0 and d["engine"] == "soa"


def test_cpu_baseline_object():
    """The `cpu_baseline` leg (oracle's C port in a child process started before torch / HIP) on a short horizon."""
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4096", "--horizon", "200", "--no-full-solves", "--no-dense-blocks"])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "value_1core", "scaling_vs_1core", "all_cores", "host"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert c["host"]["affinity_cpus"] >= c["cores"]


def test_collective_path_with_one_rank_over_rccl():
    """The N > 1 path of bench.py on the one GPU a test box has: started the way the driver starts several ranks
    (`python -m torch.distributed.run --nproc-per-node 1`), DTO_BENCH_FORCE_DIST=1 makes the single rank initialise RCCL and run
    every collective -- the sizing all-reduces, the barriers around the timed region, the max-over-ranks of the time and the
    chunked all-gather of the trajectories (131 072 x 9 004 doubles = 9.4 GB: two chunks under the 8 GiB bound)."""
    env = dict(os.environ, DTO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "131072",
           "--no-full-solves", "--no-dense-blocks", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["collective_backend"] == "nccl (RCCL)"
    assert d["gathered_trajectories"] == 131072 and d["value"] > 0
