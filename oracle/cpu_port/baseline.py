"""ORACLE / CPU baseline, run as its own process:  python -m oracle.cpu_port.baseline --horizon T --seconds S [--iters K]

bench.py starts this BEFORE it imports torch / initialises HIP, so that the OpenMP run of the C port sees an untouched
process: no second OpenMP runtime loaded by torch, no affinity mask or thread pool left behind by another library.  Prints
one JSON object.  What it reports, and why ("4 576 it/s on 256 threads vs 814 on one core"):
  * the host as this process may use it: CPUs in the affinity mask, cgroup CPU quota, logical CPUs of the machine;
  * the one-core rate and the rate on T threads, T = min(affinity, quota) -- the threads it can actually be scheduled on;
  * the scaling value / value_1core and the per-thread rate.  If the scaling is below 0.4 x T the figure is NOT called an
    all-cores figure: `all_cores` is false and `note` says so (a cpuset / quota / SMT-bound host, or an oversubscribed box).
Test infrastructure: nothing of the product imports this.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def host_limits():
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            quota = None
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(affinity_cpus=aff, cgroup_cpu_quota=quota, logical_cpus=os.cpu_count() or 1, cpu_model=model)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--iters", type=int, default=0, help="iterations per instance (0: full solves to the reference tolerances)")
    ap.add_argument("--model", default="acrobot")
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    host = host_limits()
    threads = host["affinity_cpus"]
    if host["cgroup_cpu_quota"]:
        threads = max(1, min(threads, int(host["cgroup_cpu_quota"] + 0.5)))
    os.environ["OMP_NUM_THREADS"] = str(threads)       # before libgomp is loaded
    os.environ.setdefault("OMP_PROC_BIND", "false")
    import numpy as np
    from oracle.cpu_port import guesses, run_batch
    Z, _, _ = guesses(a.model, a.horizon, a.batch, a.seed)
    it = a.iters
    # pilot runs size the samples to the time budget
    _, p1, _, _, _ = run_batch(a.model, a.horizon, Z[:1], iters_per_instance=it, threads=1)
    n1 = int(max(1, min(a.batch, a.seconds / max(p1, 1e-9))))
    it1, dt1, _, st1, _ = run_batch(a.model, a.horizon, Z[:n1], iters_per_instance=it, threads=1)
    _, pa, _, _, _ = run_batch(a.model, a.horizon, Z[:min(a.batch, threads)], iters_per_instance=it, threads=threads)
    nall = int(max(threads, min(a.batch, threads * max(1.0, (a.seconds - pa) / max(pa, 1e-9)))))
    nall = min(a.batch, (nall // threads) * threads if nall >= threads else nall)
    ita, dta, _, sta, nfa = run_batch(a.model, a.horizon, Z[:nall], iters_per_instance=it, threads=threads)
    v1, va = it1 / dt1, ita / dta
    scaling = va / v1
    all_cores = scaling >= 0.4 * threads
    what = f"{it} iterations" if it > 0 else "full solves (tol 1e-6, max_iter 1000)"
    # the port keeps every stage's blocks and factors (oracle/cpu_port/solver_port.c: stage_t, ~3.6 KB per stage): one iteration
    # streams that workspace about three times, so many threads at once are bound by the memory system, not by the cores
    ws_mb = 3.6e-3 * a.horizon
    note = ("scales with the threads used" if all_cores else
            f"NOT an all-cores figure: {threads} threads give only {scaling:.1f}x one core ({va / threads:.1f} it/s per thread against "
            f"{v1:.1f}).  Either the host confines the process (cpuset, CPU quota, SMT siblings, a shared box: see `host`) or the "
            f"port is bound by memory bandwidth ({ws_mb:.1f} MB of stage workspace per instance streamed ~3x per iteration, "
            f"{ws_mb * threads:.0f} MB for {threads} threads); effective parallelism ~{max(1, round(scaling))} cores")
    print(json.dumps(dict(
        value=va, unit="SQP iterations/s", cores=threads, kind="port",
        sample=f"{nall} instances x {what} of {a.model} T={a.horizon} (same guesses as GPU rank 0), {dta:.1f} s on {threads} threads "
               f"(OpenMP over instances, own process started before torch/HIP), {nfa / max(ita, 1):.2f} factorizations/iteration",
        value_1core=v1, sample_1core=f"{n1} instances, {dt1:.1f} s on 1 thread",
        scaling_vs_1core=round(scaling, 2), per_thread_rate=round(va / threads, 2), all_cores=bool(all_cores), note=note,
        converged_fraction=float(np.mean(sta == 1)) if it <= 0 else None,
        host=host)))


if __name__ == "__main__":
    main()
