"""Headline benchmark: SQP iterations/sec + Jacobian nnz/sec, acrobot T=1000, 1 -> 8 GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one interior-point/SQP iteration of the whole batch resident on this rank's GPU:
derivative blocks (value, gradient, Jacobian, Hessian of the Lagrangian) -> KKT assembly ->
block-tridiagonal LDL^T factor + solve (with inertia correction) -> filter line search -> update.
Workload: config 3 of BASELINE.json, acrobot swing-up, implicit midpoint h = 0.05, T = 1000, endpoint
equality constraints, exact Hessians; `--batch` independent instances per GPU (MPC rollouts / random seeds),
guess = linear interpolation + u ~ N(0,1) from a seeded stream (rank-dependent).  Instances are sharded
across ranks with no data-path collective (weak scaling); converged trajectories are all-gathered over
RCCL after the timed region.  Inputs are resident in HBM when the timed region starts.

Default run (no --steps): K = the reference's max_iter = 1000 (src/options.jl:9), W = 0 -- the timed region is the FULL
SOLVE of every instance from its guess to the reference tolerances (or to the iteration limit), which is SURVEY.md 8(d)'s
definition of the metric:  value = sum over instances of the iterations they executed / wall time.  Instances that
terminate stop counting (and stop working).  With an explicit --steps K the same formula covers K iterations after
W warm-up iterations.  `iteration_throughput` (extra key) is the same ratio over the first 25 timed iterations, where every
instance is still running -- the round-1 headline figure.
Extra keys: jacobian_nnz_per_sec (batched MOI Jacobian callback), roofline (dominant kernel of the step,
HIP events on the launch stream), cpu_baseline (oracle C port on the host, rank 0, N = 1 only), solve (how the
instances ended), full_solves (T = 101, the reference example's horizon), dense_blocks (configs[4]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np


class _LazyTorch:
    """torch is imported on first use: main() starts the CPU-baseline child process BEFORE torch / HIP are loaded into this
    process (the all-cores leg of the baseline ran inside a process that already held torch's OpenMP runtime)."""
    _m = None

    def __getattr__(self, name):
        if _LazyTorch._m is None:
            import torch as _t
            _LazyTorch._m = _t
        return getattr(_LazyTorch._m, name)


torch = _LazyTorch()

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def make_guesses(s, p, B, seed, rng=None):
    """linear_interpolation + u ~ N(0,1) (examples/acrobot/acrobot.jl:126-127), vectorised."""
    n = s.nlp
    T, nx, nu = p["T"], p["n"], p["m"]
    rng = np.random.Generator(np.random.PCG64(seed)) if rng is None else rng
    Z = np.zeros((B, n.num_variables))
    xs = np.stack([(p["xT"] - p["x1"]) / (T - 1) * t + p["x1"] for t in range(T)])  # src/utils.jl:1-10
    U = rng.standard_normal((B, T - 1, nu))
    for t in range(T):
        o = t * (nx + nu)
        Z[:, o:o + nx] = xs[t]
        if t < T - 1:
            Z[:, o + nx:o + nx + nu] = U[:, t]
    return Z


def make_guesses_device(s, p, B, seed, dev, chunk=32768):
    """The same guesses (one seeded stream, instance after instance) built on the device chunk by chunk: the host never
    holds more than `chunk` trajectories (1.3 GB) however large the batch is."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z0 = torch.empty((B, s.nlp.num_variables), device=dev, dtype=torch.float64)
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        z0[b0:b0 + nb] = torch.from_numpy(make_guesses(s, p, nb, seed, rng=rng)).to(dev)
    return z0


def event_time_ms(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def full_solve_measurement(dev, T=101, B=1024):
    """Time to solution on a workload where (almost) every instance converges: the reference example's own horizon
    (examples/acrobot/acrobot.jl:12, T = 101), 1024 seeded guesses solved to the reference Options tolerances in one batch.
    (The headline workload, T = 1000, has non-isolated minimisers and is measured as iteration throughput: DESIGN.md 5.)"""
    import time as _time
    import dto_amd
    from dto_amd import problems as P
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    nz = s.nlp.num_variables
    z0 = torch.tensor(make_guesses(s, p, B, seed=1000), device=dev)
    zo = torch.empty_like(z0)
    st = torch.cuda.current_stream().cuda_stream
    s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, stream=st)      # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, stream=st)
    torch.cuda.synchronize()
    dt = _time.perf_counter() - t0
    # the batch returns when its last instance stops: an instance that runs into max_iter = 1000 (0-2 of the 1024, which ones
    # depends on rounding) costs ~20x the median instance; iterations_p99 / iterations_max show that tail
    return dict(workload=f"acrobot swing-up T={T} (the reference example's horizon), {B} seeded guesses, solved to tol=1e-6",
                converged=int(np.sum(status == 1)), iteration_limit=int(np.sum(status == 2)), instances=B, seconds=round(dt, 4),
                solves_per_sec=round(float(np.sum(status == 1)) / dt, 1),
                iterations_median=float(np.median(iters)), iterations_p99=float(np.percentile(iters, 99)),
                iterations_max=int(np.max(iters)), sqp_iterations_per_sec=round(float(np.sum(iters)) / dt, 1))


def full_solves_headline_workload(dev, T, B=65536, seed=1000):
    """The headline workload solved to termination inside the driver's own run (VERDICT r5 item 2c / Missing 5: SURVEY section
    8(d) defines "SQP iterations/sec" over FULL solves from the fixed guess to the reference `Options`, src/options.jl:7-13): the
    first `B` instances of the bench's seeded stream of guesses, acrobot T = `T`, one dto_solve_batch call (repacking inside)."""
    import time as _time
    import dto_amd
    from dto_amd import problems as P
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    nz = s.nlp.num_variables
    z0 = make_guesses_device(s, p, B, seed, dev)
    zo = torch.empty_like(z0)
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    status, iters = s.solve_batch(z0.data_ptr(), B, nz, zo.data_ptr(), nz, stream=st)
    torch.cuda.synchronize()
    dt = _time.perf_counter() - t0
    conv = status == 1
    out = dict(workload=f"acrobot swing-up T={T} (BASELINE.json configs[2]), the first {B} guesses of the bench's seeded stream, solved to the "
                        f"reference Options (tol 1e-6, max_iter {s.options.max_iter})",
               instances=B, seconds=round(dt, 3), converged=int(np.sum(conv)), converged_fraction=round(float(np.mean(conv)), 4),
               acceptable=int(np.sum(status == 4)), iteration_limit=int(np.sum(status == 2)),
               failed=int(np.sum((status == 3) | (status == 5))), cpu_time_limit=int(np.sum(status == 6)),
               iterations_median=float(np.median(iters)), iterations_median_converged=float(np.median(iters[conv])) if conv.any() else None,
               iterations_p99_converged=float(np.percentile(iters[conv], 99)) if conv.any() else None, iterations_max=int(np.max(iters)),
               converged_solves_per_sec=round(float(np.sum(conv)) / dt, 1), sqp_iterations_per_sec=round(float(np.sum(iters)) / dt, 1))
    s.close()
    return out


FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet, dense FP64 matrix (no local guide figure)


def dense_block_measurement(dev, T=2000, B=256):
    """BASELINE.json configs[4]: acrobot embedded in 64 states, dense 129 x 129 stage blocks, one regularised KKT
    factor + solve (dto_kkt_step_batch) on the f64 matrix cores; plus the dense-block Jacobian callback.  Flops per stage
    are SURVEY.md 8(d)'s b^3/3 + 2 b^2 n' + 2 b n'^2 with b = 129, n' = 64."""
    import dto_amd
    from dto_amd import problems as P
    p = P.build_acrobot_padded(T=T)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot_padded")
    nz, nc, nj = s.nlp.num_variables, s.nlp.num_constraint, s.nlp.num_jacobian
    g = torch.Generator(device=dev); g.manual_seed(0)
    Z = torch.rand((B, nz), device=dev, dtype=torch.float64, generator=g)
    MU = torch.rand((B, nc), device=dev, dtype=torch.float64, generator=g)
    dx, dl = torch.empty_like(Z), torch.empty_like(MU)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: s.kkt_step_batch(Z.data_ptr(), B, nz, MU.data_ptr(), nc, 2.0, 1e-5, dx.data_ptr(), nz, dl.data_ptr(), nc, stream=st)
    ok = step()
    torch.cuda.synchronize()
    ms = event_time_ms(step, 3)
    b, npr = 129, 64
    flop_stage = b ** 3 / 3 + 2 * b * b * npr + 2 * b * npr * npr
    tflops = B * (T - 1) * flop_stage / (ms * 1e-3) / 1e12
    # a few iterations of the full solve (KKT step(s) + merit evaluation + filter line search, host driven)
    import time as _time
    s.options.max_iter = 3
    zo = torch.empty_like(Z)
    xs0, us0 = p["guess"](np.random.Generator(np.random.PCG64(0)))
    dto_amd.initialize_states(s, xs0); dto_amd.initialize_controls(s, us0)
    Z0 = torch.tensor(np.tile(s._z0, (B, 1)), device=dev)
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    st_, it_ = s.solve_batch(Z0.data_ptr(), B, nz, zo.data_ptr(), nz, stream=st)
    torch.cuda.synchronize()
    sqp_dt = _time.perf_counter() - t0
    sqp_its = int(np.sum(it_))
    Bj = 16
    J = torch.empty((Bj, nj), device=dev, dtype=torch.float64)
    jfn = lambda: s.nlp.eval_constraint_jacobian_batch(Z.data_ptr(), Bj, nz, J.data_ptr(), nj, st)
    jfn()
    jms = event_time_ms(jfn, 10)
    jbytes = Bj * 8 * (nj + nz)
    return dict(workload=f"acrobot embedded in 64 states, T={T}, {B} instances (BASELINE.json configs[4])",
                kernel="k_wide_step + k_wide_bwd (forward and backward sweep of one KKT step)", inertia_ok=bool(ok), avg_launch_ms=round(ms, 3), block=129, stages=B * (T - 1),
                kkt_steps_per_sec=round(B / (ms * 1e-3), 1),
                sqp_iterations_per_sec=round(sqp_its / sqp_dt, 1), sqp_sample=f"{sqp_its} iterations in {sqp_dt:.2f} s (first 3 iterations from the straight-line guess)",
                roofline=dict(bound="mfma", achieved=round(tflops, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                              frac=round(tflops / FP64_MFMA_PEAK_TFLOPS, 5), flop_per_stage=int(flop_stage),
                              **_wide_counter_figures(B, T, ms)),
                jacobian=dict(kernel="k_wide_eval<JAC>", instances=Bj, nnz_per_sec=Bj * nj / (jms * 1e-3), avg_launch_ms=round(jms, 4),
                              achieved_GBps=round(jbytes / (jms * 1e-3) / 1e9, 1),
                              frac_of_hbm_peak=round(jbytes / (jms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)))


def _wide_counter_figures(B, T, ms):
    """What the matrix-pipe counters of the committed profiler pass say about the same kernel pair (NOT measured in this run: PMC
    counters need a profiler pass): MFMA instructions issued per KKT step -> executed flops over THIS run's launch time, and the
    fraction of the matrix pipes' cycles that were busy in the profiled run.  The formula above (SURVEY section 8d's flops per
    stage) credits more flops than the kernel issues as MFMA work; both are reported."""
    for rnd in ("r05", "r04"):
        fn = os.path.join("profiles", rnd, "pmc_wide_step_summary.json" if rnd == "r05" else "pmc_wide_step_summary_final.json")
        try:
            with open(os.path.join(ROOT, fn)) as f:
                d = json.load(f)["derived"]
            if B != 256 or T != 2000:
                return {}
            mfma = d["mfma_f64_instructions"]                  # v_mfma_f64_16x16x4_f64: 2 048 flop each
            tf = mfma * 2048.0 / (ms * 1e-3) / 1e12
            return dict(counters=dict(source=fn + " (rocprofv3 --pmc pass of tools/wide_bench.py 2000 256, tools/prof_wide_mfma.sh)",
                                      mfma_f64_instructions_per_step=int(mfma), executed_mfma_tflops=round(tf, 3),
                                      executed_frac=round(tf / FP64_MFMA_PEAK_TFLOPS, 5),
                                      matrix_pipe_busy_frac=round(d["mfma_pipe_utilisation"], 4),
                                      wavefront_waiting_frac=round(d["wave_waiting_fraction"], 4)))
        except Exception:
            continue
    return {}


def _limited_by_from_counters(kernel, B):
    """What the committed SQ-counter pass of this command says about `kernel` (read by key, never a literal: VERDICT r5 item 2b):
    share of a wavefront's cycles with the vector ALU active / spent waiting, vector instructions per launch."""
    for rnd in ("r06", "r05"):
        fn = os.path.join("profiles", rnd, f"sq_counters_soa_sweeps_B{B}_steps20.json")
        try:
            with open(os.path.join(ROOT, fn)) as f:
                d = json.load(f)
            key = next(k for k in d if k.split(" ")[0] == kernel)
            c = {k: v["mean"] for k, v in d[key].items()}
            return dict(source=fn + " (rocprofv3 --pmc SQ_* passes of bench.py --loop-only --steps 20 --warmup 5, tools/prof_sq.sh)",
                        wavefronts=int(c["SQ_WAVES"]), launches_profiled=int(d[key]["SQ_WAVE_CYCLES"]["launches"]),
                        valu_active_frac_of_wave_cycles=round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4),
                        waiting_frac_of_wave_cycles=round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                        valu_instructions_per_launch=int(c["SQ_INSTS_VALU"]),
                        vmem_instructions_per_launch=int(c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]),
                        reading="not HBM: one wavefront per SIMD issuing FP64 vector instructions, and lock-step inertia-correction rounds "
                                "(a tile pays for its slowest lane); DESIGN.md section 4.2")
        except Exception:
            continue
    return None


def cpu_baseline(a):
    """The oracle's C port on the host cores, in its own process (started before torch / HIP are loaded here)."""
    import subprocess
    try:
        iters = 0 if a.steps + a.warmup >= 1000 else a.steps + a.warmup
        res = subprocess.run([sys.executable, "-m", "oracle.cpu_port.baseline", "--horizon", str(a.horizon), "--seed", "1000",
                              "--seconds", "12", "--iters", str(iters)], cwd=ROOT, capture_output=True, text=True, timeout=240)
        return json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:  # the baseline is a reported extra, never part of the measured path
        return dict(value=None, unit="SQP iterations/s", cores=1, kind="port", sample=f"unavailable: {e}")


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: run the CPU baseline once, then N ranks as children of this
    process (`python -m torch.distributed.run`, rendezvous on 127.0.0.1), forward their output and return their exit code.
    This process never imports torch."""
    import socket
    import subprocess
    env = dict(os.environ)
    if not a.no_cpu_baseline and not a.loop_only:
        env["DTO_BENCH_CPU_BASELINE"] = json.dumps(cpu_baseline(a))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.run(cmd, env=env, cwd=os.getcwd()).returncode
    if os.environ.get("DTO_BENCH_PARENT_CHECK"):
        print(f"[bench] parent imported torch: {'torch' in sys.modules}", file=sys.stderr, flush=True)
    return rc


def launcher_selftest(rank, world, cpu):
    import torch.distributed as dist
    import torch as _t
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    t = _t.tensor([float(rank)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps(dict(launcher_selftest=True, n_gpus=world, rank_sum=float(t[0]), cpu_baseline=cpu)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed solver iterations (default: the reference's max_iter = a full solve)")
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--batch", type=int, default=524288,
                    help="instances per GPU (default sized for the 288 GB of an MI355X: ~0.36 MB of solver state per instance, ~70 GB stay free; "
                         "> 65536 instances run the plain sequential sweeps, one wavefront per SIMD = 1024 tiles at a time; the larger batch keeps the "
                         "GPU filled while instances converge and leave, DESIGN.md sections 4.2, 7)")
    ap.add_argument("--no-full-solves", action="store_true", help="skip the T=101 time-to-solution side measurement")
    ap.add_argument("--loop-only", action="store_true",
                    help="run only the headline loop (warm-up + timed iterations) and print value / ms_per_step: the command "
                         "profiled under rocprofv3 so that the kernel statistics contain nothing but that loop")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-blocks", action="store_true", help="skip the configs[4] (dense 129x129 blocks) side measurement")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="(tests) ranks only join a gloo group on the CPU, all-reduce their rank and rank 0 prints one line: "
                         "exercises the self-launch path of --gpus N without a GPU")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes
        # (torch.distributed.run), before this process has imported torch or touched HIP (never an exec after a GPU call);
        # the CPU baseline runs once, here in the parent, and is handed to rank 0 through the environment.
        sys.exit(self_launch(a))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ---- CPU baseline (rank 0, N = 1 only): its own process, started and finished before torch / HIP are loaded here
    cpu = None
    if rank == 0 and os.environ.get("DTO_BENCH_CPU_BASELINE"):      # measured by the self-launching parent
        cpu = json.loads(os.environ["DTO_BENCH_CPU_BASELINE"])
    elif rank == 0 and world == 1 and not a.no_cpu_baseline and not a.loop_only:
        cpu = cpu_baseline(a)
    if a.launcher_selftest:
        launcher_selftest(rank, world, cpu)
        return
    dist = None
    # DTO_BENCH_FORCE_DIST=1: initialise RCCL and run every collective of the N > 1 path with ONE rank too (what a 1-GPU box can show
    # of the multi-GPU path on hardware: process-group set-up, the sizing all-reduces, the chunked all-gather of the trajectories)
    force_dist = os.environ.get("DTO_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k_, v_)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import dto_amd
    from dto_amd import problems as P
    from dto_amd.parallel import gather_trajectories

    T, B = a.horizon, a.batch
    # Safety net for the memory-sized default: solver state (~0.36 MB per instance at T = 1000, 0.49 MB when the batch may
    # switch to the chunked form) + z0, zout (0.08 MB) must fit what is free on this GPU with 5 GB to spare (the estimate is the
    # measured high-water mark); shrink by whole residencies of the sequential sweep (131 072 instances) if another process
    # holds part of the HBM.
    if dist is not None:
        # RCCL allocates its channel buffers at the first collective: size the batch against what is free AFTER one
        warm = torch.zeros(1, device=dev, dtype=torch.int64)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    free_b = torch.cuda.mem_get_info(dev)[0]
    # (+ 0.072 MB per instance for the second iterate / multiplier buffers of the fused UPDATE+EVAL pass: 8 (N_z + N_c) bytes)
    # measured at the default batch: 301.6 GB in use at the high-water mark (hbm_free_min_gb 7.4 of 309) = this estimate;
    # with several ranks 8 GB more stay free for what RCCL and the gather of the trajectories allocate later.
    per_inst_of = lambda b: (0.41e6 if b > 131072 else 0.54e6) * (T / 1000.0) + 2 * 8.0 * 5 * T + 8.0 * 9 * T
    margin = 5e9 if dist is None else 13e9
    while B > 131072 and B * per_inst_of(B) + 7e9 > free_b - margin:
        B -= 131072
    if dist is not None:   # every rank runs the same shard size: the smallest any of them could take
        bt = torch.tensor([B], device=dev, dtype=torch.int64)
        dist.all_reduce(bt, op=dist.ReduceOp.MIN)
        B = int(bt[0])
    if B != a.batch and rank == 0:
        print(f"[bench] batch reduced from {a.batch} to {B} instances: only {free_b / 1e9:.0f} GB of HBM free", file=sys.stderr, flush=True)
    p = P.build_acrobot(T=T, evaluate_hessian=True)
    s = dto_amd.Solver(p["dynamics"], p["objective"], p["constraints"], p["bounds"], evaluate_hessian=True, name="acrobot")
    n = s.nlp
    nz, nc, nj, nh = n.num_variables, n.num_constraint, n.num_jacobian, int(n.sizes.nnz_hess_key)
    st = torch.cuda.current_stream().cuda_stream

    # ---- solver: W warmup iterations, then exactly K timed iterations (in slices of 25 so that the throughput profile of the
    #      solve can be reported; a slice boundary is a stream synchronisation, nothing else)
    s.options.max_iter = max(1000, a.steps + a.warmup)          # the reference default; never cut a longer requested run short
    # An allocation failure in dto_solver_begin (DTO_ERR_DEVICE) must not cost the headline number: retry with one residency of
    # the sequential sweep (131 072 instances) fewer, on every rank alike, and say so.
    while True:
        ok, z0 = 1, None
        try:
            # (inside the try: after a failed allocation HIP's last-error state used to outlive the failure and the next torch
            #  launch tripped over it -- hip_fail now reads it, and a failure here is retried like one in begin: ADVICE r5)
            z0 = make_guesses_device(s, p, B, 1000 + rank, dev)
            s.begin_batch(z0.data_ptr(), B, nz, stream=st)
        except RuntimeError:
            ok = 0
        if dist is not None:
            okt = torch.tensor([ok], device=dev, dtype=torch.int64)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt[0])
        if ok:
            break
        if B <= 131072:
            raise RuntimeError(f"dto_solver_begin failed at the smallest batch ({B} instances)")
        s.release_state()
        del z0
        torch.cuda.empty_cache()
        if rank == 0:
            print(f"[bench] dto_solver_begin failed at {B} instances; retrying with {B - 131072}", file=sys.stderr, flush=True)
        B -= 131072
    if a.warmup:
        s.iterate_batch(a.warmup, stream=st)
    torch.cuda.synchronize()
    it0 = s.scalar_batch("iter").copy()
    nf0 = s.scalar_batch("nfact").copy()
    # every kernel launch of the TIMED iterations carries a HIP event pair on the stream it is launched on (include/dto.h:
    # dto_solver_trace; read after the timed region): the `roofline` object below describes exactly the iterations `value` counts
    if not a.loop_only:
        s.trace(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, profile, first_slice = 0, [], None
    while done < a.steps:
        k = min(25, a.steps - done)
        s.iterate_batch(k, stream=st)
        done += k
        if done < a.steps:
            running = s.repack_batch(stream=st)      # finished instances leave the tiles (part of the solve, hence timed)
            torch.cuda.synchronize()
            profile.append((done, time.perf_counter() - t0, running))
            if first_slice is None:
                # counters at the end of the first slice (every instance was running, the tiles are as loaded): the window the
                # `roofline` object of a full-solve run describes (one read of two scalar rows, ~20 ms once in the timed region)
                first_slice = (done, float(np.sum(s.scalar_batch("iter") - it0)), float(np.sum(s.scalar_batch("nfact") - nf0)))
    # one more evaluation classifies the last iterate (converged / iteration limit); it is part of the solve
    if a.steps + a.warmup >= s.options.max_iter:
        s.launch_op("eval", stream=st)
        s.launch_op("conv", stream=st)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    s.trace(False)
    profile.append((done, dt, None))
    it1 = s.scalar_batch("iter")
    nf1 = s.scalar_batch("nfact")
    status_end = s.scalar_batch("status").copy()
    iters_done = float(np.sum(it1 - it0))
    facts_done = float(np.sum(nf1 - nf0))
    tt = torch.tensor([dt, iters_done, facts_done], device=dev, dtype=torch.float64)
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        iters_done, facts_done = float(tt[1]), float(tt[2])
    # iteration throughput while every instance is still running: the first slice (this rank; aggregated below)
    first_k, first_t = profile[0][0], profile[0][1]
    thr0 = torch.tensor([B * first_k / first_t], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(thr0, op=dist.ReduceOp.SUM)
    if a.loop_only:
        if rank == 0:
            print(json.dumps(dict(value=iters_done / dt, unit="SQP iterations/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                                  ms_per_step=dt / a.steps * 1e3, instances_per_gpu=B, time_partitions=s.partitions(), engine=s.engine(),
                                  factorizations_per_iteration=round(facts_done / max(iters_done, 1.0), 3))), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    solve_info = dict(converged=int(np.sum(status_end == 1)), acceptable=int(np.sum(status_end == 4)),
                      iteration_limit=int(np.sum(status_end == 2)), failed=int(np.sum((status_end == 3) | (status_end == 5))),
                      still_running=int(np.sum(status_end == 0)), instances=B,
                      iterations_median=float(np.median(it1)), iterations_mean=float(np.mean(it1)),
                      converged_solves_per_sec=round(float(np.sum(status_end == 1)) / dt, 1),
                      profile=[dict(iterations=int(k), seconds=round(t, 3), running=r) for k, t, r in profile[:: max(1, len(profile) // 10)]]
                      + [dict(iterations=int(profile[-1][0]), seconds=round(profile[-1][1], 3), running=int(np.sum(status_end == 0)))],
                      note="rank 0's shard; status per instance after the timed iterations")

    # ---- per-kernel durations OF THE TIMED ITERATIONS (the launch trace recorded above)
    fp = s.footprint()
    seq_sweep = s.partitions() == 1
    tr = s.read_trace()
    # a full-solve run repacks between slices and may change the form of its sweeps: its `roofline` is taken over the first slice
    # (the iterations in which every instance runs); a --steps run has one slice: all of it
    roof_iters, roof_its_done, roof_facts_done = a.steps, iters_done / world, facts_done / world
    if first_slice is not None:
        roof_iters, roof_its_done, roof_facts_done = int(first_slice[0]), first_slice[1], first_slice[2]
        keep = tr["iteration"] < roof_iters
        tr = dict(op=tr["op"][keep], name=[n_ for n_, k_ in zip(tr["name"], keep) if k_], iteration=tr["iteration"][keep],
                  start_ms=tr["start_ms"][keep], duration_ms=tr["duration_ms"][keep])
    kname = dict(eval="k_stage_eval", conv="k_conv (+ k_part_reduce)", kkt_fwd="k_kkt_fwd_seq" if seq_sweep else "k_kkt_fwd",
                 kkt_bwd="k_kkt_bwd_seq" if seq_sweep else "k_kkt_bwd", kkt_post="k_kkt_post", linesearch="k_linesearch",
                 ls_reduce="k_ls_reduce (+ k_part_reduce)", update="k_update", update_eval="k_update_eval",
                 kkt_bwd_early="k_kkt_bwd_early", kkt_bwd_rest="k_kkt_bwd_rest", kkt_bwd_gate="k_kkt_bwd_gate (one wavefront waiting, second stream)",
                 factor_solve=("k_kkt_fwd_seq + k_kkt_bwd_seq + k_kkt_post" if seq_sweep else "k_kkt_fwd / k_kkt_sep rounds + k_kkt_bwd + k_kkt_post"),
                 kkt_sep="k_kkt_sep", kkt_refine="k_kkt_refine + k_kkt_refine_join")
    ops = sorted(set(tr["name"]))
    dur = {o: tr["duration_ms"][np.array([n == o for n in tr["name"]])] for o in ops}
    n_it = int(tr["iteration"].max()) + 1 if len(tr["op"]) else 0
    tot = {o: float(np.sum(dur[o])) for o in ops}
    avg_ms = {o: float(np.mean(dur[o])) for o in ops}
    per_iter_ms = {o: tot[o] / max(n_it, 1) for o in ops}
    # wall time of every timed iteration: from the start of its first launch to the start of the next iteration's (the last one:
    # to the end of its last launch) -- they add up to the timed region minus the repack / synchronisation of a slice boundary
    it_start = np.array([tr["start_ms"][tr["iteration"] == k].min() for k in range(n_it)]) if n_it else np.zeros(0)
    it_end = np.append(it_start[1:], (tr["start_ms"] + tr["duration_ms"]).max()) if n_it else np.zeros(0)
    it_ms = it_end - it_start
    tail = it_ms[-min(8, n_it):] if n_it else np.zeros(1)
    nfact_per_iter = facts_done / max(iters_done, 1.0)
    dom_op = max((o for o in ops if o != "kkt_bwd_gate"), key=lambda o: tot[o]) if ops else "kkt_fwd"
    launches = max(len(dur.get(dom_op, [])), 1)
    # factorisations / accepted factorisations per instance and launch of the dominant kernel, over the timed window (this rank)
    working = roof_facts_done / max(B * launches, 1)
    accepted = roof_its_done / max(B * launches, 1)
    # Algorithmic bytes per instance and launch.  The sweeps no longer read derivative values: they
    # re-evaluate them from the iterate.  What a sweep MUST move per stage is therefore: the iterate (p_t, x_{t+1}, lambda_t,
    # nu_t), the stage's right-hand side (record: r_p, d, c) and the carry that the backward sweep resumes from
    # (P_t: n(n+1)/2, p_y: n; plus the n x n spike coupling when the horizon is cut into chunks).
    nx_ = p["n"]
    parts = s.partitions()
    rec_d = fp["record_doubles"]
    carry_d = (T - 1) * (nx_ * (nx_ + 1) // 2 + nx_ + (nx_ * nx_ if parts > 1 else 0))
    sweep_read = 8 * (nz + (T - 1) * nx_ + nc + rec_d)
    fact_bytes = sweep_read + 8 * carry_d                       # one forward factorisation of one instance
    bwd_bytes = sweep_read + 8 * carry_d + 8 * (nz + nc)        # one back substitution (+ the step written)
    alg_bytes = dict(                                           # per instance and per launch
        eval=8 * (2 * nz + 2 * nc) + 8 * rec_d + 8 * 10 * T,    # iterate (+ previous stage for E'lambda), record and partials out
        kkt_fwd=working * fact_bytes,                           # per factorisation actually done by a lane
        kkt_bwd=bwd_bytes, kkt_bwd_early=bwd_bytes, kkt_bwd_rest=bwd_bytes,   # (early + rest together visit every tile once)
        factor_solve=working * fact_bytes + bwd_bytes,
        linesearch=8 * (2 * nz + (T - 1) * 2 * nx_) + 8 * 16 * T, update=8 * 3 * (nz + nc), conv=8 * 10 * T, ls_reduce=8 * 16 * T,
        kkt_sep=8 * 64, kkt_post=8 * 4, kkt_bwd_gate=0, kkt_refine=2 * 8 * rec_d + 8 * (2 * nz + 2 * nc),
        update_eval=8 * 3 * (nz + nc) + 8 * rec_d + 8 * 10 * T)   # z, dz, lam, dlam in; z', lam', record, partials out
    achieved = B * alg_bytes.get(dom_op, 0.0) / (avg_ms[dom_op] * 1e-3) / 1e9 if ops else 0.0
    achieved_accepted = B * accepted * fact_bytes / (avg_ms[dom_op] * 1e-3) / 1e9 if ops and dom_op == "kkt_fwd" else None
    # HBM bytes per launch: NOT measured in this run (PMC counters need a profiler pass) -- taken from the committed
    # rocprofv3 --pmc passes of this same command and batch size if there are any, with the file named; else null
    traffic, traffic_source = None, None
    pmc_name = dict(kkt_fwd="k_kkt_fwd_seq" if seq_sweep else "k_kkt_fwd").get(dom_op, kname.get(dom_op, dom_op))
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        fn = os.path.join("profiles", rnd, f"pmc_traffic_acrobot_T{T}_B{B}.json")
        try:
            with open(os.path.join(ROOT, fn)) as f:
                traffic = json.load(f)["kernels"][pmc_name]["hbm_bytes_per_launch_mean"]
            traffic_source = fn + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --loop-only, tools/profile_headline.sh)"
            break
        except Exception:
            continue
    # `bound` follows the contract (the kernel is priced against the HBM roofline: it is a streaming sweep whose arithmetic is
    # FP64 vector, not MFMA); what the SQ counters of the committed profiler pass say limits it is read from that file BY KEY
    limited_by = _limited_by_from_counters(pmc_name, B)
    roofline = dict(kernel=kname.get(dom_op, dom_op), bound="hbm", limited_by=limited_by, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_source,
                    avg_launch_ms=round(avg_ms.get(dom_op, 0.0), 5), launches_timed=int(launches),
                    launches_per_iteration=round(launches / max(n_it, 1), 3),
                    algorithmic_bytes_per_launch=int(B * alg_bytes.get(dom_op, 0.0)),
                    factorizations_per_launch=round(working, 4), accepted_factorizations_per_launch=round(accepted, 4),
                    achieved_accepted_work=None if achieved_accepted is None else round(achieved_accepted, 2),
                    frac_accepted_work=None if achieved_accepted is None else round(achieved_accepted / HBM_PEAK_GBS, 5),
                    bound_note="priced against the HBM roofline as the bench contract prescribes (streaming sweep, FP64 vector arithmetic, no MFMA); "
                               "see limited_by for what the counters say limits this kernel; frac counts every factorisation a lane did in the "
                               "timed launches, frac_accepted_work only the one per running instance and launch that the step uses",
                    source="HIP event pairs around every launch of the timed iterations, on the stream of the launch (dto_solver_trace); "
                           "k_kkt_bwd_early runs on the library's second stream beside k_kkt_fwd_seq: the per-kernel figures add up to more than "
                           "the iteration by what overlaps (overlapped_ms_per_iteration)",
                    iterations_traced=n_it,
                    window=(f"the {n_it} timed iterations" if first_slice is None else
                            f"the first {n_it} of the {a.steps} timed iterations (before the first repack: every instance running)"),
                    iteration_ms=dict(mean=round(float(np.mean(it_ms)), 4) if n_it else None, first=round(float(it_ms[0]), 4) if n_it else None,
                                      last=round(float(it_ms[-1]), 4) if n_it else None, min=round(float(np.min(it_ms)), 4) if n_it else None,
                                      max=round(float(np.max(it_ms)), 4) if n_it else None),
                    steady_state_ms_per_step=round(float(np.mean(tail)), 4),
                    steady_state_note=f"mean of the last {len(tail)} timed iterations (the window mean mixes the cheap penalty-phase iterations "
                                      "with the filter phase's: DESIGN.md section 7)",
                    dominant_kernel_ms_first_last=[round(float(dur[dom_op][0]), 3), round(float(dur[dom_op][-1]), 3)] if ops else None,
                    # (the gate kernel is one wavefront WAITING on the second stream until every forward block has started: not work)
                    overlapped_ms_per_iteration=round(sum(v for k, v in per_iter_ms.items() if k != "kkt_bwd_gate") - (float(np.mean(it_ms)) if n_it else 0.0), 4),
                    kernel_ms_per_iteration={kname.get(k, k): round(v, 4) for k, v in per_iter_ms.items()},
                    kernel_avg_launch_ms={kname.get(k, k): round(v, 5) for k, v in avg_ms.items()})

    # ---- Jacobian assembly (the MOI callback, instance-major, reference COO order)
    Bj = min(B, 32768)                                         # 6.8 GB of output: enough to fill the GPU, leaves HBM to the solver state
    jout = torch.empty((Bj, nj), device=dev, dtype=torch.float64)
    jfn = lambda: n.eval_constraint_jacobian_batch(z0.data_ptr(), Bj, nz, jout.data_ptr(), nj, st)
    for _ in range(3):
        jfn()
    jac_ms = event_time_ms(jfn, 20)
    jac_bytes = Bj * (8 * nz + 8 * nj)
    jac = dict(kernel="k_jac", instances=Bj, nnz_per_sec=Bj * nj / (jac_ms * 1e-3), avg_launch_ms=round(jac_ms, 4),
               achieved_GBps=round(jac_bytes / (jac_ms * 1e-3) / 1e9, 1),
               frac_of_hbm_peak=round(jac_bytes / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    jt = torch.tensor([jac["nnz_per_sec"]], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(jt, op=dist.ReduceOp.SUM)

    # ---- converged trajectories: finish the solves of this shard and all-gather them (RCCL over xGMI).  The solver state
    #      (and the bench's own buffers) go back first: at 8 ranks every rank receives 8 x 10.5 GB of trajectories, and the
    #      side measurements below build their own problems.
    zout = torch.empty((B, nz), device=dev, dtype=torch.float64)
    hbm_free_min_gb = round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1)   # the bench's high-water mark: solver state + z0 + Jacobian + zout
    s.end_batch(zout.data_ptr(), nz, stream=st)
    status = torch.tensor(s.scalar_batch("status"), device=dev, dtype=torch.float64)
    time_partitions = s.partitions()
    engine_name = s.engine()
    del jout, z0, jfn
    s.close()
    torch.cuda.empty_cache()
    # bounded exchange (dto_amd/parallel.py): <= 8 GiB of receive buffer per collective, every chunk consumed at once (here: a
    # checksum) -- at the default batch a rank holds its own 21 GB of trajectories + <= 9 GB of exchange buffers
    gsum = torch.zeros((), device=dev, dtype=torch.float64)
    def _sink(g0, rows):
        gsum.add_(rows[:, -1].sum())
    n_gathered = int(gather_trajectories(zout, status, dist, sink=_sink, force_collective=force_dist))
    del zout, status
    torch.cuda.empty_cache()

    full = None
    if rank == 0 and world == 1 and not a.no_full_solves:
        try:
            full = full_solve_measurement(dev)
        except Exception as e:  # side measurement, never part of `value`
            full = dict(error=str(e))

    # the --steps form times K iterations; the time to solution of the same workload rides along as a side block
    full_T = None
    if rank == 0 and world == 1 and not a.no_full_solves and a.steps + a.warmup < 1000:
        try:
            full_T = full_solves_headline_workload(dev, T, B=min(65536, B))
        except Exception as e:  # side measurement, never part of `value`
            full_T = dict(error=str(e))

    dense = None
    if rank == 0 and world == 1 and not a.no_dense_blocks:
        try:
            dense = dense_block_measurement(dev)
        except Exception as e:  # a side measurement of another config, never part of `value`
            dense = dict(error=str(e))

    if rank == 0:
        out = dict(
            metric="SQP iterations/sec + Jacobian nnz/sec, acrobot T=1000, 1->8 GPU",
            value=iters_done / dt, unit="SQP iterations/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
            ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="f64", data="synthetic",
            config=dict(workload=f"acrobot swing-up, implicit midpoint h=0.05, T={T}, exact Hessians, "
                                 f"{B} independent instances per GPU (BASELINE.json configs[2])",
                        horizon=T, instances_per_gpu=B, instances_total=B * world, num_variables=nz, num_constraint=nc,
                        jacobian_nnz=nj, hessian_key=nh, parallelism=f"instance sharding x{world}, all-gather of trajectories",
                        collective_backend=("nccl (RCCL)" if dist is not None else None)),
            jacobian_nnz_per_sec=float(jt[0]), jacobian=jac,
            iteration_throughput=dict(value=float(thr0[0]), unit="SQP iterations/s",
                                      note=f"first {first_k} timed iterations, every instance still running"),
            solve=solve_info,
            factorizations_per_iteration=round(nfact_per_iter, 3), time_partitions=time_partitions,
            gathered_trajectories=n_gathered, gathered_status_sum=float(gsum), hbm_free_min_gb=hbm_free_min_gb, engine=engine_name,
            roofline=roofline, cpu_baseline=cpu, full_solves=full, dense_blocks=dense,
            **({f"full_solves_T{T}": full_T} if full_T is not None else {}),
        )
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
