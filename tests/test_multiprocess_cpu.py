"""Multi-process path on CPU (gloo, world_size 2): instance sharding and the all-gather of trajectories."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

import dto_amd
from dto_amd.parallel import shard_range


def test_shard_ranges_partition_the_batch():
    for B in (1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(B, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == B
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    import dto_amd
    from dto_amd.parallel import gather_trajectories, shard_range
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B, nz = 7, 5                                  # uneven shards: 3 and 4 instances
    lo, hi = shard_range(B, rank, world)
    z = torch.tensor([[100.0 * b + i for i in range(nz)] for b in range(lo, hi)], dtype=torch.float64)
    status = torch.tensor([float(b % 3) for b in range(lo, hi)], dtype=torch.float64)
    out = gather_trajectories(z, status, dist)
    exp = np.array([[100.0 * b + i for i in range(nz)] + [float(b % 3)] for b in range(B)])
    assert out.shape == (B, nz + 1), out.shape
    assert np.array_equal(out.numpy(), exp)
    # bounded-memory forms: a receive budget of two rows per collective -> chunks; with a sink nothing is assembled
    got = np.full((B, nz + 1), np.nan)
    def sink(g0, rows):
        got[g0:g0 + rows.shape[0]] = rows.numpy()
    n = gather_trajectories(z, status, dist, sink=sink, max_bytes=2 * world * (nz + 1) * 8)
    assert n == B and np.array_equal(got, exp)
    out2 = gather_trajectories(z, status, dist, max_bytes=world * (nz + 1) * 8)      # one row per collective, assembled on the host
    assert out2.shape == (B, nz + 1) and np.array_equal(out2.numpy(), exp)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_all_gather_of_trajectories_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


def test_single_process_gather_is_identity():
    import torch
    from dto_amd.parallel import gather_trajectories
    z = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    st = torch.tensor([1.0, 0.0, 2.0], dtype=torch.float64)
    out = gather_trajectories(z, st, None)
    assert out.shape == (3, 5) and torch.equal(out[:, :4], z) and torch.equal(out[:, 4], st)


def test_chunked_guess_generation_is_the_same_stream():
    """bench.py builds the guesses of its (memory-sized) batch on the device 32 768 instances at a time; the chunks continue
    one seeded stream, so the batch is bit-identical to generating it in one piece (and a rank's instances do not depend on
    the chunk size)."""
    import numpy as np
    import torch
    from conftest import product_solver
    from bench import make_guesses, make_guesses_device
    s, p = product_solver("acrobot", 12)
    whole = make_guesses(s, p, 23, seed=5)
    for chunk in (1, 7, 23, 100):
        got = make_guesses_device(s, p, 23, 5, torch.device("cpu"), chunk=chunk).numpy()
        assert np.array_equal(got, whole), chunk


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts two ranks as child processes
    (torch.distributed.run, 127.0.0.1) without importing torch itself; rank 0 prints ONE line with n_gpus = 2.  The ranks run
    the launcher self-test (gloo all-reduce on the CPU) instead of the GPU workload."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DTO_BENCH_PARENT_CHECK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["launcher_selftest"] is True and d["n_gpus"] == 2 and d["rank_sum"] == 1.0
    assert "parent imported torch: False" in out.stderr
