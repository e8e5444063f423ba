"""A plain C99 host program (tests/c_abi/drive_solve.c) drives a solve purely through include/dto.h and libdto_hip.so --
no Python in the loop -- and gets bit for bit what the Python mirror gets over ctypes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, product_solver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,T", [("pendulum", 50), ("car", 51)])
def test_c_program_solves_through_the_header_only(model, T, tmp_path):
    import dto_amd
    from dto_amd import build
    s, p = product_solver(model, T)
    xs, us = p["guess"](np.random.Generator(np.random.PCG64(0)))
    dto_amd.initialize_states(s, xs)
    dto_amd.initialize_controls(s, us)
    assert dto_amd.solve(s) == 1
    n = s._solve_nlp
    lo, hi = n.variable_bounds
    prob = tmp_path / "problem.txt"
    with open(prob, "w") as f:
        f.write(f"{n.plugin_path}\n{n.T}\n" + " ".join(str(k) for k in n.structure.stage_kind) + f"\n{n.num_variables}\n")
        for arr in (lo, hi, s._z0):
            f.write(" ".join(repr(float(v)) for v in arr) + "\n")
    exe = tmp_path / "drive_solve"
    lib = build.LIB
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "drive_solve.c"), "-o", str(exe), lib, "-ldl",
                    f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    sol = tmp_path / "solution.bin"
    res = subprocess.run([str(exe), str(prob), str(sol)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    words = res.stdout.split()
    assert int(words[1]) == 1 and int(words[3]) == s.iterations
    data = np.fromfile(sol, dtype=np.float64)
    assert data.size == n.num_variables + n.num_constraint
    assert np.array_equal(data[:n.num_variables], s._solution)            # same kernels, same inputs: identical bits
    assert abs(float(words[5]) - s.nlp.eval_objective(s._solution)) == 0.0
