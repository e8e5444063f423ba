"""The C-ABI library loads and exports every symbol include/dto.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT

import dto_amd
from dto_amd import capi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dto.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dto_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = capi.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libdto_hip.so does not export {n}"


def test_compute_entry_points_fail_loudly_without_gpu():
    n = ctypes.c_int(-1)
    capi.check(capi.lib().dto_device_count(ctypes.byref(n)))
    if n.value > 0:
        return  # on the GPU box the compute path is exercised by the -m gpu tests
    import numpy as np
    import pytest
    from conftest import product_solver
    s, _ = product_solver("pendulum", 6)
    with pytest.raises(capi.DtoError) as e:
        s.nlp.eval_objective(np.zeros(s.nlp.num_variables))
    assert e.value.code == 3  # DTO_ERR_DEVICE: there is no CPU fallback


def test_header_compiles_as_c99_and_structs_match_the_ctypes_mirror():
    """include/dto.h is plain C (no C++ needed by a host language); the ctypes structures have the C compiler's layout."""
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "dto.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(dto_problem_spec), sizeof(dto_options), sizeof(dto_batch), sizeof(dto_kkt_system), sizeof(dto_sizes_t));
  printf("%zu %zu %zu %zu %zu\n", offsetof(dto_problem_spec, evaluate_hessian), offsetof(dto_options, mu_target), offsetof(dto_kkt_system, delta_c),
         offsetof(dto_options, hessian_approximation), offsetof(dto_options, kkt_refinement));
  return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    sizes = [int(v) for v in out]
    assert sizes[:5] == [ctypes.sizeof(capi.ProblemSpec), ctypes.sizeof(capi.COptions), ctypes.sizeof(capi.Batch),
                         ctypes.sizeof(capi.KktSystem), ctypes.sizeof(capi.Sizes)]
    assert sizes[5:] == [capi.ProblemSpec.evaluate_hessian.offset, capi.COptions.mu_target.offset, capi.KktSystem.delta_c.offset,
                         capi.COptions.hessian_approximation.offset, capi.COptions.kkt_refinement.offset]   # ABI 3 / ABI 4 fields


def test_shard_range_on_the_c_abi():
    """dto_shard_range: contiguous blocks in rank order that cover every instance exactly once."""
    from dto_amd.parallel import shard_range
    import pytest
    for total, world in ((512, 8), (10, 4), (3, 8), (0, 2), (131072, 7)):
        seen = []
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            assert lo == (total * r) // world and hi == (total * (r + 1)) // world
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    with pytest.raises(capi.DtoError):
        shard_range(8, 3, 2)
