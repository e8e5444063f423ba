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
