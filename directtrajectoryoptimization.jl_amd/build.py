"""In-tree build of the runtime library (libdto_hip.so) with hipcc for gfx950.

The library is built next to this file so it travels with the repo snapshot to the GPU box;
nothing is installed into site-packages.  Rebuilds only when the content hash of the sources changes.
"""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libdto_hip.so")
SOURCES = ["dto_core.cpp", "dto_solver.cpp"]


def _digest() -> str:
    """Content hash of every source the runtime depends on (mtimes do not survive the snapshot copy)."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    deps.append(os.path.join(_HERE, "..", "include", "dto.h"))
    deps.append(os.path.abspath(__file__))     # the compiler flags are part of what the library is
    for d in deps:
        if os.path.isfile(d):
            with open(d, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    stamp = LIB + ".srchash"
    if not (os.path.exists(LIB) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest()


def build_runtime(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB + f".tmp{os.getpid()}"
    # (-amdgpu-remove-redundant-endcf=0: see plugin.py BASE_CXXFLAGS -- every device build keeps the exec restores)
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wall",
           "-mllvm", "-amdgpu-remove-redundant-endcf=0", "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed building libdto_hip.so:\n" + res.stderr[-4000:])
    os.replace(tmp, LIB)
    with open(LIB + ".srchash", "w") as f:
        f.write(_digest())
    return LIB
