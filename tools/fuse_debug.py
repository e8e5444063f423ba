import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import product_solver
from test_fused_update_gpu import _guesses, _run, NAMES
s, p = product_solver("acrobot", 101)
Z = _guesses(s, p, 130, seed=3)
for calls in ([3], [5], [7], [7, 1], [7, 1, 4], [23]):
    a = _run(s, Z, False, calls); b = _run(s, Z, True, calls)
    msg = []
    for n in NAMES:
        if a[n].size and not np.array_equal(a[n], b[n]):
            d = np.abs(a[n] - b[n]); i = np.unravel_index(np.argmax(d), d.shape)
            msg.append(f"{n}: max {d.max():.3g} at {i} rows_diff {np.unique(np.nonzero(d)[0])[:8]} cols {np.unique(np.nonzero(d)[1])[:12]}")
    for k in a["stats"]:
        if not np.array_equal(a["stats"][k], b["stats"][k]):
            msg.append(f"stat {k} differs at {np.nonzero(a['stats'][k] != b['stats'][k])[0][:8]}")
    print(calls, "partitions", s.partitions(), "OK" if not msg else msg, flush=True)
import os
def scal(fused, calls):
    import torch
    os.environ["DTO_FUSE_UPDATE"] = "1" if fused else "0"
    d = torch.tensor(Z, device="cuda")
    s.begin_batch(d.data_ptr(), Z.shape[0], Z.shape[1])
    for n in calls:
        s.iterate_batch(n)
    return {k: s.scalar_batch(k) for k in ("f", "theta1", "theta_inf", "dinf", "compl", "e0", "logbar", "mu", "alpha")}
a = scal(False, [3]); b = scal(True, [3])
for k in a:
    d = a[k] != b[k]
    print(k, int(d.sum()), (np.abs(a[k] - b[k]) / np.maximum(np.abs(a[k]), 1e-300))[d][:5])
