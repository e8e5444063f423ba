"""How many factorisation rounds a 64-instance tile needs per iteration (= the largest per-instance count of the tile, the
lanes run in lockstep) vs. the per-instance mean -- CPU experiment with the oracle's C port:
python tools/port_tile_rounds.py T iters [model]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.cpu_port import PortSolver, guesses

T, iters = int(sys.argv[1]), int(sys.argv[2])
model = sys.argv[3] if len(sys.argv) > 3 else "acrobot"
Z, _, _ = guesses(model, T, 64, 1000)
S = [PortSolver(model, T, max_iter=100000) for _ in range(64)]
for s, z in zip(S, Z):
    s.begin(z)
prev = np.zeros(64)
rounds, mean = [], []
hist = np.zeros(12)
for it in range(iters):
    for s in S:
        s.iterate()
    nf = np.array([s.nfact for s in S], dtype=float)
    d = nf - prev
    prev = nf
    rounds.append(d.max()); mean.append(d.mean())
    for v in d:
        hist[int(min(v, 11))] += 1
print(f"{model} T={T}: iterations 0..{iters}: factorisations/instance/iteration {np.mean(mean):.2f}, tile rounds/iteration {np.mean(rounds):.2f}")
for a, b in ((0, 25), (25, 100), (100, iters)):
    if b <= iters:
        print(f"  iterations {a}-{b}: per instance {np.mean(mean[a:b]):.2f}, tile rounds {np.mean(rounds[a:b]):.2f}")
print("  histogram of attempts per instance-iteration:", (hist / hist.sum()).round(3).tolist())
