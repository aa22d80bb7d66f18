"""Chunk chains where they do NOT pay: a design of correlated columns with a weak signal (n = 4000, p = 2400, 160 levels)
on which the chunks' own chains do not merge with the warm chain -- ms per path for 1 / 2 / 4 / 8 chains and the stitch's
refits (bounded by its budget: the rest of the path is then walked as one chain).   python tools/kchunks_hard_design.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi
rng = np.random.default_rng(3)
n, p = 4000, 2400
Z = rng.standard_normal((n, p)); X = Z.copy()
for j in range(1, p): X[:, j] = 0.6 * X[:, j - 1] + 0.8 * Z[:, j]
beta = np.zeros(p); beta[rng.choice(p, 60, replace=False)] = rng.uniform(0.2, 1.0, 60) * rng.choice([-1.0, 1.0], 60)
y = X @ beta + 2.0 * rng.standard_normal(n)
seq = np.arange(1, 161)
with capi.Session(X, y) as s:
    for C in (1, 2, 4, 8):
        s.set_kpath_chains(C)
        s.sequential_path(seq, ic_type=3)
        c0 = s.counters()
        t0 = time.time()
        for _ in range(10): s.sequential_path(seq, ic_type=3)
        dt = (time.time() - t0) / 10
        c1 = s.counters()
        print(C, round(1e3 * dt, 2), "ms per path; refits per path", (c1["kpath_stitch_refits"] - c0["kpath_stitch_refits"]) / 10.0, flush=True)
