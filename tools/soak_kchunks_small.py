"""Rendezvous stress of the chunk chains: many chains on a mid-size design whose coarse chain leaves columns uncached, so
chunk chains park and fill the shared cache while the others stand still; every path against the single chain.
   python tools/soak_kchunks_small.py [N] [C]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(3)
n, p = 4000, 2400
Z = rng.standard_normal((n, p))
X = Z.copy()
for j in range(1, p):  # correlated columns: the chains wander through noise columns the coarse chain never met
    X[:, j] = 0.6 * X[:, j - 1] + 0.8 * Z[:, j]
beta = np.zeros(p)
beta[rng.choice(p, 60, replace=False)] = rng.uniform(0.2, 1.0, 60) * rng.choice([-1.0, 1.0], 60)
y = X @ beta + 2.0 * rng.standard_normal(n)
seq = np.arange(1, 161)
bad = 0
t0 = time.time()
with capi.Session(X, y) as s:
    s.set_kpath_chains(1)
    single = s.sequential_path(seq, ic_type=3)
    for i in range(N):
        s.set_kpath_chains(C if i % 3 else max(2, C // 2))
        out = s.sequential_path(seq, ic_type=3)
        ok = (np.array_equal(out["cand_support"], single["cand_support"]) and
              np.array_equal(out["cand_iters"], single["cand_iters"]) and
              np.allclose(out["cand_ic"], single["cand_ic"], rtol=1e-9))
        bad += 0 if ok else 1
        if not ok:
            print("path %d differs" % i, flush=True)
        if i % 100 == 99:
            print("%d paths, %d differ, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    cnt = s.counters()
print({k: cnt[k] for k in ("kpath_chunked_paths", "kpath_stitch_refits", "kpath_chunk_fills", "passes_over_X")})
print("soak done: %d paths with up to %d chains, %d differ" % (N, C, bad))
sys.exit(1 if bad else 0)
