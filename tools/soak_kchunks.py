"""Soak of the chunk chains (bessx_kchunks.cpp): N paths of configs[1] with C chains, every one compared with the single
chain candidate by candidate (supports, iteration counts, criteria).   python tools/soak_kchunks.py [N] [C]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
os.environ["BESSX_KPATH_CHAINS"] = "1"
with capi.Session(X, y) as s:
    single = s.sequential_path(seq, ic_type=3)
os.environ["BESSX_KPATH_CHAINS"] = str(C)
bad = 0
t0 = time.time()
with capi.Session(X, y) as s:
    for i in range(N):
        out = s.sequential_path(seq if i % 7 else seq[:150 + i % 50], ic_type=3)
        m = out["n_candidates"]
        ok = (np.array_equal(out["cand_support"][:, :m], single["cand_support"][:m, :m]) and
              np.array_equal(out["cand_iters"], single["cand_iters"][:m]) and
              np.allclose(out["cand_ic"], single["cand_ic"][:m], rtol=1e-9))
        bad += 0 if ok else 1
        if not ok:
            print("path %d differs" % i, flush=True)
        if i % 50 == 49:
            print("%d paths, %d differ, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    print(s.counters())
print("soak done: %d paths with %d chains, %d differ" % (N, C, bad))
sys.exit(1 if bad else 0)
