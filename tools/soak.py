"""Soak test of the chained / fused covariance path: the same 200-candidate path over and over on one session,
every repetition must reproduce the first one bit for bit (supports, coefficients, ICs, iteration counts).
  python tools/soak.py [repetitions] [n p kmax]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bess_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, p, kmax = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (50000, 10000, 200)
X, y, _, _ = synth.make_lm(n, p, min(100, p // 4))
seq = np.arange(1, kmax + 1)
with capi.Session(X, y) as s:
    base = s.sequential_path(seq, ic_type=3)
    t0 = time.time()
    for r in range(reps):
        o = s.sequential_path(seq, ic_type=3)
        for k in ("cand_support", "cand_beta", "cand_ic", "cand_iters", "cand_train_loss"):
            assert np.array_equal(o[k], base[k]), "repetition %d: %s differs" % (r, k)
        assert o["best_T0"] == base["best_T0"] and np.array_equal(o["beta"], base["beta"])
        if r % 50 == 49:
            print("rep", r + 1, "ok, %.2f ms per path" % (1e3 * (time.time() - t0) / (r + 1)), flush=True)
    print("soak OK: %d repetitions identical; counters %r" % (reps, s.counters()))
