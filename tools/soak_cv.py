"""Soak test of the fold fits that run side by side (cross-validation): the same cross-validated
golden-section and sequential paths over and over -- on one session, and on fresh sessions (thread pool, fold contexts
and streams created and torn down) -- every repetition must reproduce the first one bit for bit.
  python tools/soak_cv.py [repetitions] [n p smax K]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bess_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, p, smax, K = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (50000, 10000, 200, 5)
X, y, _, _ = synth.make_lm(n, p, min(100, p // 4))
fold = synth.make_cv_folds(n, K)
KEYS = ("cand_support", "cand_beta", "cand_ic", "cand_iters", "cand_train_loss", "beta")


def same(o, base, what):
    for k in KEYS:
        assert np.array_equal(o[k], base[k]), "%s: %s differs" % (what, k)
    assert o["best_T0"] == base["best_T0"] and o["n_fits"] == base["n_fits"] and o["n_pdas_iters"] == base["n_pdas_iters"]


base = None
t0 = time.time()
for sess in range(3):
    with capi.Session(X, y) as s:
        s.set_cv(K, fold)
        for r in range(reps):
            o = (s.gs_path(1, smax, ic_type=3, is_cv=True), s.sequential_path(np.arange(1, 13), [0.0, 0.02], ic_type=3, is_cv=True))
            if base is None:
                base = o
            same(o[0], base[0], "session %d repetition %d gs_path" % (sess, r))
            same(o[1], base[1], "session %d repetition %d sequential_path" % (sess, r))
        c = s.counters()
        print("session", sess, "ok:", reps, "repetitions, %.1f s so far; rounds %d union fills %d" %
              (time.time() - t0, c["cv_side_by_side_rounds"], c["cv_union_fills"]), flush=True)
print("soak OK")
