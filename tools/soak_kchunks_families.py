"""Soak of the chunk chains of the IRLS / Newton families at full size: N chunked paths, each compared with the single
chain candidate by candidate (supports, iteration counts, criteria).   python tools/soak_kchunks_families.py fam [N] [C]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

fam = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
C = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if fam == "logistic":
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=2), 100
elif fam == "poisson":
    X, y, _, _ = synth.make_poisson(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=3), 100
else:
    X, _, y, _, _ = synth.make_cox(int(os.environ.get("COX_N", 200000)), int(os.environ.get("COX_P", 20000)), 75)
    kw, kmax = dict(data_type=3, model_type=4), 150
seq = np.arange(1, kmax + 1)
bad = 0
t0 = time.time()
with capi.Session(X, y, **kw) as s:
    del X
    s.set_kpath_chains(1)
    single = s.sequential_path(seq, ic_type=3)
    for i in range(N):
        s.set_kpath_chains(C if i % 4 else 2)
        m = kmax if i % 5 else kmax - 7
        out = s.sequential_path(seq[:m], ic_type=3)
        ok = (np.array_equal(out["cand_support"][:, :m], single["cand_support"][:m, :m]) and
              np.array_equal(out["cand_iters"], single["cand_iters"][:m]) and
              np.allclose(out["cand_ic"], single["cand_ic"][:m], rtol=1e-9) and
              np.allclose(out["cand_beta"][:, :m], single["cand_beta"][:m, :m], rtol=1e-6, atol=1e-9))
        bad += 0 if ok else 1
        if not ok:
            print("path %d differs" % i, flush=True)
        if i % 10 == 9:
            print("%d paths, %d differ, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    print(s.counters())
print("soak done: %s, %d paths, %d differ" % (fam, N, bad))
sys.exit(1 if bad else 0)
