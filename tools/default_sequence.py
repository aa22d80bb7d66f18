"""The reference's DEFAULT call, sequence = 1..min(p, n / log n) (python/bess/linear.py:285-287), at n = 25000, p = 3000:
2468 candidates, sparsity levels far beyond the register-resident solvers.  Prints seconds, candidates/s and the
session counters; under rocprofv3 --kernel-trace --stats it is the workload of profiles/r04_default_sequence_*.csv.

    python tools/default_sequence.py [n p [kmax]]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else min(p, int(n / np.log(n)))
X, y, support, _ = synth.make_lm(n, p, 12, seed=9)
seq = np.arange(1, kmax + 1)
with capi.Session(X, y, max_sparsity=kmax) as s:
    s.sequential_path(seq[:40], ic_type=4)  # code objects
    s.enable_kernel_timing(True)
    s.score_pass_stats(reset=True)
    t0 = time.time()
    out = s.sequential_path(seq, ic_type=4)
    dt = time.time() - t0
    k1 = s.score_pass_stats()
    cnt = s.counters()
ok = bool(np.array_equal(np.nonzero(out["beta"])[0], support))
print(json.dumps({"n": n, "p": p, "candidates": int(kmax), "seconds": dt, "candidates_per_s": kmax / dt,
                  "pdas_iterations": int(out["n_pdas_iters"]), "passes_over_X": k1["algorithmic_bytes"] / (8.0 * n * p),
                  "seconds_in_the_kernel_that_streams_X": k1["seconds"], "selected_k": int(out["best_T0"]),
                  "selects_the_planted_support": ok, "counters": cnt}))
