"""One session of a family at full size, the sequential path timed with 1..C chunk chains (bessx_session_set_kpath_chains).
   python tools/kchunks_family_sweep.py logistic|poisson|cox C [C ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

fam = sys.argv[1]
if fam == "logistic":
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=2), 100
elif fam == "poisson":
    X, y, _, _ = synth.make_poisson(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=3), 100
else:
    X, _, y, _, _ = synth.make_cox(200000, 20000, 75)
    kw, kmax = dict(data_type=3, model_type=4), 150
seq = np.arange(1, kmax + 1)
with capi.Session(X, y, **kw) as s:
    del X
    base = None
    for C in [1] + [int(v) for v in sys.argv[2:]]:
        s.set_kpath_chains(C)
        s.sequential_path(seq, ic_type=3)
        ts = []
        for _ in range(2 if fam == "cox" else 4):
            t0 = time.time()
            out = s.sequential_path(seq, ic_type=3)
            ts.append(time.time() - t0)
        base = base or out
        print(json.dumps({"family": fam, "chains": C, "ms_per_path": round(1e3 * min(ts), 1),
                          "candidates_per_s": round(kmax / min(ts), 1),
                          "same": bool(np.array_equal(out["cand_support"], base["cand_support"]) and
                                       np.array_equal(out["cand_iters"], base["cand_iters"]))}), flush=True)
