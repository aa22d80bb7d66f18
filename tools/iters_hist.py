"""configs[1]: how many PDAS iterations the candidates take, and what the chained fits did (BESSX_DEBUG counters)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
with capi.Session(X, y) as s:
    s.set_kpath_chains(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    for _ in range(3):
        out = s.sequential_path(seq, ic_type=3)
    it = np.asarray(out["cand_iters"])
    print("iterations histogram:", dict(zip(*[a.tolist() for a in np.unique(it, return_counts=True)])))
    for lo in range(0, 200, 50):
        print("k %3d..%3d: " % (lo + 1, lo + 50), dict(zip(*[a.tolist() for a in np.unique(it[lo:lo + 50], return_counts=True)])))
    print(s.counters())
