import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth
X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
for chains in (1, 4):
    with capi.Session(X, y) as s:
        s.set_kpath_chains(chains)
        s.sequential_path(seq, ic_type=3)
        c0 = s.counters()["passes_over_X"]
        out = s.sequential_path(seq, ic_type=3)
        c1 = s.counters()["passes_over_X"]
        bd, slot = s.cov_state()
        used = np.unique(out["cand_support"][out["cand_support"] >= 0])
        cached = np.nonzero(slot >= 0)[0]
        print({"chains": chains, "groups_per_path": c1 - c0, "columns_computed": 32 * (c1 - c0), "columns_cached": int(cached.size),
               "columns_ever_active": int(used.size), "cached_but_never_active": int(np.setdiff1d(cached, used).size)})
