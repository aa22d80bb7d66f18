"""configs[1]: the Gram column cache after a path, 32-slot group by group (= pass over X by pass): how many of a
group's columns were ever active, and at which level each is first needed"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
for chains in (1, 4):
    with capi.Session(X, y) as s:
        s.set_kpath_chains(chains)
        s.sequential_path(seq, ic_type=3)
        out = s.sequential_path(seq, ic_type=3)
        bd, slot = s.cov_state()
    first = {}
    for k in range(200):
        for c in out["cand_support"][k]:
            if c >= 0 and int(c) not in first:
                first[int(c)] = k + 1
    order = np.argsort(np.where(slot >= 0, slot, 1 << 30))
    ncached = int(np.sum(slot >= 0))
    print("chains", chains, "cached", ncached)
    for g in range((ncached + 31) // 32):
        cols = order[32 * g:min(32 * g + 32, ncached)]
        used = [first.get(int(c), 0) for c in cols]
        print("  group %2d: %2d of %2d ever active; first needed at level: %s" % (g, sum(1 for u in used if u), len(cols), " ".join("%3d" % u if u else "  ." for u in used)))
