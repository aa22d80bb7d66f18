"""One rank's link of the 8-rank Cox k-path (configs[4], levels lo+1..hi cold-started) timed alone with 1 / 2 / 3 chunk
chains (shared passes).   python tools/cox_link_probe.py [rank world]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth, dist as bdist  # noqa: E402
rank, world = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 8)
X, _, st, _, _ = synth.make_cox(200000, 20000, 75)
seq = np.arange(1, 151)
lo, hi = bdist.partition(len(seq), world, rank)
with capi.Session(X, st, data_type=3, model_type=4) as s:
    del X
    base = None
    for C in (0, 1, 3):
        s.set_kpath_chains(C)
        s.sequential_path_chain(seq[lo:hi], ic_type=3)
        ts = []
        for _ in range(2):
            t0 = time.time()
            out = s.sequential_path_chain(seq[lo:hi], ic_type=3)
            ts.append(time.time() - t0)
        base = base or out
        cnt = s.counters()
        print(json.dumps({"rank": rank, "world": world, "levels": [int(seq[lo]), int(seq[hi - 1])], "chains_asked": C,
                          "chains_run": cnt["kpath_chains_last_path"] if cnt["kpath_chunked_paths"] else 1,
                          "ms": round(1e3 * min(ts), 1),
                          "same": bool(np.array_equal(out["cand_support"], base["cand_support"]))}), flush=True)
