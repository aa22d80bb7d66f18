"""Soak of the chunk chains' staged fills: the same path again and again on one session, on designs whose chunks park
for many fills of their own, with host threads burning cycles beside it (the timing between the chains' host threads
is what varies); every path must be the single chain's.   python tools/soak_staged.py [reps] [load threads]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nload = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stop = False


def burn():
    x = 1.0
    while not stop:
        for _ in range(20000):
            x = x * 1.0000001 + 1e-9
        time.sleep(0.0002)


th = [threading.Thread(target=burn, daemon=True) for _ in range(nload)]
for t in th:
    t.start()
bad = 0
report = []
for (n, p, k, seed, top, chains) in ((3000, 2600, 60, 3, 160, 8), (3000, 1500, 60, 5, 128, 4), (50000, 10000, 100, 0, 200, 4)):
    X, y, _, _ = synth.make_lm(n, p, k, seed=seed) if seed else synth.make_lm(n, p, k)
    seq = np.arange(1, top + 1)
    os.environ["BESSX_KPATH_CHAINS"] = "1"
    with capi.Session(X, y) as s:
        want = s.sequential_path(seq, ic_type=3)
    os.environ["BESSX_KPATH_CHAINS"] = str(chains)
    t0 = time.time()
    fills = 0
    with capi.Session(X, y) as s:
        for r in range(reps):
            got = s.sequential_path(seq, ic_type=3)
            same = (np.array_equal(got["cand_support"], want["cand_support"]) and np.array_equal(got["cand_iters"], want["cand_iters"])
                    and np.allclose(got["cand_ic"], want["cand_ic"], rtol=1e-9, atol=0) and got["best_T0"] == want["best_T0"])
            if not same:
                bad += 1
                print("MISMATCH design", (n, p), "rep", r, flush=True)
        fills = s.counters()["kpath_chunk_fills"]
    report.append({"design": [n, p, top], "chains": chains, "paths": reps, "chunk_fills": int(fills), "seconds": round(time.time() - t0, 1)})
    print(report[-1], flush=True)
stop = True
print(json.dumps({"soak": "staged fills", "load_threads": nload, "mismatches": bad, "runs": report}))
sys.exit(1 if bad else 0)
