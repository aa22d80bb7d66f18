"""Soak of the shared passes: hundreds of chunked paths of the four streaming families on medium-sized
problems, a random number of chains each time (2..8; Cox 2..6), optionally under host load (BESSX_SOAK_LOAD=threads of
busy Python), every path compared with the single chain's candidate by candidate.   python tools/soak_shared_pass.py [paths]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

paths = int(sys.argv[1]) if len(sys.argv) > 1 else 200
load = int(os.environ.get("BESSX_SOAK_LOAD", "0"))
stop = False


def burn():
    x = 0
    while not stop:
        x = (x * 1103515245 + 12345) % (1 << 31)


for _ in range(load):
    threading.Thread(target=burn, daemon=True).start()
rng = np.random.default_rng(2026)
fams = {
    "lm": (synth.make_lm(6000, 1500, 30, seed=31)[:2], dict(score_mode=1), 128),
    "logistic": (synth.make_logistic(6000, 1200, 20, seed=32)[:2], dict(data_type=2, model_type=2), 72),
    "poisson": (synth.make_poisson(6000, 1200, 20, seed=33)[:2], dict(data_type=2, model_type=3), 64),
}
Xc, _, stc, _, _ = synth.make_cox(6000, 1200, 20, seed=34)
fams["cox"] = ((Xc, stc), dict(data_type=3, model_type=4), 64)
t_all = time.time()
for fam, ((X, y), kw, top) in fams.items():
    seq = np.arange(1, top + 1)
    with capi.Session(X, y, **kw) as s:
        s.set_kpath_chains(1)
        want = s.sequential_path(seq, ic_type=3)
        bad, partial, slots, launches = 0, 0, 0, 0
        t0 = time.time()
        for i in range(paths):
            C = int(rng.integers(2, 7 if fam == "cox" else 9))
            s.set_kpath_chains(C)
            c0 = s.counters()
            out = s.sequential_path(seq, ic_type=3)
            c1 = s.counters()
            ok = (np.array_equal(out["cand_support"], want["cand_support"]) and np.array_equal(out["cand_iters"], want["cand_iters"])
                  and np.allclose(out["cand_ic"], want["cand_ic"], rtol=1e-9, atol=0))
            bad += 0 if ok else 1
            partial += c1["shared_pass_partial_batches"] - c0["shared_pass_partial_batches"]
            slots += c1["shared_pass_chain_slots"] - c0["shared_pass_chain_slots"]
            launches += c1["shared_pass_launches"] - c0["shared_pass_launches"]
            if (i + 1) % 50 == 0:
                print(json.dumps({"family": fam, "paths": i + 1, "differing_paths": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
        print(json.dumps({"family": fam, "paths": paths, "differing_paths": bad, "shared_launches": launches,
                          "partial_batches": partial, "host_load_threads": load,
                          "giveups": s.counters()["kpath_stitch_giveups"], "seconds": round(time.time() - t0, 1)}), flush=True)
stop = True
print(json.dumps({"total_seconds": round(time.time() - t_all, 1)}))
