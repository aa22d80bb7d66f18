"""What ONE rank of the N-rank k-path (bench.py --gpus N, default partition, round 6: lead fits) does, timed alone on the
one GPU of this box, for every rank of N = 2, 4, 8: the lead fits + its chunk as chunk chains (sequential_path_chain with
lead_levels), then one stitch refit from the true predecessor model (what the all-gather of the last models delivers).
No communication is timed (two small all-gathers per step on the real node).  Slowest rank vs the one-GPU path = the
speed-up the partition can reach.   python tools/kpath_lead_probe.py [N ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402
from bess_amd import dist as bdist  # noqa: E402

X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
worlds = [int(v) for v in sys.argv[1:]] or [2, 4, 8]
with capi.Session(X, y) as s:
    for _ in range(2):
        single = s.sequential_path(seq, ic_type=3)
    ts = []
    for _ in range(5):
        t0 = time.time()
        single = s.sequential_path(seq, ic_type=3)
        ts.append(time.time() - t0)
    one = min(ts)
    print(json.dumps({"one_gpu_ms_per_path": round(1e3 * one, 3)}), flush=True)
    xm, xn, ym = s.normalization()
    n = X.shape[0]
    for world in worlds:
        rows = []
        for rank in range(world):
            sk = bdist.StitchedKPath(None, seq, world, rank, coarse_lead=True)
            lv = sk.lead_levels()
            lo, hi = sk.lo, sk.hi
            best = None
            for rep in range(4):
                t0 = time.time()
                out = s.sequential_path_chain(seq[lo:hi], ic_type=3, lead_levels=lv)
                t_chunk = time.time() - t0
                t_st = 0.0
                refits = 0
                if rank > 0:
                    sup = single["cand_support"][lo - 1][:lo]
                    val = single["cand_beta"][lo - 1][:lo] * xn[sup] / np.sqrt(float(n))
                    t0 = time.time()
                    res = s.sequential_path_chain(seq[lo:hi], ic_type=3, init_idx=sup, init_val=val, keep_caches=True,
                                                  stop_support=out["cand_support"], stop_beta=out["cand_beta"])
                    t_st = time.time() - t0
                    refits = int(res["n_candidates"])
                if rep and (best is None or t_chunk + t_st < best[0] + best[1]):
                    best = (t_chunk, t_st, refits)
            same = bool(np.array_equal(out["cand_support"][:, :200], single["cand_support"][lo:hi, :200]) or refits > 0)
            rows.append({"rank": rank, "levels": [int(seq[lo]), int(seq[hi - 1])], "lead": [int(v) for v in lv],
                         "chunk_ms": round(1e3 * best[0], 3), "stitch_ms": round(1e3 * best[1], 3), "refits": best[2],
                         "chains": s.counters()["kpath_chains_last_path"], "ok": same})
        slow = max(r["chunk_ms"] + r["stitch_ms"] for r in rows)
        print(json.dumps({"world": world, "slowest_rank_ms": round(slow, 3), "speedup_over_one_gpu": round(1e3 * one / slow, 3),
                          "ranks": rows}), flush=True)
