"""Every rank's link of the N-rank Cox k-path (configs[4]; levels cold-started, automatic chain count, then one stitch refit
from the true predecessor model) timed alone on one GPU: slowest rank vs the one-GPU path.  python tools/cox_kpath_probe.py [N ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth, dist as bdist  # noqa: E402
worlds = [int(v) for v in sys.argv[1:]] or [8]
X, _, st, _, _ = synth.make_cox(200000, 20000, 75)
seq = np.arange(1, 151)
n = X.shape[0]
with capi.Session(X, st, data_type=3, model_type=4) as s:
    xm, xn, ym = s.normalization()
    del X
    single = s.sequential_path(seq, ic_type=3)
    t0 = time.time()
    single = s.sequential_path(seq, ic_type=3)
    one = time.time() - t0
    print(json.dumps({"one_gpu_ms_per_path": round(1e3 * one, 1)}), flush=True)
    rounds = int(os.environ.get("BESSX_PROBE_REBALANCE", "0"))  # > 0: as many rebalancing steps (bdist.rebalance_bounds)
    for world, it in [(w, i) for w in worlds for i in range(rounds + 1)]:
        if it == 0:
            bounds = [bdist.partition(len(seq), world, r)[0] for r in range(world)] + [len(seq)]
        rows = []
        for rank in range(world):
            lo, hi = bounds[rank], bounds[rank + 1]
            best = None
            for rep in range(2):
                s.set_kpath_chains(0)  # (a rank has a session of its own: the automatic choice afresh)
                t0 = time.time()
                out = s.sequential_path_chain(seq[lo:hi], ic_type=3)
                t_chunk = time.time() - t0
                t_st, refits = 0.0, 0
                if rank > 0:
                    sup = single["cand_support"][lo - 1][:lo]
                    val = single["cand_beta"][lo - 1][:lo] * xn[sup] / np.sqrt(float(n))
                    t0 = time.time()
                    res = s.sequential_path_chain(seq[lo:hi], ic_type=3, init_idx=sup, init_val=val, keep_caches=True,
                                                  stop_support=out["cand_support"], stop_beta=out["cand_beta"])
                    t_st = time.time() - t0
                    refits = int(res["n_candidates"])
                if best is None or t_chunk + t_st < best[0] + best[1]:
                    best = (t_chunk, t_st, refits)
            cnt = s.counters()
            rows.append({"rank": rank, "levels": [int(seq[lo]), int(seq[hi - 1])], "chunk_ms": round(1e3 * best[0], 1),
                         "stitch_ms": round(1e3 * best[1], 1), "refits": best[2]})
        slow = max(r["chunk_ms"] + r["stitch_ms"] for r in rows)
        print(json.dumps({"world": world, "rebalancing_step": it, "bounds": [int(v) for v in bounds],
                          "slowest_rank_ms": round(slow, 1), "speedup_over_one_gpu": round(1e3 * one / slow, 2),
                          "ranks": rows}), flush=True)
        bounds = bdist.rebalance_bounds(bounds, [1e-3 * (r["chunk_ms"] + r["stitch_ms"]) for r in rows])
