"""configs[1] (LM n=50000 p=10000, k = 1..200) as C chunk chains side by side on one GPU (BESSX_KPATH_CHAINS = C) against
the single chain: ms per path, and the path compared candidate by candidate.   python tools/kchunks_bench.py C [C ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

n, p, kmax = 50000, 10000, 200
X, y, _, _ = synth.make_lm(n, p, 100)
seq = np.arange(1, kmax + 1)
single = None
for C in [1] + [int(v) for v in sys.argv[1:]]:
    os.environ["BESSX_KPATH_CHAINS"] = str(C)
    with capi.Session(X, y) as s:
        for _ in range(3):
            out = s.sequential_path(seq, ic_type=3)
        ts = []
        c0 = s.counters()  # (the phase timers and fill counts below are those of the ten timed paths only)
        for _ in range(10):
            t0 = time.time()
            out = s.sequential_path(seq, ic_type=3)
            ts.append(time.time() - t0)
        cnt = {k: v - c0[k] if k not in ("kpath_chains_last_path",) else v for k, v in s.counters().items()}
    if single is None:
        single = out
    rec = {"chains": C, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "ms_per_path_min": round(1e3 * min(ts), 2),
           "ms_per_path_median": round(1e3 * float(np.median(ts)), 2), "candidates_per_s": round(kmax / float(np.median(ts)), 0),
           "supports_equal": int(np.sum([np.array_equal(out["cand_support"][k], single["cand_support"][k]) for k in range(kmax)])),
           "iterations_equal": int(np.sum(out["cand_iters"] == single["cand_iters"])), "of": kmax,
           "max_rel_ic_diff": float(np.max(np.abs(out["cand_ic"] - single["cand_ic"]) / np.abs(single["cand_ic"]))),
           "pdas_iterations": int(out["n_pdas_iters"]), "best_k": int(out["best_T0"]),
           "passes_over_X_per_path": cnt["passes_over_X"] / 10.0, "stitch_refits_per_path": cnt["kpath_stitch_refits"] / 10.0,
           "chunk_fills_per_path": cnt["kpath_chunk_fills"] / 10.0,
           "phase_ms_per_path": {"coarse": round(cnt["kpath_coarse_us"] / 10e3, 2), "chunks": round(cnt["kpath_chunks_us"] / 10e3, 2),
                                 "stitch": round(cnt["kpath_stitch_us"] / 10e3, 2)},
           "merged_chunk_phases": cnt["kpath_merged_chunk_phases"], "chains_taken_over": cnt["kpath_chains_taken_over"]}
    print(json.dumps(rec), flush=True)
