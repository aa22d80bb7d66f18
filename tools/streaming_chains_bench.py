"""configs[1] in the STREAMING form of the score pass (score_mode = 1) as C chunk chains against the single chain.
   python tools/streaming_chains_bench.py C [C ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402

X, y, _, _ = synth.make_lm(50000, 10000, 100)
seq = np.arange(1, 201)
single = None
for C in [1] + [int(v) for v in sys.argv[1:]]:
    os.environ["BESSX_KPATH_CHAINS"] = str(C)
    with capi.Session(X, y, score_mode=1) as s:
        out = s.sequential_path(seq, ic_type=3)
        s.enable_kernel_timing(True)
        s.score_pass_stats(reset=True)
        ts = []
        for _ in range(3):
            t0 = time.time()
            out = s.sequential_path(seq, ic_type=3)
            ts.append(time.time() - t0)
        st = s.score_pass_stats()
        cnt = s.counters()
    if single is None:
        single = out
    per_pass = st["seconds"] / max(st["launches"], 1)
    print(json.dumps({"chains": C, "ms_per_path_min": round(1e3 * min(ts), 1), "candidates_per_s": round(200 / min(ts), 1),
                      "passes_over_X_per_path": st["launches"] / 3.0, "k_xtv_ms_per_pass": round(1e3 * per_pass, 4),
                      "k_xtv_frac_of_hbm": round(4e9 / per_pass / 8e12, 3),
                      "whole_path_frac_of_hbm": round(st["launches"] / 3.0 * 4e9 / min(ts) / 8e12, 3),
                      "supports_equal": int(np.sum([np.array_equal(out["cand_support"][k], single["cand_support"][k]) for k in range(200)])),
                      "iterations_equal": int(np.sum(out["cand_iters"] == single["cand_iters"])),
                      "stitch_refits": cnt["kpath_stitch_refits"], "chains_last_path": cnt["kpath_chains_last_path"],
                      "shared_pass_launches": cnt["shared_pass_launches"], "shared_pass_chain_slots": cnt["shared_pass_chain_slots"],
                      "shared_pass_partial_batches": cnt["shared_pass_partial_batches"]}), flush=True)
