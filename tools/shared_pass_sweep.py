"""Chains x groups sweep of the shared passes.  python tools/shared_pass_sweep.py lm|logistic|poisson|cox "C:G[:lendiv]" ..."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402
fam = sys.argv[1]
if fam == "lm":
    X, y, _, _ = synth.make_lm(50000, 10000, 100)
    kw, kmax = dict(score_mode=1), 200
elif fam == "logistic":
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=2), 100
elif fam == "poisson":
    X, y, _, _ = synth.make_poisson(100000, 5000, 50)
    kw, kmax = dict(data_type=2, model_type=3), 100
else:
    X, _, y, _, _ = synth.make_cox(200000, 20000, 75)
    kw, kmax = dict(data_type=3, model_type=4), 150
seq = np.arange(1, kmax + 1)
base = None
with capi.Session(X, y, **kw) as s:
    del X
    for spec in ["1:1"] + sys.argv[2:]:
        f = spec.split(":")
        C, G = int(f[0]), int(f[1])
        hooks = "kchunks_pass_groups=%d" % G + (",kchunks_len_div=%s" % f[2] if len(f) > 2 else "") + (",kchunks_shared_pass=0" if G == 0 else "")
        os.environ["BESSX_TEST_HOOKS"] = hooks
        s.set_kpath_chains(C)
        s.sequential_path(seq, ic_type=3)
        s.enable_kernel_timing(True)
        s.score_pass_stats(reset=True)
        c0 = s.counters()
        reps = 2 if fam == "cox" else 4
        ts = []
        for _ in range(reps):
            t0 = time.time()
            out = s.sequential_path(seq, ic_type=3)
            ts.append(time.time() - t0)
        base = base or out
        st = s.score_pass_stats()
        c1 = s.counters()
        s.enable_kernel_timing(False)
        print(json.dumps({"family": fam, "chains": C, "groups": G, "spec": spec, "ms_per_path": round(1e3 * min(ts), 1),
                          "candidates_per_s": round(kmax / min(ts), 1), "passes_per_path": st["launches"] / float(reps),
                          "ms_per_pass": round(1e3 * st["seconds"] / max(st["launches"], 1), 4),
                          "chain_slots_per_path": (c1["shared_pass_chain_slots"] - c0["shared_pass_chain_slots"]) / float(reps),
                          "partial_batches": c1["shared_pass_partial_batches"] - c0["shared_pass_partial_batches"],
                          "stitch_refits": (c1["kpath_stitch_refits"] - c0["kpath_stitch_refits"]) / float(reps),
                          "same": bool(np.array_equal(out["cand_support"], base["cand_support"]) and
                                       np.array_equal(out["cand_iters"], base["cand_iters"]) and
                                       np.allclose(out["cand_ic"], base["cand_ic"], rtol=1e-9, atol=0))}), flush=True)
