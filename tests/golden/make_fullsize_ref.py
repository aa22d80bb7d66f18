"""Generate full-size golden vectors from the compiled reference (oracle/_ref/libbess_ref.so).

Run once in the build container (it needs /root/reference to have been compiled by
`make -C oracle ref`); takes hours of one CPU core and ~25 GB for config 2:

    python tests/golden/make_fullsize_ref.py lm        # BASELINE configs[1], k = 1..200
    python tests/golden/make_fullsize_ref.py logistic  # BASELINE configs[2], k = 1..100
    python tests/golden/make_fullsize_ref.py poisson   # the Poisson family at the configs[2] shape, k = 1..100
    python tests/golden/make_fullsize_ref.py lmcv      # BASELINE configs[3], gs_path on [1,200] + 5-fold CV
    python tests/golden/make_fullsize_ref.py cox 4000  # BASELINE configs[4] recipe at the largest n the
                                                       # reference's n x n risk-set matrix allows (p=2000, k=1..40)
    python tests/golden/make_fullsize_ref.py cox-port 20  # BASELINE configs[4] at FULL size (n=200000, p=20000), k = 1..20,
                                                       # by the plain-C oracle (oracle/bess_oracle.c, kind "port": the
                                                       # reference needs a 320 GB n x n matrix there, src/Algorithm.h:1386;
                                                       # the port is pinned against the compiled reference by
                                                       # tests/test_oracle_vs_reference.py).  Needs ~100 GB of host memory
                                                       # (X, its sorted copy, the oracle's column-major copy): run on the
                                                       # GPU box's host, ~10 minutes of one core

BESS_REF_PROGRESS=1 prints one line per fit; BESS_REF_BUDGET_S=<s> stops the reference at the first fit that
would start after the budget: the file then holds a PREFIX of the path (`truncated` = 1, no best model).

Output: tests/golden/fullsize_<name>.npz holding, for every candidate of the warm-start
chain, the active set of every PDAS iteration, the fitted coefficients, the loss and the
information criterion.  Inputs are NOT stored: they are regenerated bit-identically from
bess_amd/synth.py (numpy PCG64 with a fixed seed).
"""
import importlib.util
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth = _load("synth", os.path.join(ROOT, "bess_amd", "synth.py"))
ref = _load("ref_ctypes", os.path.join(ROOT, "oracle", "ref_ctypes.py"))


def pack(t, extra):
    fits = t["fits"]
    out = dict(extra)
    out["truncated"] = int(t.get("truncated", False))
    out["best_beta_idx"] = np.nonzero(t["beta"])[0].astype(np.int32)
    out["best_beta_val"] = t["beta"][np.nonzero(t["beta"])[0]]
    out["best_coef0"] = t["coef0"]
    out["best_train_loss"] = t["train_loss"]
    out["best_ic"] = t["ic"]
    out["loss_calls"] = t["loss_calls"]
    out["ic_calls"] = t["ic_calls"]
    out["fit_T0"] = np.array([f["T0"] for f in fits], dtype=np.int32)
    out["fit_train_n"] = np.array([f["train_n"] for f in fits], dtype=np.int32)
    out["fit_iters"] = np.array([len(f["iters"]) for f in fits], dtype=np.int32)
    out["A_flat"] = np.concatenate([a for f in fits for a in f["iters"]]).astype(np.int32)
    out["beta_flat"] = np.concatenate([b for f in fits for b in f["betas"]])
    out["coef0_flat"] = np.array([c for f in fits for c in f["coef0s"]])
    return out


def main():
    which = sys.argv[1]
    kmax = int(sys.argv[2]) if len(sys.argv) > 2 else None
    small = os.environ.get("BESS_GOLDEN_SMALL")  # smoke-test the script itself on a tiny shape
    if which == "lm":
        kmax = kmax or 200
        X, y, support, beta = synth.make_lm(2000, 500, 10) if small else synth.make_lm()
        t0 = time.time()
        t = ref.trace(X, y, data_type=1, model_type=1, ic_type=3, sequence=np.arange(1, kmax + 1))
        extra = {"n": X.shape[0], "p": X.shape[1], "seed": synth.SEED_LM, "true_support": support, "ic_type": 3}
    elif which == "logistic":
        kmax = kmax or 100
        X, y, support, beta = synth.make_logistic(2000, 500, 10) if small else synth.make_logistic()
        t0 = time.time()
        t = ref.trace(X, y, data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, kmax + 1))
        extra = {"n": X.shape[0], "p": X.shape[1], "seed": synth.SEED_LOGISTIC, "true_support": support, "ic_type": 3}
    elif which == "poisson":
        kmax = kmax or 100
        X, y, support, beta = synth.make_poisson(2000, 500, 10) if small else synth.make_poisson()
        t0 = time.time()
        t = ref.trace(X, y, data_type=2, model_type=3, ic_type=3, sequence=np.arange(1, kmax + 1))
        extra = {"n": X.shape[0], "p": X.shape[1], "seed": synth.SEED_POISSON, "true_support": support, "ic_type": 3}
    elif which == "lmcv":
        X, y, support, beta = synth.make_lm(2000, 500, 10) if small else synth.make_lm()
        fold = synth.make_cv_folds(X.shape[0])
        smax = kmax or (40 if small else 200)
        t0 = time.time()
        t = ref.trace(X, y, data_type=1, model_type=1, ic_type=3, path_type=2, s_min=1, s_max=smax, is_cv=True, K=5,
                      cv_fold_id=fold)
        extra = {"n": X.shape[0], "p": X.shape[1], "seed": synth.SEED_LM, "fold_seed": synth.SEED_CV,
                 "true_support": support, "ic_type": 3, "s_max": smax}
        kmax = smax
    elif which == "cox":
        n = kmax or 4000
        p, ktrue, kmax = (300, 8, 12) if small else (2000, 20, 40)
        if small:
            n = 600
        X, tm, status, support, beta = synth.make_cox(n, p, ktrue)
        t0 = time.time()
        t = ref.trace(X, status, data_type=3, model_type=4, ic_type=3, sequence=np.arange(1, kmax + 1))
        extra = {"n": n, "p": p, "k_true": ktrue, "seed": synth.SEED_COX, "true_support": support, "ic_type": 3}
    elif which == "cox-port":
        port = _load("port_ctypes", os.path.join(ROOT, "oracle", "port_ctypes.py"))
        kmax = kmax or 20
        n, p, ktrue = (600, 300, 8) if small else (200000, 20000, 75)
        X, tm, status, support, beta = synth.make_cox(n, p, ktrue)
        print("cox-port: data drawn (n=%d p=%d), oracle starts" % (n, p), flush=True)
        t0 = time.time()
        t = port.trace(X, status, data_type=3, model_type=4, ic_type=3, sequence=np.arange(1, kmax + 1))
        setup_s, path_s = port.last_timing()
        # fingerprints of the inputs the vectors belong to (the test regenerates them from bess_amd/synth.py)
        extra = {"n": n, "p": p, "k_true": ktrue, "seed": synth.SEED_COX, "true_support": support, "ic_type": 3,
                 "kind": "port", "oracle_setup_seconds": setup_s, "oracle_path_seconds": path_s,
                 "status_sum": float(np.sum(status)), "x_probe": np.array([X[0, 0], X[n // 2, p // 2], X[n - 1, p - 1]]),
                 "time_probe": np.array([tm[0], tm[n // 2], tm[n - 1]])}
    elif which == "bigk":
        # sparsity levels beyond the register / LDS resident solvers and beyond the default session capacity (2046)
        X, y, support, beta = synth.make_lm(300, 120, 5, seed=5) if small else synth.make_lm(8000, 2600, 10, seed=5)
        seq = [100, 110] if small else [2040, 2100, 2300]
        kmax = max(seq)
        t0 = time.time()
        t = ref.trace(X, y, data_type=1, model_type=1, ic_type=3, sequence=seq)
        extra = {"n": X.shape[0], "p": X.shape[1], "seed": 5, "sequence": np.array(seq), "ic_type": 3}
    else:
        raise SystemExit("unknown config " + which)
    extra["ref_wall_seconds"] = time.time() - t0
    extra["kmax"] = kmax
    name = {"cox": "fullsize_cox_n%d.npz" % extra["n"], "bigk": "ref_bigk.npz",
            "cox-port": "fullsize_cox_port_prefix.npz"}.get(which, "fullsize_%s.npz" % which)
    out = "/tmp/small_%s.npz" % which if small else os.path.join(os.environ.get("BESS_GOLDEN_OUT", HERE), name)
    np.savez_compressed(out, **pack(t, extra))
    print("done", which, "in", extra["ref_wall_seconds"], "s")


if __name__ == "__main__":
    main()
