"""Generate the committed golden vectors from the COMPILED REFERENCE (oracle/_ref/libbess_ref.so).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
Writes
  tests/golden/prostate.csv      the reference's own data set data/prostate.RData (BASELINE configs[0]) as CSV
  tests/golden/ref_small.npz     expected outputs of the reference for the cases in cases.py: the active set of
                                 every PDAS iteration of every fit, the fitted coefficients, every loss / IC value
                                 the path function asked for, and the model it returned; for the screening
                                 cases the kept columns and the model pywrap_bess(is_screening=True) returned.
Inputs of the synthetic cases are not stored: cases.py regenerates them from fixed seeds.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from rdata import read_dataframe  # noqa: E402
import cases  # noqa: E402
from oracle import ref_ctypes as R  # noqa: E402


def flatten(prefix, t, out):
    fits = t["fits"]
    out[prefix + "/beta"] = t["beta"]
    out[prefix + "/scalars"] = np.array([t["coef0"], t["train_loss"], t["ic"], t.get("lambda", 0.0)])
    out[prefix + "/loss_calls"] = t["loss_calls"]
    out[prefix + "/ic_calls"] = t["ic_calls"]
    out[prefix + "/fit_T0"] = np.array([f["T0"] for f in fits], dtype=np.int32)
    out[prefix + "/fit_train_n"] = np.array([f["train_n"] for f in fits], dtype=np.int32)
    out[prefix + "/fit_iters"] = np.array([len(f["iters"]) for f in fits], dtype=np.int32)
    out[prefix + "/iter_len"] = np.array([len(a) for f in fits for a in f["iters"]], dtype=np.int32)
    out[prefix + "/A_flat"] = np.concatenate([a for f in fits for a in f["iters"]]).astype(np.int32)
    out[prefix + "/beta_flat"] = np.concatenate([b for f in fits for b in f["betas"]])
    out[prefix + "/coef0_flat"] = np.array([c for f in fits for c in f["coef0s"]])


def main():
    names, cols = read_dataframe("/root/reference/data/prostate.RData", "prostate")
    M = np.array(cols, dtype=float).T
    with open(os.path.join(HERE, "prostate.csv"), "w") as f:
        f.write(",".join(names) + "\n")
        for row in M:
            f.write(",".join(repr(float(v)) for v in row) + "\n")
    out = {}
    for name, (X, y, kw) in cases.all_cases().items():
        t = R.trace(X, y, **kw)
        flatten(name, t, out)
        print("%-24s fits=%4d iters=%5d best ic=%.10g" % (name, len(t["fits"]), sum(len(f["iters"]) for f in t["fits"]),
                                                         t["ic"]))
    for name, (X, y, ss, kw) in cases.screening_cases().items():
        n, p = X.shape
        w = kw.get("weight")
        w = np.ones(n) if w is None else w
        al = kw.get("always_select", [])
        A = R.screening(X, y, w, kw.get("model_type", 1), ss, al)
        beta, coef0, loss, ic = R.pywrap_bess(
            X, y, kw.get("data_type", 1), w, True, 1, kw.get("model_type", 1), 20, 5, kw.get("path_type", 1), True,
            kw.get("ic_type", 4), False, 5, np.arange(p, dtype=np.int32), [0.0], kw.get("sequence", [1]), [0.0],
            kw.get("s_min", 1), kw.get("s_max", 1), 1, 1e-4, 0.0, 0.0, 1, True, ss, 1, al, 0.0)
        out["scr/" + name + "/A"] = np.asarray(A, dtype=np.int32)
        out["scr/" + name + "/beta"] = np.asarray(beta)
        out["scr/" + name + "/scalars"] = np.array([coef0, loss, ic])
        print("%-24s kept=%3d nnz=%2d ic=%.10g" % (name, len(A), int(np.count_nonzero(beta)), ic))
    np.savez_compressed(os.path.join(HERE, "ref_small.npz"), **out)


if __name__ == "__main__":
    main()
