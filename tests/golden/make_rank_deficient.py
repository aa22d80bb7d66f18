"""Golden vectors of the COMPILED REFERENCE for designs with exactly dependent columns (duplicated and mirrored TRUE
variables, which a PDAS iteration selects together: a rank-deficient restricted fit that the reference solves with a
pivoted factorisation -- ColPivHouseholderQR of the Gram for LM, src/Algorithm.h:1131-1135; Eigen's LDLT for the IRLS /
Newton systems).  Run in the build container only:   python tests/golden/make_rank_deficient.py
Writes tests/golden/rank_deficient_ref.npz (same layout per case as ref_small.npz).  Cases whose outcome in the
reference is decided by rounding (all coefficients non-zero although two columns are identical: the factorisation
did not see the rank deficiency) are recorded with deterministic = 0 and only replayed for finiteness."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import flatten  # noqa: E402
from oracle import ref_ctypes as R  # noqa: E402
from bess_amd import synth  # noqa: E402


def cases():
    X, y, sup, _ = synth.make_lm(300, 40, 5, seed=5)
    X = np.array(X)
    X[:, 33] = X[:, sup[0]]    # a twin of a true variable
    X[:, 34] = -X[:, sup[1]]   # and a mirror image of another
    out = {"lm_seq": (X, y, dict(ic_type=3, sequence=np.arange(1, 11)), (33, sup[0], 34, sup[1])),
           "lm_gs": (X, y, dict(ic_type=3, path_type=2, s_min=1, s_max=12), (33, sup[0], 34, sup[1])),
           "lm_cold": (X, y, dict(ic_type=3, sequence=np.arange(2, 9), is_warm_start=False), (33, sup[0], 34, sup[1])),
           "lm_cv": (X, y, dict(ic_type=3, sequence=np.arange(1, 8), is_cv=True, K=5,
                                cv_fold_id=synth.make_cv_folds(300, 5)), (33, sup[0], 34, sup[1]))}
    Xl, yl, supl, _ = synth.make_logistic(400, 30, 4, seed=3)
    Xl = np.array(Xl)
    Xl[:, 9] = Xl[:, supl[0]]
    out["logistic_seq"] = (Xl, yl, dict(ic_type=3, sequence=np.arange(1, 7), data_type=2, model_type=2), (9, supl[0]))
    Xc, _, st, supc, _ = synth.make_cox(500, 40, 4, seed=8)
    Xc = np.array(Xc)
    Xc[:, 21] = Xc[:, supc[0]]
    out["cox_seq"] = (Xc, st, dict(ic_type=3, sequence=np.arange(1, 7), data_type=3, model_type=4), (21, supc[0]))
    return out


def main():
    out = {}
    for name, (X, y, kw, twins) in cases().items():
        t = R.trace(X, y, **kw)
        flatten(name, t, out)
        # deterministic: whenever both members of a twin pair are active, one of them has coefficient exactly 0
        det = 1
        for f in t["fits"]:
            for a, b in zip(f["iters"], f["betas"]):
                for u in range(0, len(twins), 2):
                    ia, ib = np.where(a == twins[u])[0], np.where(a == twins[u + 1])[0]
                    if len(ia) and len(ib) and b[ia[0]] != 0.0 and b[ib[0]] != 0.0:
                        det = 0
        out[name + "/deterministic"] = np.array(det)
        print("%-14s fits=%3d iters=%4d deterministic=%d max|beta|=%.3g" % (
            name, len(t["fits"]), sum(len(f["iters"]) for f in t["fits"]), det,
            max(np.max(np.abs(b)) for f in t["fits"] for b in f["betas"])))
    np.savez_compressed(os.path.join(HERE, "rank_deficient_ref.npz"), **out)


if __name__ == "__main__":
    main()
