"""The golden cases: seeded inputs + the argument set of each reference run (shared by make_golden.py, which
produced tests/golden/ref_small.npz from the compiled reference, and by the tests that replay them)."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _synth():
    spec = importlib.util.spec_from_file_location("bess_synth", os.path.join(ROOT, "bess_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def prostate():
    M = np.loadtxt(os.path.join(HERE, "prostate.csv"), delimiter=",", skiprows=1)
    return M[:, :8].copy(), M[:, 8].copy()


def readme_lm():
    """The reference's docstring example, python/bess/linear.py:441-465 (legacy NumPy seeding)."""
    rs = np.random.RandomState(12345)
    x = rs.normal(0, 1, 100 * 150).reshape((100, 150))
    beta = np.hstack((np.array([1, 1, -1, -1, -1]), np.zeros(145)))
    noise = rs.normal(0, 1, 100)
    return x, np.matmul(x, beta) + noise


def poisson_data(n=800, p=150, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    b = np.zeros(p)
    b[:5] = [0.3, -0.4, 0.5, 0.2, -0.3]
    return X, rng.poisson(np.exp(X @ b)).astype(float)


def group_data(n=500, p=60, seed=7):
    """Columns in 17 groups of sizes 1..10, correlated inside a group; three groups carry signal."""
    rng = np.random.default_rng(seed)
    gi = np.array([0, 3, 5, 6, 10, 15, 16, 20, 24, 30, 31, 33, 40, 45, 50, 52, 57])
    X = rng.standard_normal((n, p))
    for g in range(len(gi)):
        lo, hi = gi[g], (gi[g + 1] if g + 1 < len(gi) else p)
        X[:, lo:hi] += 0.5 * rng.standard_normal((n, 1))
    beta = np.zeros(p)
    beta[3:5] = [1.5, -1.0]
    beta[20:24] = [0.8, 0.8, -0.6, 0.5]
    beta[45:50] = 0.7
    y = X @ beta + rng.standard_normal(n)
    eta = np.clip(X @ beta * 0.8, -30, 30)
    yb = (rng.uniform(size=n) < 1 / (1 + np.exp(-eta))).astype(float)
    yp = rng.poisson(np.exp(np.clip(X @ beta * 0.3, -5, 3))).astype(float)
    return X, y, yb, yp, gi


def cox_group_data(n=400, p=54, seed=17):
    """Survival data (rows sorted by time, y = status) with 18 groups of sizes 1..6; three groups carry signal."""
    rng = np.random.default_rng(seed)
    gi = np.array([0, 1, 4, 6, 10, 11, 15, 20, 21, 25, 30, 31, 36, 40, 44, 46, 50, 51])
    X = rng.standard_normal((n, p))
    for g in range(len(gi)):
        lo, hi = gi[g], (gi[g + 1] if g + 1 < len(gi) else p)
        X[:, lo:hi] += 0.4 * rng.standard_normal((n, 1))
    beta = np.zeros(p)
    beta[1:4] = [0.8, -0.6, 0.5]
    beta[15:20] = 0.4
    beta[36:40] = [-0.7, 0.5, 0.5, -0.4]
    time = np.power(-np.log(rng.uniform(size=n)) / np.exp(X @ beta), 0.5)
    ctime = np.quantile(time, 0.7) * 2.0 * rng.uniform(size=n)
    status = (time < ctime).astype(float)
    order = np.argsort(np.minimum(time, ctime), kind="stable")
    return np.ascontiguousarray(X[order]), status[order], gi


def all_cases():
    S = _synth()
    c = {}
    # group selection (GroupPdas*, group sizes 1..10)
    Xg, yg, ybg, ypg, gi = group_data()
    G = dict(algorithm_type=2, g_index=gi)
    c["grp_lm_seq"] = (Xg, yg, dict(G, ic_type=3, sequence=np.arange(1, 9)))
    c["grp_lm_gs"] = (Xg, yg, dict(G, ic_type=4, path_type=2, s_min=1, s_max=10))
    c["grp_lm_l0l2"] = (Xg, yg, dict(algorithm_type=3, g_index=gi, ic_type=3, sequence=np.arange(1, 6),
                                     lambda_seq=[0.0, 0.05]))
    c["grp_lm_cv"] = (Xg, yg, dict(G, is_cv=True, K=4, cv_fold_id=S.make_cv_folds(500, 4), sequence=np.arange(1, 7)))
    c["grp_lm_all"] = (Xg, yg, dict(G, ic_type=3, sequence=[len(gi)]))
    c["grp_lm_always"] = (Xg, yg, dict(G, ic_type=3, sequence=np.arange(2, 7), always_select=[4]))
    c["grp_logit_seq"] = (Xg, ybg, dict(G, data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, 8)))
    c["grp_logit_cv"] = (Xg, ybg, dict(G, data_type=2, model_type=2, is_cv=True, K=4,
                                       cv_fold_id=S.make_cv_folds(500, 4), sequence=np.arange(1, 6)))
    c["grp_poisson_seq"] = (Xg, ypg, dict(G, data_type=2, model_type=3, ic_type=3, sequence=np.arange(1, 8)))
    c["grp_lm_powell"] = (Xg, yg, dict(algorithm_type=3, g_index=gi, ic_type=3, path_type=3, s_min=1, s_max=8,
                                       lambda_min=0.01, lambda_max=5.0, nlambda=8, powell_path=2))
    # Cox with groups: the group branch of GroupPdasCox::get_A (algorithm_type 2 / 3, src/Algorithm.h:1497-1568)
    Xcg, stg, gic = cox_group_data()
    GC = dict(data_type=3, model_type=4, algorithm_type=2, g_index=gic)
    c["grp_cox_seq"] = (Xcg, stg, dict(GC, ic_type=3, sequence=np.arange(1, 8)))
    c["grp_cox_cv"] = (Xcg, stg, dict(GC, is_cv=True, K=4, cv_fold_id=S.make_cv_folds(Xcg.shape[0], 4),
                                      sequence=np.arange(1, 6)))
    c["grp_cox_l0l2"] = (Xcg, stg, dict(GC, algorithm_type=3, ic_type=4, sequence=np.arange(1, 5), lambda_seq=[0.0, 0.05]))
    Xp, yp = prostate()
    c["prostate_seq_gic"] = (Xp, yp, dict(ic_type=3, sequence=np.arange(1, 9)))
    c["prostate_one_k3"] = (Xp, yp, dict(ic_type=3, sequence=[3]))
    c["prostate_gs_gic"] = (Xp, yp, dict(ic_type=3, path_type=2, s_min=1, s_max=8))
    c["prostate_seq_ebic"] = (Xp, yp, dict(ic_type=4, sequence=np.arange(1, 9)))
    c["prostate_nonorm"] = (Xp, yp, dict(ic_type=3, sequence=[3], is_normal=False))
    Xr, yr = readme_lm()
    c["readme_seq5"] = (Xr, yr, dict(ic_type=4, sequence=[5]))
    c["readme_seq1_9"] = (Xr, yr, dict(ic_type=4, sequence=np.arange(1, 10)))
    c["readme_gs20"] = (Xr, yr, dict(ic_type=4, path_type=2, s_min=1, s_max=20))
    X, y, _, _ = S.make_lm(1000, 300, 10)
    fold = S.make_cv_folds(1000, 5)
    w = np.random.default_rng(1).uniform(0.5, 2, 1000)
    c["lm_seq"] = (X, y, dict(ic_type=3, sequence=np.arange(1, 31)))
    c["lm_seq_nowarm"] = (X, y, dict(ic_type=4, sequence=np.arange(1, 21), is_warm_start=False))
    c["lm_gs"] = (X, y, dict(ic_type=3, path_type=2, s_min=1, s_max=40))
    c["lm_seq_cv"] = (X, y, dict(is_cv=True, K=5, cv_fold_id=fold, sequence=np.arange(1, 21)))
    c["lm_gs_cv"] = (X, y, dict(is_cv=True, K=5, cv_fold_id=fold, path_type=2, s_min=1, s_max=40))
    c["lm_lambda"] = (X, y, dict(ic_type=3, sequence=np.arange(1, 11), lambda_seq=[0.0, 0.01, 0.1]))
    c["lm_weight"] = (X, y, dict(ic_type=3, sequence=np.arange(1, 16), weight=w))
    c["lm_nonorm"] = (X, y, dict(ic_type=3, sequence=np.arange(1, 16), is_normal=False))
    c["lm_always"] = (X, y, dict(ic_type=3, sequence=np.arange(3, 16), always_select=[5, 7]))
    # Powell path of the L0L2 / bsrr types (pgs_path, src/path.cpp:1138-1309)
    Xs, ys, _, _ = S.make_lm(400, 60, 6)
    PW = dict(algorithm_type=5, path_type=3, s_min=1)
    c["lm_powell_gs"] = (Xs, ys, dict(PW, ic_type=3, s_max=12, lambda_min=0.001, lambda_max=10.0, nlambda=10,
                                      powell_path=1))
    c["lm_powell_seq"] = (Xs, ys, dict(PW, ic_type=4, s_max=20, lambda_min=0.01, lambda_max=100.0, nlambda=20,
                                       powell_path=2))
    c["lm_powell_cv"] = (Xs, ys, dict(PW, is_cv=True, K=4, cv_fold_id=S.make_cv_folds(400, 4), s_max=10,
                                      lambda_min=0.01, lambda_max=10.0, nlambda=10, powell_path=1))
    Xq, yq, _, _ = S.make_logistic(500, 50, 5)
    c["logit_powell_seq"] = (Xq, yq, dict(PW, data_type=2, model_type=2, ic_type=3, s_max=10, lambda_min=0.01,
                                          lambda_max=5.0, nlambda=8, powell_path=2))
    X, y, _, _ = S.make_logistic(1000, 200, 8)
    L = dict(data_type=2, model_type=2)
    c["logit_seq"] = (X, y, dict(L, ic_type=3, sequence=np.arange(1, 21)))
    c["logit_gs"] = (X, y, dict(L, ic_type=4, path_type=2, s_min=1, s_max=30))
    c["logit_cv"] = (X, y, dict(L, is_cv=True, K=5, cv_fold_id=fold, sequence=np.arange(1, 13)))
    c["logit_weight"] = (X, y, dict(L, ic_type=3, sequence=np.arange(1, 11), weight=w))
    X, y = poisson_data()
    Pm = dict(data_type=2, model_type=3)
    c["poisson_seq"] = (X, y, dict(Pm, ic_type=3, sequence=np.arange(1, 13)))
    c["poisson_cv"] = (X, y, dict(Pm, is_cv=True, K=5, cv_fold_id=S.make_cv_folds(800, 5), sequence=np.arange(1, 9)))
    X, _, st, _, _ = S.make_cox(600, 100, 6)
    C = dict(data_type=3, model_type=4)
    c["cox_seq"] = (X, st, dict(C, ic_type=3, sequence=np.arange(1, 13)))
    c["cox_gs"] = (X, st, dict(C, ic_type=4, path_type=2, s_min=1, s_max=20))
    c["cox_cv"] = (X, st, dict(C, is_cv=True, K=5, cv_fold_id=S.make_cv_folds(600, 5), sequence=np.arange(1, 9)))
    c["cox_weight"] = (X, st, dict(C, ic_type=3, sequence=np.arange(1, 9), weight=w[:600]))
    return c


def screening_cases():
    """is_screening=True runs of pywrap_bess (src/bess.cpp:57-61, 186-209): name -> (X, y, screening_size, kwargs).
    Golden: the kept columns (screening_A) and the model pywrap_bess returned, in ref_small.npz under scr/<name>/."""
    S = _synth()
    c = {}
    X, y, _, _ = S.make_lm(600, 400, 8)
    w = np.random.default_rng(2).uniform(0.5, 2, 600)
    c["scr_lm_seq"] = (X, y, 60, dict(ic_type=3, sequence=np.arange(1, 16)))
    c["scr_lm_gs"] = (X, y, 100, dict(ic_type=4, path_type=2, s_min=1, s_max=20))
    c["scr_lm_always"] = (X, y, 50, dict(ic_type=3, sequence=np.arange(3, 12), always_select=[17, 399]))
    c["scr_lm_weight"] = (X, y, 80, dict(ic_type=3, sequence=np.arange(1, 12), weight=w))
    Xr, yr = readme_lm()
    c["scr_readme"] = (Xr, yr, 30, dict(ic_type=4, sequence=np.arange(1, 10)))
    X, y, _, _ = S.make_logistic(700, 300, 6)
    L = dict(data_type=2, model_type=2)
    c["scr_logit_seq"] = (X, y, 50, dict(L, ic_type=3, sequence=np.arange(1, 11)))
    c["scr_logit_weight"] = (X, y, 40, dict(L, ic_type=3, sequence=np.arange(1, 9), weight=np.random.default_rng(3).uniform(0.5, 2, 700)))
    c["scr_logit_always"] = (X, y, 30, dict(L, ic_type=3, sequence=np.arange(2, 9), always_select=[250]))
    X, _, st, _, _ = S.make_cox(500, 200, 5)
    C = dict(data_type=3, model_type=4)
    c["scr_cox_seq"] = (X, st, 40, dict(C, ic_type=3, sequence=np.arange(1, 9)))
    c["scr_cox_always"] = (X, st, 25, dict(C, ic_type=3, sequence=np.arange(2, 8), always_select=[150]))
    return c


def load_screening_golden(name):
    z = np.load(os.path.join(HERE, "ref_small.npz"))
    sc = z["scr/" + name + "/scalars"]
    return {"A": z["scr/" + name + "/A"], "beta": z["scr/" + name + "/beta"], "coef0": float(sc[0]),
            "train_loss": float(sc[1]), "ic": float(sc[2])}


def load_golden(name):
    """Golden trace of one case in the same dict shape the oracle / GPU loaders return."""
    z = np.load(os.path.join(HERE, "ref_small.npz"))
    T0, tn, ni = z[name + "/fit_T0"], z[name + "/fit_train_n"], z[name + "/fit_iters"]
    A, B, C0 = z[name + "/A_flat"], z[name + "/beta_flat"], z[name + "/coef0_flat"]
    ilen = z[name + "/iter_len"]  # columns per iteration (= T0 for singleton groups, sum of group sizes otherwise)
    fits, off, it = [], 0, 0
    for f in range(len(T0)):
        fit = {"T0": int(T0[f]), "train_n": int(tn[f]), "iters": [], "betas": [], "coef0s": []}
        for _ in range(int(ni[f])):
            k = int(ilen[it])
            fit["iters"].append(A[off:off + k])
            fit["betas"].append(B[off:off + k])
            fit["coef0s"].append(float(C0[it]))
            off += k
            it += 1
        fits.append(fit)
    sc = z[name + "/scalars"]
    return {"beta": z[name + "/beta"], "coef0": float(sc[0]), "train_loss": float(sc[1]), "ic": float(sc[2]),
            "lambda": float(sc[3]) if len(sc) > 3 else 0.0,
            "fits": fits, "loss_calls": z[name + "/loss_calls"], "ic_calls": z[name + "/ic_calls"]}
