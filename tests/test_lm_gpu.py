"""LM path parity on the GPU: every PDAS iteration of every fit (full data and CV folds) must pick
the same active set as the plain-C oracle, coefficients within 1e-6 relative (north_star tolerance)."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P
from helpers import assert_same_trace

pytestmark = pytest.mark.gpu


def run_gpu(capi, X, y, kw, trace=True):
    s = capi.Session(X, y, weight=kw.get("weight"), data_type=kw.get("data_type", 1),
                     is_normal=kw.get("is_normal", True), model_type=kw.get("model_type", 1),
                     algorithm_type=kw.get("algorithm_type", 1),
                     max_iter=kw.get("max_iter", 20), is_warm_start=kw.get("is_warm_start", True),
                     always_select=kw.get("always_select", ()), g_index=kw.get("g_index"),
                     is_screening=kw.get("screening_size", 0) > 0, screening_size=kw.get("screening_size", 0),
                     score_mode=kw.get("score_mode", 0))
    s.trace_enable(bool(trace))  # (trace=False: the fast paths -- chained fits, fused launches, fold fits side by side)
    kept = s.screening()
    if kw.get("is_cv"):
        s.set_cv(kw["K"], kw["cv_fold_id"])
    if kw.get("path_type", 1) == 1:
        out = s.sequential_path(kw["sequence"], kw.get("lambda_seq", (0.0,)), kw.get("ic_type", 4),
                                kw.get("is_cv", False))
    elif kw["path_type"] == 3:
        out = s.pgs_path(kw["s_min"], kw["s_max"], kw["lambda_min"], kw["lambda_max"], kw.get("nlambda", 100),
                         kw.get("powell_path", 1), kw.get("ic_type", 4), kw.get("is_cv", False))
    else:
        out = s.gs_path(kw["s_min"], kw["s_max"], kw.get("ic_type", 4), kw.get("is_cv", False))
    s.close()
    out["screening_A"] = kept
    return out


def check(capi, X, y, kw, what):
    want = P.trace(X, y, **kw)
    got = run_gpu(capi, X, y, kw)
    assert_same_trace(got["trace"], want, what=what)
    sup_w = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(got["beta"])[0], sup_w), what
    np.testing.assert_allclose(got["beta"][sup_w], want["beta"][sup_w], rtol=1e-6)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-8, atol=1e-10)


CASES = {
    "seq": dict(ic_type=3, sequence=np.arange(1, 31)),
    "seq_nowarm": dict(ic_type=4, sequence=np.arange(1, 21), is_warm_start=False),
    "gs": dict(ic_type=3, path_type=2, s_min=1, s_max=40),
    "lambda_snake": dict(ic_type=3, sequence=np.arange(1, 11), lambda_seq=[0.0, 0.01, 0.1]),
    "nonorm": dict(ic_type=3, sequence=np.arange(1, 16), is_normal=False),
    "always": dict(ic_type=3, sequence=np.arange(3, 16), always_select=[5, 7]),
    "maxiter2": dict(ic_type=1, sequence=np.arange(1, 16), max_iter=2, is_warm_start=False),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_lm_paths(gpu, name):
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    check(gpu, X, y, CASES[name], name)


def test_lm_weighted(gpu):
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    w = np.random.default_rng(1).uniform(0.5, 2, 1000)
    check(gpu, X, y, dict(ic_type=3, sequence=np.arange(1, 16), weight=w), "weighted")


@pytest.mark.parametrize("path", ["seq", "gs"])
def test_lm_cv(gpu, path):
    X, y, _, _ = synth.make_lm(1000, 300, 10)
    fold = synth.make_cv_folds(1000, 5)
    kw = dict(is_cv=True, K=5, cv_fold_id=fold)
    kw.update(dict(sequence=np.arange(1, 21)) if path == "seq" else dict(path_type=2, s_min=1, s_max=40))
    check(gpu, X, y, kw, "cv_" + path)


@pytest.mark.parametrize("n,p,kmax", [(97, 8, 8), (130, 500, 30), (5000, 700, 60), (4100, 64, 40)])
def test_lm_shapes(gpu, n, p, kmax):
    X, y, _, _ = synth.make_lm(n, p, min(5, p // 2), seed=n + p)
    check(gpu, X, y, dict(ic_type=4, sequence=np.arange(1, kmax + 1)), "shape %dx%d" % (n, p))


def test_lm_large_k(gpu):
    X, y, _, _ = synth.make_lm(3000, 600, 100, seed=5)
    check(gpu, X, y, dict(ic_type=3, sequence=np.array([1, 50, 100, 150, 200, 254])), "large k")


def test_lm_beyond_register_solver(gpu):
    """Sparsity levels above 254 take the blocked global-memory Cholesky (the reference's default sequence goes up
    to min(p, n / log n), python/bess/linear.py:285-287)."""
    X, y, _, _ = synth.make_lm(2500, 900, 40, seed=9)
    check(gpu, X, y, dict(ic_type=3, sequence=np.array([250, 254, 255, 256, 300, 400])), "k > 254")


def test_logistic_beyond_register_solver(gpu):
    X, y, _, _ = synth.make_logistic(3000, 500, 10, seed=3)
    check(gpu, X, y, dict(data_type=2, model_type=2, ic_type=3, sequence=np.array([253, 254, 260])), "logit k > 253")
