"""Logistic and Poisson PDAS + IRLS on the GPU against the plain-C oracle: identical active sets at every
PDAS iteration; coefficients within 1e-6 relative (IRLS iterates are compared after the same number of
steps, so the tolerance covers summation-order differences only)."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P
from helpers import assert_same_trace, hooks
from test_lm_gpu import run_gpu

pytestmark = pytest.mark.gpu


def check(capi, X, y, kw, what, beta_rtol=1e-6):
    want = P.trace(X, y, **kw)
    got = run_gpu(capi, X, y, kw)
    assert_same_trace(got["trace"], want, beta_rtol=beta_rtol, what=what)
    sup_w = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(got["beta"])[0], sup_w), what
    np.testing.assert_allclose(got["beta"][sup_w], want["beta"][sup_w], rtol=beta_rtol)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-7, atol=1e-9)


LOGIT = dict(data_type=2, model_type=2)


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(ic_type=3, sequence=np.arange(1, 21))),
    ("gs", dict(ic_type=4, path_type=2, s_min=1, s_max=30)),
    ("nowarm", dict(ic_type=2, sequence=np.arange(1, 13), is_warm_start=False)),
    ("lambda", dict(ic_type=3, sequence=np.arange(1, 9), lambda_seq=[0.0, 0.05])),
    ("always", dict(ic_type=3, sequence=np.arange(2, 10), always_select=[3])),
])
def test_logistic_paths(gpu, name, kw):
    X, y, _, _ = synth.make_logistic(1000, 200, 8)
    check(gpu, X, y, dict(LOGIT, **kw), "logistic " + name)


def test_logistic_cv_and_weights(gpu):
    X, y, _, _ = synth.make_logistic(1000, 200, 8)
    fold = synth.make_cv_folds(1000, 5)
    check(gpu, X, y, dict(LOGIT, is_cv=True, K=5, cv_fold_id=fold, sequence=np.arange(1, 13)), "logistic cv")
    w = np.random.default_rng(1).uniform(0.5, 2, 1000)
    check(gpu, X, y, dict(LOGIT, ic_type=3, sequence=np.arange(1, 11), weight=w), "logistic weighted")


def test_logistic_bigger(gpu):
    X, y, _, _ = synth.make_logistic(6000, 500, 20, seed=11)
    check(gpu, X, y, dict(LOGIT, ic_type=3, sequence=np.arange(1, 41)), "logistic 6000x500")


def _poisson_data(n=800, p=150, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    b = np.zeros(p)
    b[:5] = [0.3, -0.4, 0.5, 0.2, -0.3]
    return X, rng.poisson(np.exp(X @ b)).astype(float)


POIS = dict(data_type=2, model_type=3)


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(ic_type=3, sequence=np.arange(1, 13))),
    ("gs", dict(ic_type=4, path_type=2, s_min=1, s_max=20)),
])
def test_poisson_paths(gpu, name, kw):
    X, y = _poisson_data()
    check(gpu, X, y, dict(POIS, **kw), "poisson " + name)


def test_poisson_cv(gpu):
    X, y = _poisson_data()
    check(gpu, X, y, dict(POIS, is_cv=True, K=5, cv_fold_id=synth.make_cv_folds(800, 5), sequence=np.arange(1, 9)),
          "poisson cv")


def test_fused_irls_step_gives_the_same_fits(gpu, monkeypatch):
    """The three-launch IRLS step (k_irls_gram: linear predictor, weights, working response and Gram in one pass; the
    convergence test at the head of the solve; the default) against the five-launch step (test hook irls_fuse=0): the same
    arithmetic per row, the linear predictor summed in another order -- identical supports, PDAS iteration counts and
    IRLS step counts, coefficients to rounding."""
    X, y, _, _ = synth.make_logistic(1500, 300, 8, seed=6)
    Xp, yp = X[:, :120], np.random.default_rng(5).poisson(np.exp(np.clip(0.3 * X[:, 0] - 0.2 * X[:, 3], -3, 3))).astype(float)
    outs = []
    for flag in ("0", "1"):
        hooks(monkeypatch, irls_fuse=flag)
        with gpu.Session(X, y, data_type=2, model_type=2) as s:
            a = s.sequential_path(np.arange(1, 25), ic_type=3)
        with gpu.Session(Xp, yp, data_type=2, model_type=3) as s:
            b = s.sequential_path(np.arange(1, 9), ic_type=3)
        outs.append((a, b))
    for u, v in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(u["cand_support"], v["cand_support"])
        np.testing.assert_array_equal(u["cand_iters"], v["cand_iters"])
        np.testing.assert_allclose(u["cand_beta"], v["cand_beta"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(u["cand_ic"], v["cand_ic"], rtol=1e-10)


@pytest.mark.parametrize("fam", ["lm_streaming", "logistic", "cox"])
def test_the_confirming_iteration_queued_light_walks_the_trace_of_the_whole_one(gpu, monkeypatch, fam):
    """Round 6: from the second PDAS iteration of a fit on, the tail (LM: a light slot) goes in right behind the
    selection -- a repeated active set commits after two launches instead of a fall-through batch of IRLS / Newton steps
    (LM: Gram, solve, commit, residual).  Hook light_confirm=0 queues every iteration whole, as rounds 1-5 did: the same
    active set at every iteration of every fit (traced), the same candidates untraced, on designs where fits take up to
    six iterations (a NEW set in a later iteration: the heavy rest is queued after the light part)."""
    from helpers import hooks, assert_same_trace
    from bess_amd import synth
    rng = np.random.default_rng(77)
    if fam == "lm_streaming":
        X, y, _, _ = synth.make_lm(1200, 300, 12, seed=41)
        X = X + 0.7 * X[:, rng.permutation(300)]  # correlated columns: fits of 3-6 iterations
        kw = dict(score_mode=1)
    elif fam == "logistic":
        X, y, _, _ = synth.make_logistic(1500, 250, 10, seed=42)
        X = X + 0.7 * X[:, rng.permutation(250)]
        kw = dict(data_type=2, model_type=2)
    else:
        X, _, y, _, _ = synth.make_cox(1500, 250, 10, seed=43)
        kw = dict(data_type=3, model_type=4)
    seq = np.arange(1, 31)
    outs = {}
    for name, val in (("light", "1"), ("whole", "0")):
        hooks(monkeypatch, light_confirm=val)
        with gpu.Session(X, y, **kw) as s:
            s.trace_enable(True)
            traced = s.sequential_path(seq, ic_type=3)
            s.trace_enable(False)
            fast = s.sequential_path(seq, ic_type=3)
        outs[name] = (traced, fast)
    assert_same_trace(outs["light"][0]["trace"], outs["whole"][0]["trace"], what=fam)
    assert max(len(f["iters"]) for f in outs["light"][0]["trace"]["fits"]) >= 3  # (fits beyond "change, confirm" exist)
    for k in ("cand_support", "cand_iters"):
        assert np.array_equal(outs["light"][1][k], outs["whole"][1][k]), k
        assert np.array_equal(outs["light"][1][k], outs["light"][0][k]), k
    np.testing.assert_allclose(outs["light"][1]["cand_ic"], outs["whole"][1]["cand_ic"], rtol=1e-9)
