"""Cox PDAS on the GPU (two-pass suffix-scan score, device-gated Newton with step halving) against the plain-C
oracle.  Supports must agree at every PDAS iteration; coefficients within 1e-6 relative -- the Newton iterates are
compared after the same number of steps (the trace would differ in length otherwise)."""
import numpy as np
import pytest

from bess_amd import synth
from test_glm_gpu import check

from helpers import hooks  # noqa: E402

pytestmark = pytest.mark.gpu
COX = dict(data_type=3, model_type=4)


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(ic_type=3, sequence=np.arange(1, 13))),
    ("gs", dict(ic_type=4, path_type=2, s_min=1, s_max=20)),
    ("nowarm", dict(ic_type=2, sequence=np.arange(1, 9), is_warm_start=False)),
])
def test_cox_paths(gpu, name, kw):
    X, _, status, _, _ = synth.make_cox(600, 100, 6)
    check(gpu, X, status, dict(COX, **kw), "cox " + name)


def test_cox_cv_and_weights(gpu):
    X, _, status, _, _ = synth.make_cox(600, 100, 6)
    check(gpu, X, status, dict(COX, is_cv=True, K=5, cv_fold_id=synth.make_cv_folds(600, 5), sequence=np.arange(1, 9)),
          "cox cv")
    w = np.random.default_rng(1).uniform(0.5, 2, 600)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 9), weight=w), "cox weighted")


@pytest.mark.parametrize("n,p", [(2500, 300), (5000, 130)])
def test_cox_bigger(gpu, n, p):
    X, _, status, _, _ = synth.make_cox(n, p, 8, seed=n)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 17)), "cox %dx%d" % (n, p))


def test_cox_sparsity_levels_above_254(gpu):
    """Beyond the register-resident solver: the n x k work space grows on demand and the Newton system is solved by
    the blocked Cholesky (the reference's default s.list reaches min(p, n / log n))."""
    X, _, status, _, _ = synth.make_cox(1200, 400, 10, seed=77)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=[250, 256, 300]), "cox k>254")


def test_cox_score_pass_forms_agree(gpu, monkeypatch):
    """test hook cox_score=2pass (block totals, carry, rescan: X read twice) and the default one-pass form (summation
    order exchanged, X read once) give the same path: supports at every iteration, coefficients, IC values."""
    X, _, status, _, _ = synth.make_cox(3000, 260, 8, seed=12)
    fold = synth.make_cv_folds(3000, 5)
    outs = []
    for form in ("1pass", "2pass"):
        hooks(monkeypatch, cox_score=form)
        with gpu.Session(X, status, data_type=3, model_type=4) as s:
            s.set_cv(5, fold)
            s.trace_enable(True)
            outs.append((s.sequential_path(np.arange(1, 21), ic_type=3),
                         s.sequential_path(np.arange(1, 7), ic_type=3, is_cv=True)))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
        np.testing.assert_array_equal(a["cand_iters"], b["cand_iters"])
        assert a["n_pdas_iters"] == b["n_pdas_iters"]
        np.testing.assert_allclose(a["cand_beta"], b["cand_beta"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-10)


@pytest.mark.parametrize("form", ["1pass", "2pass"])
def test_cox_hessian_forms_against_the_oracle(gpu, monkeypatch, form):
    """The Newton step's Hessian in one pass over the active columns (k_cox_hess: both Grams and the gradient from one
    read, slab-local suffix sums + carry terms; default from n = 1024) and in the two-pass form built like the
    reference's formulas (M = S1 / S0 materialised; default below), each FORCED on the same problems: plain, weighted,
    ridge, CV row masks, a slab count that does not divide the rows, sparsity levels across tile boundaries."""
    hooks(monkeypatch, cox_hess=form)
    X, _, status, _, _ = synth.make_cox(600, 100, 6)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 13)), "cox seq, hessian " + form)
    w = np.random.default_rng(1).uniform(0.5, 2, 600)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 9), weight=w), "cox weighted, hessian " + form)
    check(gpu, X, status, dict(COX, is_cv=True, K=5, cv_fold_id=synth.make_cv_folds(600, 5), sequence=np.arange(1, 9)),
          "cox cv, hessian " + form)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 7), lambda_seq=[0.0, 0.02]),
          "cox ridge, hessian " + form)
    X, _, status, _, _ = synth.make_cox(5000, 130, 8, seed=5000)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=[3, 15, 16, 17, 31, 32, 33, 40]), "cox 5000x130, hessian " + form)


def test_cox_hessian_forms_agree(gpu, monkeypatch):
    """One-pass and two-pass Hessian on a sample large enough for many row slabs (n = 20 000: 256 slabs, the last one
    short) and sparsity levels up to 10 tile rows: the same path, coefficients to rounding."""
    X, _, status, _, _ = synth.make_cox(20000, 400, 20, seed=31)
    outs = []
    for form in ("1pass", "2pass"):
        hooks(monkeypatch, cox_hess=form)
        with gpu.Session(X, status, data_type=3, model_type=4) as s:
            outs.append(s.sequential_path(np.array([1, 2, 5, 17, 40, 64, 65, 100, 129, 150]), ic_type=3))
    a, b = outs
    np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
    np.testing.assert_array_equal(a["cand_iters"], b["cand_iters"])
    np.testing.assert_allclose(a["cand_beta"], b["cand_beta"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-10)


# ---- Cox with groups of size > 1: the group branch of GroupPdasCox::get_A (algorithm_type 2 / 3) --------------
def _cox_groups(seed, n=500, p=64):
    X, _, status, _, _ = synth.make_cox(n, p, 5, seed=seed)
    rng = np.random.default_rng(seed)
    cuts = np.sort(rng.choice(np.arange(1, p), 19, replace=False))
    return X, status, np.concatenate([[0], cuts]).astype(np.int32)


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(algorithm_type=2, ic_type=3, sequence=np.arange(1, 8))),
    ("l0l2", dict(algorithm_type=3, ic_type=4, sequence=np.arange(1, 5), lambda_seq=[0.0, 0.05])),
    ("gs", dict(algorithm_type=2, ic_type=3, path_type=2, s_min=1, s_max=10)),
    ("nowarm", dict(algorithm_type=2, ic_type=2, sequence=np.arange(1, 6), is_warm_start=False)),
])
def test_cox_group_selection(gpu, name, kw):
    """X_g^T h X_g formed from two weighted group moments (over X and over the suffix-sum matrix) instead of the
    reference's n x n Hessian: same groups at every PDAS iteration as the oracle, which builds the Hessian."""
    X, status, gi = _cox_groups(31)
    check(gpu, X, status, dict(COX, g_index=gi, **kw), "cox groups " + name)


def test_cox_group_selection_cv_weights_and_wide_panels(gpu):
    X, status, gi = _cox_groups(32)
    check(gpu, X, status, dict(COX, algorithm_type=2, g_index=gi, is_cv=True, K=4, cv_fold_id=synth.make_cv_folds(500, 4),
                               sequence=np.arange(1, 5)), "cox groups cv")
    w = np.random.default_rng(2).uniform(0.5, 2, 500)
    check(gpu, X, status, dict(COX, algorithm_type=2, g_index=gi, ic_type=3, sequence=np.arange(1, 6), weight=w),
          "cox groups weighted")
    # more than 256 columns: the suffix-sum matrix is formed one panel of whole groups at a time
    Xw, _, stw, _, _ = synth.make_cox(400, 700, 6, seed=33)
    giw = np.arange(0, 700, 5).astype(np.int32)
    check(gpu, Xw, stw, dict(COX, algorithm_type=2, g_index=giw, ic_type=3, sequence=np.arange(1, 6)), "cox groups wide")


def test_cox_groups_need_the_group_algorithm(gpu):
    X, status, gi = _cox_groups(34)
    with pytest.raises(gpu.BessxError) as e:
        gpu.Session(X, status, data_type=3, model_type=4, algorithm_type=1, g_index=gi)
    assert e.value.code == 3


def test_cox_newton_system_that_is_not_positive_definite(gpu):
    """A ridge term that outweighs the information matrix (small n, lambda > 0.1; the reference adds 2 lambda with the
    sign that subtracts, src/Algorithm.h:1471) makes the Newton system indefinite: the Cholesky kernel gives up and
    the LDL^T fallback kernel solves it, as the reference's LDLT does.  Found by tests/fuzz_parity.py (seed 1, case
    59).  Supports are compared exactly; the coefficients of such a system only to 1e-5."""
    X, _, status, _, _ = synth.make_cox(97, 272, 9, seed=222610918)
    kw = dict(COX, max_iter=3, sequence=np.arange(3, 15), lambda_seq=[0.12957971161634982, 0.15912213204776818,
                                                                     0.17984158281889373],
              always_select=[21, 233], ic_type=3)
    from oracle import port_ctypes as P
    from test_lm_gpu import run_gpu
    want, got = P.trace(X, status, **kw), run_gpu(gpu, X, status, kw)
    assert len(got["trace"]["fits"]) == len(want["fits"]) == 36
    for a, b in zip(got["trace"]["fits"], want["fits"]):
        assert len(a["iters"]) == len(b["iters"])
        for u, v in zip(a["iters"], b["iters"]):
            assert np.array_equal(u, v)
        for u, v in zip(a["betas"], b["betas"]):
            assert np.max(np.abs(u - v)) <= 1e-5 * max(np.max(np.abs(v)), 1e-300)
    np.testing.assert_allclose(got["trace"]["ic_calls"], want["ic_calls"], rtol=1e-6)
    assert np.array_equal(np.nonzero(got["beta"])[0], np.nonzero(want["beta"])[0])
