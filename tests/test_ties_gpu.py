"""Exact score ties at the selection boundary -- duplicated columns, 0/1 designs with repeated and complementary
columns -- on the GPU: every PDAS iteration's active set equals the oracle's, whose tie rule is the moves of the
reference's std::nth_element inside max_k (src/utilities.cpp:179-188; the oracle is checked against the compiled
reference on the same designs by tests/test_oracle_vs_reference.py::test_score_ties_follow_the_reference_selection).
Covers the selection kernels of every path: streaming LM, both LM score forms incl. the fused covariance-form launches
(arg-max shortcut, repeated-set shortcut, k_sel_cgr: the fit is parked and the slot redone with k_topk_ties), logistic,
Cox, groups, cross-validation and screening."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P
from helpers import assert_same_trace
from test_lm_gpu import run_gpu

pytestmark = pytest.mark.gpu


def _designs():
    """Five true variables plus a twin (and a mirror-image) pair of weak ones that rank sixth: at sparsity level 6 --
    and in the PDAS iterations on the way -- the k-th and the (k + 1)-th largest score are EQUAL.  (At level 7 both
    twins would be selected: a rank-deficient restricted fit, which the reference solves by a pivoted factorisation
    whose outcome on exactly dependent columns is decided by rounding -- DESIGN.md, out of scope; the paths below stop
    at 6.)"""
    rng = np.random.default_rng(1)
    X, y, sup, _ = synth.make_lm(300, 40, 5, seed=5)
    X = np.array(X)
    noise = [j for j in range(40) if j not in set(sup)]
    a, b, c = noise[0], noise[3], noise[9]
    X[:, b] = X[:, a]     # twins
    X[:, c] = -X[:, a]    # and a mirror image: the same score again
    y = y + 0.6 * X[:, a]
    Xb = (rng.random((300, 40)) < 0.3).astype(float)
    Xb[:, 11] = Xb[:, 20]
    Xb[:, 12] = 1.0 - Xb[:, 20]  # complementary 0/1 column: equal after centring and scaling
    yb = Xb[:, [3, 5, 8]] @ np.array([2.0, -1.5, 1.0]) + 0.5 * Xb[:, 20] + 0.3 * rng.standard_normal(300)
    return (("duplicates", X, y, 6), ("binary", Xb, yb, 4))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_lm_paths_with_tied_scores(gpu, mode):
    for name, X, y, kmax in _designs():
        kws = [dict(ic_type=3, sequence=np.arange(1, kmax + 1)),
               dict(ic_type=3, path_type=2, s_min=1, s_max=kmax),
               dict(ic_type=3, sequence=np.arange(1, kmax + 1), is_cv=True, K=5, cv_fold_id=synth.make_cv_folds(300, 5))]
        if name == "binary":  # (a cold start at level 6 of the other design takes both twins at once: rank-deficient)
            kws.append(dict(ic_type=3, sequence=np.arange(1, kmax + 1), is_warm_start=False))
        for kw in kws:
            want = P.trace(X, y, **kw)
            got = run_gpu(gpu, X, y, dict(kw, score_mode=mode))
            assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="ties %s score_mode %d" % (name, mode))
            assert np.array_equal(np.nonzero(got["beta"])[0], np.nonzero(want["beta"])[0])


def test_untraced_chained_path_with_tied_scores(gpu):
    """The fast (untraced, chained) covariance-form path -- fits queued behind each other on the device, the grow-by-one
    arg-max -- selects the same supports as the traced one and the oracle."""
    for name, X, y, kmax in _designs():
        seq = np.arange(1, kmax + 1)
        want = P.trace(X, y, ic_type=3, sequence=seq)
        with gpu.Session(X, y) as s:
            out = s.sequential_path(seq, ic_type=3)
        for i, f in enumerate(want["fits"]):
            assert np.array_equal(out["cand_support"][i, :seq[i]], f["iters"][-1]), (name, i)
            assert out["cand_iters"][i] == len(f["iters"])


def test_glm_and_cox_paths_with_tied_scores(gpu):
    Xl, yl, supl, _ = synth.make_logistic(400, 30, 4, seed=3)
    Xl = np.array(Xl)
    noise = [j for j in range(30) if j not in set(supl)]
    Xl[:, noise[4]] = Xl[:, noise[1]]   # twin noise columns: tied whenever both are outside the set
    Xl[:, noise[6]] = -Xl[:, noise[1]]
    kw = dict(ic_type=3, sequence=np.arange(1, 5), data_type=2, model_type=2)
    assert_same_trace(run_gpu(gpu, Xl, yl, kw)["trace"], P.trace(Xl, yl, **kw), beta_rtol=1e-6, what="logistic ties")
    Xc, _, st, supc, _ = synth.make_cox(500, 40, 4, seed=8)
    Xc = np.array(Xc)
    noise = [j for j in range(40) if j not in set(supc)]
    Xc[:, noise[5]] = Xc[:, noise[2]]
    kw = dict(ic_type=3, sequence=np.arange(1, 5), data_type=3, model_type=4)
    assert_same_trace(run_gpu(gpu, Xc, st, kw)["trace"], P.trace(Xc, st, **kw), beta_rtol=1e-5, what="cox ties")


def test_screening_with_tied_marginal_scores(gpu):
    X, y, _, _ = synth.make_lm(300, 60, 4, seed=9)
    X = np.array(X)
    X[:, 30:36] = X[:, 10:16]  # six duplicated columns: equal marginal coefficients at the screening boundary
    for ss in (8, 12, 33):
        keep = P.screening(X, y, None, 1, ss)
        with gpu.Session(X, y, is_screening=True, screening_size=ss) as s:
            assert np.array_equal(s.screening(), keep), ss
