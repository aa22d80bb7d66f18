"""CPU, build container only: the plain-C oracle against the compiled reference (oracle/_ref/libbess_ref.so)
on fresh random problems that are NOT among the committed golden cases (different seeds, shapes, options).
Skipped where the compiled reference is absent."""
import numpy as np
import pytest

from bess_amd import synth
from helpers import assert_same_trace
from oracle import port_ctypes as P
from oracle import ref_ctypes as R

pytestmark = pytest.mark.skipif(not R.available(), reason="compiled reference (oracle/_ref) not present")


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_lm_random(seed):
    rng = np.random.default_rng(seed)
    n, p = int(rng.integers(150, 600)), int(rng.integers(20, 200))
    X, y, _, _ = synth.make_lm(n, p, min(6, p // 3), seed=seed)
    kmax = min(p, 12)
    for kw in (dict(ic_type=int(rng.integers(1, 5)), sequence=np.arange(1, kmax + 1)),
               dict(ic_type=3, path_type=2, s_min=1, s_max=kmax),
               dict(is_cv=True, K=4, cv_fold_id=synth.make_cv_folds(n, 4, seed=seed), sequence=np.arange(1, 7)),
               dict(ic_type=2, sequence=[kmax, 3, 5], lambda_seq=[0.02, 0.0], max_iter=3)):
        assert_same_trace(P.trace(X, y, **kw), R.trace(X, y, **kw), beta_rtol=1e-8, what="lm %d %r" % (seed, kw))


@pytest.mark.parametrize("seed", [11, 12])
def test_glm_random(seed):
    X, y, _, _ = synth.make_logistic(700, 90, 5, seed=seed)
    kw = dict(data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, 10))
    assert_same_trace(P.trace(X, y, **kw), R.trace(X, y, **kw), beta_rtol=1e-8, what="logit %d" % seed)
    Xc, _, st, _, _ = synth.make_cox(400, 60, 4, seed=seed)
    kw = dict(data_type=3, model_type=4, ic_type=3, sequence=np.arange(1, 8))
    assert_same_trace(P.trace(Xc, st, **kw), R.trace(Xc, st, **kw), beta_rtol=1e-8, what="cox %d" % seed)


@pytest.mark.parametrize("seed", [21, 22])
def test_cox_group_branch_random(seed):
    """GroupPdasCox::get_A for algorithm_type 2 / 3 (explicit n x n Hessian in both the reference and the oracle)."""
    rng = np.random.default_rng(seed)
    X, _, st, _, _ = synth.make_cox(300, 48, 4, seed=seed)
    cuts = np.sort(rng.choice(np.arange(1, 48), 15, replace=False))
    gi = np.concatenate([[0], cuts]).astype(np.int32)
    for kw in (dict(algorithm_type=2, g_index=gi, ic_type=3, sequence=np.arange(1, 6)),
               dict(algorithm_type=3, g_index=gi, ic_type=4, sequence=np.arange(1, 4), lambda_seq=[0.0, 0.1]),
               dict(algorithm_type=2, ic_type=3, sequence=np.arange(1, 6)),  # singleton groups, group formula
               dict(algorithm_type=2, g_index=gi, is_cv=True, K=3, cv_fold_id=synth.make_cv_folds(300, 3, seed=seed),
                    sequence=np.arange(1, 4))):
        kw = dict(kw, data_type=3, model_type=4)
        assert_same_trace(P.trace(X, st, **kw), R.trace(X, st, **kw), beta_rtol=1e-8, what="cox groups %d" % seed)


def _wide_groups(p, seed):
    """group starts with widths 1..40 mixed (some beyond the 16-column register path of the GPU kernels)"""
    rng = np.random.default_rng(seed)
    starts, c = [], 0
    while c < p:
        starts.append(c)
        c += int(rng.choice([1, 2, 5, 16, 17, 24, 40]))
    return np.array(starts, dtype=np.int32)


@pytest.mark.parametrize("seed", [31, 32])
def test_wide_groups_random(seed):
    """Groups wider than 16 columns: the oracle's Jacobi square root against Eigen's sqrt() / LDLT in the reference
    (src/utilities.cpp:142-177), LM and logistic."""
    X, y, _, _ = synth.make_lm(500, 150, 6, seed=seed)
    gi = _wide_groups(150, seed)
    kw = dict(algorithm_type=2, g_index=gi, ic_type=3, sequence=np.arange(1, 5))
    assert_same_trace(P.trace(X, y, **kw), R.trace(X, y, **kw), beta_rtol=1e-8, what="lm wide groups %d" % seed)
    Xl, yl, _, _ = synth.make_logistic(900, 120, 5, seed=seed)
    gl = _wide_groups(120, seed + 1)
    kw = dict(algorithm_type=2, g_index=gl, data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, 4))
    assert_same_trace(P.trace(Xl, yl, **kw), R.trace(Xl, yl, **kw), beta_rtol=1e-7, what="logit wide groups %d" % seed)


def test_score_ties_follow_the_reference_selection():
    """Exact score ties at the selection boundary -- duplicated columns, 0/1 designs with repeated columns -- are broken
    by the moves of libstdc++'s std::nth_element inside max_k (src/utilities.cpp:179-188).  The oracle restates those
    moves (bess_oracle.c: nth_element_libstdcxx), so every PDAS iteration's active set equals the compiled reference's
    where a lower-index tie rule differed in 17 of 25 and 21 of 21 iterations on these two designs."""
    from bess_amd import synth
    rng = np.random.default_rng(1)
    X, y, sup, _ = synth.make_lm(300, 40, 5, seed=5)
    X = np.array(X)
    X[:, 20] = X[:, 7]
    X[:, 33] = X[:, sup[0]]
    Xb = (rng.random((300, 40)) < 0.3).astype(float)
    Xb[:, 11] = Xb[:, 3]
    Xb[:, 12] = Xb[:, 3]
    yb = Xb[:, [3, 5, 8]] @ np.array([2.0, -1.5, 1.0]) + rng.standard_normal(300)
    Xl, yl, _, _ = synth.make_logistic(400, 30, 4, seed=3)
    Xl = np.array(Xl)
    Xl[:, 9] = Xl[:, 2]
    for Xc, yc, kw in ((X, y, dict(ic_type=3, sequence=np.arange(1, 11))), (Xb, yb, dict(ic_type=3, sequence=np.arange(1, 11))),
                       (Xl, yl, dict(ic_type=3, sequence=np.arange(1, 8), data_type=2, model_type=2))):
        a, b = P.trace(Xc, yc, **kw), R.trace(Xc, yc, **kw)
        assert len(a["fits"]) == len(b["fits"])
        for fa, fb in zip(a["fits"], b["fits"]):
            assert len(fa["iters"]) == len(fb["iters"])
            for u, v in zip(fa["iters"], fb["iters"]):
                assert np.array_equal(u, v)
        assert np.array_equal(np.nonzero(a["beta"])[0], np.nonzero(b["beta"])[0])


def test_max_k_against_the_reference_incl_its_heap_select_branch():
    """max_k itself (src/utilities.cpp:179-188), oracle against the compiled reference: random quantised scores (ties at
    every boundary position), and the vectors that exhaust the depth limit of std::nth_element (its __heap_select branch,
    tests/golden/heap_select_ties.npz) with fresh perturbations of their tie groups."""
    import os
    rng = np.random.default_rng(5)
    for _ in range(3000):
        L = int(rng.integers(2, 200))
        k = int(rng.integers(1, L + 1))
        sc = rng.integers(0, int(rng.integers(1, 8)), L).astype(float)
        assert np.array_equal(P.max_k(sc, k), R.max_k(sc, k))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "heap_select_ties.npz"))
    base = g["scores"][0]
    order = np.argsort(-base)
    hits = P.nth_heap_selects()
    for k in range(2, 40):
        for lo in range(0, k + 1):
            for hi in range(k + 1, min(63, k + 6)):
                sc = base.copy()
                sc[order[lo:hi]] = base[order[lo]]
                assert np.array_equal(P.max_k(sc, k), R.max_k(sc, k)), (k, lo, hi)
    assert P.nth_heap_selects() > hits
