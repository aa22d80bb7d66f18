"""Parity of every HIP kernel, called alone through the C ABI (bessx_op_*), against NumPy / the
plain-C oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import port_ctypes as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p", [(97, 8), (130, 33), (1024, 17), (2500, 100), (5000, 257), (9000, 64)])
def test_xtv_matches_numpy(gpu, n, p):
    rng = np.random.default_rng(n * 1000 + p)
    x = rng.standard_normal((n, p))
    v = rng.standard_normal(n)
    h = rng.uniform(0.1, 1.0, n)
    want = x.T @ v
    got = gpu.op_xtv(x, v)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    g1, g2 = gpu.op_xtv(x, v, h)
    np.testing.assert_allclose(g1, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    np.testing.assert_allclose(g2, (x * x).T @ h, rtol=1e-12)
    # bitwise reproducible run to run (fixed reduction tree, no atomics)
    assert np.array_equal(got, gpu.op_xtv(x, v))


@pytest.mark.parametrize("n,p,nc", [(97, 8, 2), (130, 33, 3), (1024, 131, 4), (2500, 100, 8), (5000, 257, 5), (9000, 300, 8)])
def test_multi_chain_score_pass_is_bitwise_the_single_passes(gpu, n, p, nc):
    """k_xtv_mc (round 6): one pass over X for nc chains' vectors leaves bitwise the sums of nc k_xtv launches -- the chunk
    chains that share a pass walk exactly the path they walk on passes of their own."""
    rng = np.random.default_rng(7 * n + p + nc)
    x = rng.standard_normal((n, p))
    vs = rng.standard_normal((nc, n))
    hs = rng.uniform(0.1, 1.0, (nc, n))
    got = gpu.op_xtv_multi(x, vs)
    g1, g2 = gpu.op_xtv_multi(x, vs, hs)
    for c in range(nc):
        assert np.array_equal(got[c], gpu.op_xtv(x, vs[c])), c
        w1, w2 = gpu.op_xtv(x, vs[c], hs[c])
        assert np.array_equal(g1[c], w1) and np.array_equal(g2[c], w2), c
        assert np.array_equal(g1[c], got[c])


@pytest.mark.parametrize("length,k", [(8, 3), (8, 8), (1000, 1), (1000, 37), (10000, 200), (20000, 150), (32768, 254),
                                      (70000, 50), (100000, 200)])
def test_topk_matches_oracle(gpu, length, k):
    rng = np.random.default_rng(length + k)
    s = rng.standard_normal(length) ** 2
    got = gpu.op_topk(s, k)
    assert np.array_equal(got, P.max_k(s, k))


def test_topk_ties_and_specials(gpu):
    s = np.array([1.0, 5.0, 5.0, 0.0, 5.0, 2.0, np.finfo(float).max, 5.0, 0.0, 1.0])
    for k in range(1, 11):
        assert np.array_equal(gpu.op_topk(s, k), P.max_k(s, k)), k
    z = np.zeros(5000)
    assert np.array_equal(gpu.op_topk(z, 7), P.max_k(z, 7))
    s = np.arange(40000, dtype=float) % 97  # many ties across both selection levels
    assert np.array_equal(gpu.op_topk(s, 120), P.max_k(s, 120))


@pytest.mark.parametrize("length", [5, 64, 1000, 4097, 10000, 32768, 50000])
def test_topk_follows_nth_element_under_ties(gpu, length):
    """Equal scores at the selection boundary: the set is the one libstdc++'s std::nth_element leaves in front (what
    max_k of the reference returns, src/utilities.cpp:179-188; restated move by move in the oracle and checked against
    the compiled reference on paths, tests/test_oracle_vs_reference.py) -- k_topk_ties redoes those moves on the device.
    Quantised scores in several densities, every boundary position from 'all tied' to 'one pair tied'."""
    rng = np.random.default_rng(length)
    for levels in (1, 2, 3, 7, 50, 1000):
        s = rng.integers(0, levels, length).astype(float)
        if levels == 50:
            s[rng.choice(length, max(1, length // 10), replace=False)] = np.finfo(float).max  # always_select entries
        ks = sorted({1, 2, 3, length // 2, length - 1, length} | set(int(v) for v in rng.integers(1, length + 1, 4)))
        for k in ks:
            if k > 2046 or k < 1 or (length > 32768 and k > 200):
                continue
            assert np.array_equal(gpu.op_topk(s, k), P.max_k(s, k)), (levels, k)


def test_topk_heap_select_branch_under_ties(gpu):
    """Score vectors on which std::nth_element exhausts its depth limit and finishes with __heap_select (a PDAS iteration
    produced the first; tests/golden/make_heap_select.py), with equal scores at the selection boundary: k_topk_ties takes
    the same branch (one thread, move by move) -- selections of the compiled reference."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "heap_select_ties.npz"))
    for sc, k, want in zip(g["scores"], g["k"], g["selected"]):
        assert np.array_equal(gpu.op_topk(sc, int(k)), want[:k]), int(k)
    base = g["scores"][0]
    order = np.argsort(-base)
    for k in range(2, 40):  # every tie group across every boundary: whichever branch the moves take, as the oracle
        for lo in range(0, k + 1, 2):
            for hi in (k + 1, k + 3):
                sc = base.copy()
                sc[order[lo:hi]] = base[order[lo]]
                assert np.array_equal(gpu.op_topk(sc, k), P.max_k(sc, k)), (k, lo, hi)


@pytest.mark.parametrize("n,p,m", [(97, 8, 3), (500, 40, 16), (1000, 60, 17), (3000, 300, 100), (5000, 400, 200),
                                   (4096, 300, 255), (3000, 700, 600)])
def test_gram_matches_numpy(gpu, n, p, m):
    rng = np.random.default_rng(n + p + m)
    x = rng.standard_normal((n, p))
    cols = np.sort(rng.choice(p, m, replace=False))
    w = rng.uniform(0.0, 2.0, n)
    xa = x[:, cols]
    g = gpu.op_gram(x, cols)
    np.testing.assert_allclose(g, xa.T @ xa, rtol=1e-12, atol=1e-10)
    gw = gpu.op_gram(x, cols, w)
    np.testing.assert_allclose(gw, xa.T @ (xa * w[:, None]), rtol=1e-12, atol=1e-10)
    assert np.array_equal(g, g.T)


@pytest.mark.parametrize("m", [1, 2, 15, 16, 17, 31, 32, 100, 129, 200, 255, 256, 300, 511, 1000])
def test_chol_solve_matches_oracle(gpu, m):
    rng = np.random.default_rng(m)
    a = rng.standard_normal((m + 20, m))
    g = a.T @ a + 0.1 * np.eye(m)
    b = rng.standard_normal(m)
    got = gpu.op_chol_solve(g, b)  # m <= 255: register-resident kernel; above: blocked global-memory Cholesky
    want = P.sym_solve(g, b)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10 * np.abs(want).max())
    np.testing.assert_allclose(g @ got, b, rtol=0, atol=1e-9 * max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("data_type", [1, 2, 3])
def test_normalize_matches_reference_formulas(gpu, data_type):
    rng = np.random.default_rng(data_type)
    n, p = 333, 21
    x = rng.standard_normal((n, p)) * rng.uniform(0.5, 3, p) + rng.uniform(-2, 2, p)
    y = rng.standard_normal(n) + 3
    w = rng.uniform(0.5, 2.0, n)
    xs, ys, xm, xn, ym = gpu.op_normalize(x, y, w, data_type, True, data_type == 1)
    # src/normalize.cpp:20-85 restated with NumPy
    xc = x.copy()
    mean = (w @ xc) / n if data_type in (1, 2) else np.zeros(p)
    xc = xc - mean
    norm = np.sqrt(w @ (xc * xc))
    xc = np.sqrt(n) * xc / norm
    yc = y - (y @ w) / n if data_type == 1 else y.copy()
    if data_type == 1:  # add_weight, src/Data.h:70-77
        xc = xc * np.sqrt(w)[:, None]
        yc = yc * np.sqrt(w)
    np.testing.assert_allclose(xs, xc, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ys, yc, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(xn, norm, rtol=1e-13)
    if data_type in (1, 2):
        np.testing.assert_allclose(xm, mean, rtol=1e-12, atol=1e-14)
    if data_type == 1:
        assert abs(ym - (y @ w) / n) < 1e-13


@pytest.mark.parametrize("data_type,is_normal,add_weight", [(1, True, True), (2, True, False), (3, True, False),
                                                              (1, False, True), (2, False, False)])
def test_normalize_matches_the_oracle(gpu, data_type, is_normal, add_weight):
    """Row a16 against the pinned oracle's own Data::normalize / add_weight (the NumPy restatement above is the
    builder's; this is the one the paths are checked with)."""
    rng = np.random.default_rng(10 * data_type + is_normal)
    n, p = 517, 33
    x = rng.standard_normal((n, p)) * rng.uniform(0.5, 3, p) + rng.uniform(-2, 2, p)
    y = rng.standard_normal(n) + 3
    w = rng.uniform(0.5, 2.0, n)
    got = gpu.op_normalize(x, y, w, data_type, is_normal, add_weight)
    want = P.normalize(x, y, w, data_type, is_normal, add_weight)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12, atol=1e-13)
    if is_normal:
        np.testing.assert_allclose(got[3], want[3], rtol=1e-13)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-12, atol=1e-14)
        assert abs(got[4] - want[4]) < 1e-13


def test_seed_drawn_cv_partition_has_the_reference_shape_and_drives_the_same_path(gpu):
    """Row a17: bessx_session_set_cv(fold_id = NULL) draws a permutation and cuts it like Metric::set_cv_train_test_mask
    (src/Metric.h:66-78): K contiguous chunks of floor(n / K) rows, the last one taking the remainder.  The drawn folds
    are read back and handed to the ORACLE: the cross-validated path must then be the oracle's, fit by fit."""
    from bess_amd import synth
    from helpers import assert_same_trace
    X, y, _, _ = synth.make_lm(503, 60, 5)
    with gpu.Session(X, y) as s:
        s.set_cv(5, None, seed=99)
        fold = s.cv_folds()
        s.trace_enable(True)
        got = s.sequential_path(np.arange(1, 9), ic_type=3, is_cv=True)
    sizes = np.bincount(fold, minlength=5)
    assert list(sizes) == [100, 100, 100, 100, 103] and fold.min() == 0 and fold.max() == 4
    with gpu.Session(X, y) as s2:  # the same seed draws the same folds; another seed does not
        s2.set_cv(5, None, seed=99)
        assert np.array_equal(s2.cv_folds(), fold)
        s2.set_cv(5, None, seed=100)
        assert not np.array_equal(s2.cv_folds(), fold)
    want = P.trace(X, y, ic_type=3, is_cv=True, K=5, cv_fold_id=fold, sequence=np.arange(1, 9))
    assert_same_trace(got["trace"], want, what="seed-drawn folds")
    np.testing.assert_allclose(got["beta"], want["beta"], rtol=1e-6, atol=1e-12)
