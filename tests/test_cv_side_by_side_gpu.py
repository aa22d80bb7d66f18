"""GPU: the K fold fits of a cross-validation evaluation (Metric::test_loss, src/Metric.h:150-195) run SIDE BY SIDE --
one fit context per fold on its own stream, one fill of the shared Gram-column caches for every fold that is parked
on missing columns -- and walk exactly the path they walk one after another (test hook cv_side_by_side=0) and the
oracle's: golden-section, sequential (two ridge values: the score-only restart of a fit) and Powell paths, a cache
small enough to be started over while chains are in the middle of their fits, tied scores (the exact tie rule inside
a chain) and nearly collinear columns (a chain's conjugate-gradient solve handed to Cholesky)."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P

from helpers import hooks  # noqa: E402

pytestmark = pytest.mark.gpu


def _paths(s, kmax):
    return (s.gs_path(1, kmax, ic_type=3, is_cv=True),
            s.sequential_path(np.arange(1, min(kmax, 14) + 1), [0.0, 0.03], ic_type=3, is_cv=True),
            s.pgs_path(1, min(kmax, 10), 0.01, 1.0, n_lambda=20, ic_type=3, is_cv=True))


def _both(gpu, monkeypatch, X, y, K, fold, kmax, **kw):
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, cv_side_by_side=mode)
        with gpu.Session(X, y, score_mode=2, **kw) as s:
            s.set_cv(K, fold)
            outs[mode] = (_paths(s, kmax), s.counters())
    (a, ca), (b, cb) = outs["1"], outs["0"]
    assert ca["cv_side_by_side_rounds"] > 0 and cb["cv_side_by_side_rounds"] == 0
    for x, z in zip(a, b):
        assert x["n_fits"] == z["n_fits"] and x["n_pdas_iters"] == z["n_pdas_iters"]
        assert np.array_equal(x["cand_support"], z["cand_support"]) and np.array_equal(x["cand_iters"], z["cand_iters"])
        np.testing.assert_allclose(x["cand_ic"], z["cand_ic"], rtol=1e-10)
        np.testing.assert_allclose(x["cand_beta"], z["cand_beta"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(x["beta"], z["beta"], rtol=1e-9, atol=1e-13)
        assert x["best_T0"] == z["best_T0"]
    return a, ca, cb


@pytest.mark.parametrize("n,p,K", [(1000, 300, 5), (777, 150, 4), (1000, 200, 8)])
def test_fold_chains_side_by_side_walk_the_same_paths(gpu, monkeypatch, n, p, K):
    X, y, _, _ = synth.make_lm(n, p, 10, seed=n)
    fold = synth.make_cv_folds(n, K, seed=3)
    a, ca, cb = _both(gpu, monkeypatch, X, y, K, fold, 30)
    assert ca["cv_union_fills"] > 0
    assert ca["passes_over_X"] <= cb["passes_over_X"]  # a column two folds miss is formed once
    want = P.trace(X, y, ic_type=3, is_cv=True, K=K, cv_fold_id=fold, path_type=2, s_min=1, s_max=30)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(a[0]["beta"])[0], sup)
    np.testing.assert_allclose(a[0]["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose([a[0]["ic"], a[0]["train_loss"]], [want["ic"], want["train_loss"]], rtol=1e-7)
    assert a[0]["n_fits"] == len(want["fits"])


def test_cache_started_over_under_the_chains(gpu, monkeypatch):
    """A cache of 160 columns for 3 chains of up to 24 columns: it fills up along the path, a chain that finds it full is
    parked (cov_stall = 4) and the host starts it over for all chains at once -- also for the chains that are in the
    middle of a fit, whose current columns are formed again by the same fill."""
    hooks(monkeypatch, cov_cap="160")
    X, y, _, _ = synth.make_lm(900, 2500, 20, seed=4)
    fold = synth.make_cv_folds(900, 3, seed=1)
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, cv_side_by_side=mode)
        with gpu.Session(X, y, score_mode=2) as s:
            s.set_cv(3, fold)
            outs[mode] = (s.gs_path(1, 24, ic_type=3, is_cv=True), s.counters())
    (a, ca), (b, cb) = outs["1"], outs["0"]
    assert ca["cv_side_by_side_rounds"] > 0 and ca["cache_restarts"] > 0 and cb["cache_restarts"] > 0
    assert a["n_fits"] == b["n_fits"] and a["n_pdas_iters"] == b["n_pdas_iters"]
    assert np.array_equal(a["cand_support"], b["cand_support"])
    np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-10)
    want = P.trace(X, y, ic_type=3, is_cv=True, K=3, cv_fold_id=fold, path_type=2, s_min=1, s_max=24)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(a["beta"])[0], sup)
    np.testing.assert_allclose(a["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose(a["ic"], want["ic"], rtol=1e-7)


def test_ties_and_collinear_columns_inside_the_chains(gpu, monkeypatch):
    rng = np.random.default_rng(1)
    X, y, sup, _ = synth.make_lm(300, 40, 5, seed=5)
    X = np.array(X)
    noise = [j for j in range(40) if j not in set(sup)]
    X[:, noise[3]] = X[:, noise[0]]   # twins and a mirror image: equal scores at the boundary of level 6
    X[:, noise[9]] = -X[:, noise[0]]
    y = y + 0.6 * X[:, noise[0]]
    fold = synth.make_cv_folds(300, 5)
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, cv_side_by_side=mode)
        with gpu.Session(X, y, score_mode=2) as s:
            s.set_cv(5, fold)
            outs[mode] = (s.sequential_path(np.arange(1, 7), ic_type=3, is_cv=True), s.counters())
    assert outs["1"][1]["tie_rescues"] > 0 and outs["1"][1]["cv_side_by_side_rounds"] > 0
    assert np.array_equal(outs["1"][0]["cand_support"], outs["0"][0]["cand_support"])
    np.testing.assert_allclose(outs["1"][0]["cand_ic"], outs["0"][0]["cand_ic"], rtol=1e-10)
    want = P.trace(X, y, ic_type=3, sequence=np.arange(1, 7), is_cv=True, K=5, cv_fold_id=fold)
    np.testing.assert_allclose(outs["1"][0]["ic"], want["ic"], rtol=1e-7)
    # six clusters of almost equal columns: the conjugate-gradient solve of a chain gives up, Cholesky finishes its slot
    n, p = 1500, 300
    z = rng.standard_normal((n, 6))
    X = np.repeat(z, 50, axis=1) + 1e-4 * rng.standard_normal((n, p))
    y = X[:, 0] - 2 * X[:, 60] + 1.5 * X[:, 130] + rng.standard_normal(n)
    fold = synth.make_cv_folds(n, 4, seed=2)
    a, ca, cb = _both(gpu, monkeypatch, X, y, 4, fold, 20)
    assert ca["cg_fallbacks"] > 0


def test_more_folds_than_contexts_fall_back_to_one_fit_at_a_time(gpu):
    """K = 10 > 8: no fold contexts are created; the folds are fitted one after another on the session's own state."""
    X, y, _, _ = synth.make_lm(1200, 200, 8, seed=9)
    fold = synth.make_cv_folds(1200, 10, seed=2)
    with gpu.Session(X, y, score_mode=2) as s:
        s.set_cv(10, fold)
        out = s.gs_path(1, 20, ic_type=3, is_cv=True)
        assert s.counters()["cv_side_by_side_rounds"] == 0
    want = P.trace(X, y, ic_type=3, is_cv=True, K=10, cv_fold_id=fold, path_type=2, s_min=1, s_max=20)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(out["beta"])[0], sup) and out["n_fits"] == len(want["fits"])
    np.testing.assert_allclose(out["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose(out["ic"], want["ic"], rtol=1e-7)


def test_fold_contexts_are_rebuilt_and_torn_down(gpu):
    """set_cv twice (other K, other folds): contexts, streams and host threads of the first call are dropped and rebuilt;
    a session that never runs a CV path, or runs plain paths between CV paths, tears them down cleanly."""
    X, y, _, _ = synth.make_lm(900, 180, 7, seed=3)
    with gpu.Session(X, y, score_mode=2) as s:
        s.set_cv(4, synth.make_cv_folds(900, 4, seed=1))
    with gpu.Session(X, y, score_mode=2) as s:
        s.set_cv(3, synth.make_cv_folds(900, 3, seed=1))
        a = s.gs_path(1, 15, ic_type=3, is_cv=True)
        fold = synth.make_cv_folds(900, 6, seed=2)
        s.set_cv(6, fold)
        plain = s.sequential_path(np.arange(1, 10), ic_type=3)
        b = s.gs_path(1, 15, ic_type=3, is_cv=True)
        one = s.fit(5, 0.0, fold=2)  # a fold fitted on the session's own state: the contexts start over afterwards
        c = s.gs_path(1, 15, ic_type=3, is_cv=True)
        assert s.counters()["cv_side_by_side_rounds"] > 0
    assert a["n_fits"] > 0 and len(one["support"]) == 5 and plain["n_candidates"] == 9
    assert np.array_equal(b["cand_support"], c["cand_support"]) and np.array_equal(b["cand_ic"], c["cand_ic"])
    want = P.trace(X, y, ic_type=3, is_cv=True, K=6, cv_fold_id=fold, path_type=2, s_min=1, s_max=15)
    np.testing.assert_allclose(b["ic"], want["ic"], rtol=1e-7)
    assert np.array_equal(np.nonzero(b["beta"])[0], np.nonzero(want["beta"])[0])


def test_chains_opened_from_uploaded_supports_share_one_fill(gpu, monkeypatch):
    """All K chains start from uploaded supports (their device state was lost to a fold fitted on the session's own
    state) into a nearly full cache: the Gram columns of the K initial supports are formed by ONE fill whose restart is
    decided once -- until round 3 every chain ran its own slot-0 lookup, and a later chain's restart evicted what an
    earlier chain had just filled (BESSX_ERR_NUMERIC 'an active column was missing from the Gram column cache')."""
    hooks(monkeypatch, cov_cap="160")
    X, y, _, _ = synth.make_lm(900, 2500, 20, seed=4)
    fold = synth.make_cv_folds(900, 3, seed=1)
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, cv_side_by_side=mode)
        with gpu.Session(X, y, score_mode=2) as s:
            s.set_cv(3, fold)
            recs, init = [], (np.zeros(0, np.int32), np.zeros(0))
            for T0 in (26, 28, 30):
                r = s.cv_eval(T0, 0.0, True, init[0], init[1], 0.0, [0, 1, 2])
                init = (r[0]["support"], r[0]["beta"])
                recs.append(r)
            s.fit(6, 0.0, fold=1)  # a fold on the session's own state: the contexts lose their device state
            recs.append(s.cv_eval(31, 0.0, True, init[0], init[1], 0.0, [0, 1, 2]))
            recs.append(s.cv_eval(32, 0.0, False, init[0], init[1], 0.0, [0, 2]))
            outs[mode] = (recs, s.counters())
    (a, ca), (b, cb) = outs["1"], outs["0"]
    assert ca["cv_side_by_side_rounds"] > 0 and cb["cv_side_by_side_rounds"] == 0
    assert ca["cache_restarts"] > 0
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb)
        for x, z in zip(ra, rb):
            np.testing.assert_array_equal(x["support"], z["support"])
            assert x["iters"] == z["iters"]
            np.testing.assert_allclose(x["beta"], z["beta"], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose([x["train_loss"], x["test_loss"]], [z["train_loss"], z["test_loss"]], rtol=1e-10)


def test_dropped_contexts_are_visible(gpu):
    X, y, _, _ = synth.make_lm(600, 100, 5)
    with gpu.Session(X, y, score_mode=2) as s:
        s.set_cv(4, synth.make_cv_folds(600, 4))
        c = s.counters()
        assert c["cv_fold_contexts"] == 4 and c["cv_contexts_dropped"] == 0
