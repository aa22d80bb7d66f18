"""BASELINE configs[2], [3], [4] at FULL size on the GPU (configs[1]: tests/test_fullsize_gpu.py).

Each config is checked two ways:

1. Size-independent properties that need no oracle: run-to-run bitwise determinism, the IC formula, the stopping rule
   of Algorithm::fit (src/Algorithm.h:164-170), agreement of the alternative forms of the same arithmetic (LM
   covariance / streaming score pass, Cox one-pass / two-pass score, fold-sharded CV driver / the library's own CV
   path), recovery of the planted support where the signal allows it.
2. Golden vectors of the COMPILED REFERENCE (oracle/_ref/libbess_ref.so run in the build container by
   tests/golden/make_fullsize_ref.py; hours of one CPU core each): the active set of EVERY PDAS iteration of every
   fit, coefficients, losses and criteria.  A golden file whose `truncated` flag is set holds a PREFIX of the path
   (the reference ran out of its time budget, BESS_REF_BUDGET_S); the comparison then covers that prefix of fits and
   says so in the assertion message.  configs[4] cannot run on the reference at n = 200 000 (its n x n risk-set
   matrix, src/Algorithm.h:1386, is 320 GB): the golden file is the same recipe at n = 4000, the largest the
   reference holds comfortably, and the full size is covered by properties.
"""
import os

import numpy as np
import pytest

from bess_amd import synth
from bess_amd import dist as bdist

from helpers import assert_untraced_path_matches_golden, hooks  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    """The committed golden vectors of the compiled reference.  A missing file is a FAILURE, not a skip: these are
    the strongest comparisons of the suite and the files are part of the repository."""
    path = os.path.join(GOLDEN, name)
    assert os.path.exists(path), "golden file %s is missing (tests/golden/make_fullsize_ref.py regenerates it)" % path
    return np.load(path)


def assert_matches_golden(trace, g, what, beta_rtol=1e-6, metric_rtol=1e-8):
    """Compare a traced path with a golden file of make_fullsize_ref.py (whole path, or its prefix if truncated)."""
    nfit = len(g["fit_T0"])
    label = "%s (%s%d fits of the compiled reference)" % (what, "first " if int(g["truncated"]) else "", nfit)
    fits = trace["fits"]
    assert len(fits) >= nfit if int(g["truncated"]) else len(fits) == nfit, label
    fits = fits[:nfit]
    assert [f["T0"] for f in fits] == list(g["fit_T0"]), label + ": order of sparsity levels"
    assert [f["train_n"] for f in fits] == list(g["fit_train_n"]), label + ": training rows"
    assert [len(f["iters"]) for f in fits] == list(g["fit_iters"]), label + ": PDAS iterations per fit"
    got_A = np.concatenate([a for f in fits for a in f["iters"]])
    assert np.array_equal(got_A, g["A_flat"]), label + ": active sets"  # bit-exact, every iteration of every fit
    got_b = np.concatenate([b for f in fits for b in f["betas"]])
    scale = np.max(np.abs(g["beta_flat"]))
    np.testing.assert_allclose(got_b, g["beta_flat"], rtol=beta_rtol, atol=beta_rtol * scale, err_msg=label)
    got_c = np.array([c for f in fits for c in f["coef0s"]])
    np.testing.assert_allclose(got_c, g["coef0_flat"], rtol=beta_rtol, atol=beta_rtol * scale, err_msg=label)
    nic, nloss = len(g["ic_calls"]), len(g["loss_calls"])
    np.testing.assert_allclose(trace["ic_calls"][:nic], g["ic_calls"], rtol=metric_rtol, err_msg=label + " ic")
    np.testing.assert_allclose(trace["loss_calls"][:nloss], g["loss_calls"], rtol=metric_rtol, err_msg=label + " loss")
    return nfit


def assert_best_model(out, g, rtol=1e-6):
    if int(g["truncated"]):
        return
    assert np.array_equal(np.nonzero(out["beta"])[0], g["best_beta_idx"])
    np.testing.assert_allclose(out["beta"][g["best_beta_idx"]], g["best_beta_val"], rtol=rtol)
    np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                               [float(g["best_coef0"]), float(g["best_train_loss"]), float(g["best_ic"])], rtol=rtol,
                               atol=1e-9)


def assert_fits_stop_on_a_repeat(fits, max_iter=20):
    """Algorithm::fit returns when the new active set equals an earlier one of the same fit -- or column 0 of A_list,
    which is all zeros -- or after max_iter iterations."""
    for f in fits:
        last = f["iters"][-1]
        seen = [np.zeros_like(last)] + list(f["iters"][:-1])
        assert len(f["iters"]) == max_iter or any(np.array_equal(last, a) for a in seen), (f["T0"], len(f["iters"]))


# ------------------------------------------------------------------------------------------------ configs[2]
@pytest.fixture(scope="module")
def logistic_full(gpu):
    n, p = 100000, 5000
    X, y, support, beta = synth.make_logistic()
    s = gpu.Session(X, y, data_type=2, model_type=2)
    del X
    s.trace_enable(True)
    out = s.sequential_path(np.arange(1, 101), ic_type=3)
    yield s, out, support, (n, p)
    s.close()


def test_logistic_config_properties_at_full_size(logistic_full):
    s, out, support, (n, p) = logistic_full
    again = s.sequential_path(np.arange(1, 101), ic_type=3)
    for k in ("cand_support", "cand_beta", "cand_ic", "cand_iters", "cand_coef0"):
        assert np.array_equal(out[k], again[k]), k  # bitwise reproducible
    fits = out["trace"]["fits"]
    assert len(fits) == 100 and [f["T0"] for f in fits] == list(range(1, 101))
    assert all(f["train_n"] == n for f in fits)
    assert_fits_stop_on_a_repeat(fits)
    # LogisticMetric::ic, src/Metric.h:365-414: loss + log(p) log(log n) T0 (no n log(.))
    c = np.log(p) * np.log(np.log(n))
    np.testing.assert_allclose(out["cand_ic"], out["cand_train_loss"] + c * np.arange(1, 101), rtol=1e-12)
    assert np.all(np.diff(out["cand_train_loss"][:40]) < 0)
    # 50 planted variables with |beta| in [2m, 10m]: the GIC-selected model keeps only planted variables
    chosen = np.nonzero(out["beta"])[0]
    assert 35 <= out["best_T0"] <= 50 and np.all(np.isin(chosen, support)), (out["best_T0"], chosen)
    # golden section on the same range ends at the same size (the criterion is unimodal on this problem)
    gs = s.gs_path(1, 100, ic_type=3)
    assert gs["best_T0"] == out["best_T0"]
    assert np.array_equal(np.nonzero(gs["beta"])[0], chosen)


def test_logistic_config_matches_compiled_reference_at_full_size(logistic_full):
    g = _gold("fullsize_logistic.npz")
    _, out, _, _ = logistic_full
    nfit = assert_matches_golden(out["trace"], g, "configs[2] logistic n=100000 p=5000", beta_rtol=1e-6)
    assert nfit >= 20
    assert_best_model(out, g)


def test_logistic_benchmarked_path_matches_compiled_reference_at_full_size(gpu, logistic_full):
    """The UNTRACED path bench.py times (tracing switches the chunk chains off): chain count automatic, one chain and
    three chains forced -- every candidate's final support, iteration count, coefficients, intercept, loss and criterion
    against the compiled reference's golden."""
    g = _gold("fullsize_logistic.npz")
    s = logistic_full[0]
    X, y, _, _ = synth.make_logistic()
    s.trace_enable(False)
    try:
        for chains in (0, 1, 3):
            s.set_kpath_chains(chains)
            out = s.sequential_path(np.arange(1, 101), ic_type=3)
            assert_untraced_path_matches_golden(out, g, X, 2, "configs[2] untraced, chains=%d" % chains, metric_rtol=1e-8)
            if chains == 3:
                assert s.counters()["kpath_chains_last_path"] == 3
    finally:
        s.set_kpath_chains(0)
        s.trace_enable(True)


# ------------------------------------------------------------------------------------ Poisson
@pytest.fixture(scope="module")
def poisson_full(gpu):
    """The Poisson family at the shape of configs[2] (north_star names poisson.cpp; no BASELINE config of its own):
    n = 100000, p = 5000, k = 1..100, warm-started IRLS (src/Algorithm.h:1273-1322), no clamp in get_A (:1338-1339)."""
    n, p = 100000, 5000
    X, y, support, beta = synth.make_poisson()
    s = gpu.Session(X, y, data_type=2, model_type=3)
    del X
    s.trace_enable(True)
    out = s.sequential_path(np.arange(1, 101), ic_type=3)
    yield s, out, support, (n, p)
    s.close()


def test_poisson_properties_at_full_size(poisson_full):
    s, out, support, (n, p) = poisson_full
    again = s.sequential_path(np.arange(1, 101), ic_type=3)
    for k in ("cand_support", "cand_beta", "cand_ic", "cand_iters", "cand_coef0"):
        assert np.array_equal(out[k], again[k]), k  # bitwise reproducible
    fits = out["trace"]["fits"]
    assert len(fits) == 100 and [f["T0"] for f in fits] == list(range(1, 101))
    assert_fits_stop_on_a_repeat(fits)
    # PoissonMetric::ic, src/Metric.h:504-553: loss + log(p) log(log n) T0
    c = np.log(p) * np.log(np.log(n))
    np.testing.assert_allclose(out["cand_ic"], out["cand_train_loss"] + c * np.arange(1, 101), rtol=1e-12)
    chosen = np.nonzero(out["beta"])[0]
    assert 30 <= out["best_T0"] <= 50 and np.all(np.isin(chosen, support)), (out["best_T0"], chosen)
    s.trace_enable(False)
    fast = s.sequential_path(np.arange(1, 101), ic_type=3)  # untraced: the IRLS chains batched by the last step count
    assert np.array_equal(fast["cand_support"], out["cand_support"])
    np.testing.assert_allclose(fast["cand_ic"], out["cand_ic"], rtol=1e-12)


def test_poisson_matches_compiled_reference_at_full_size(poisson_full):
    g = _gold("fullsize_poisson.npz")
    _, out, _, _ = poisson_full
    nfit = assert_matches_golden(out["trace"], g, "Poisson n=100000 p=5000", beta_rtol=1e-6)
    assert nfit >= 20
    assert_best_model(out, g)


def test_poisson_benchmarked_path_matches_compiled_reference_at_full_size(gpu, poisson_full):
    g = _gold("fullsize_poisson.npz")
    s = poisson_full[0]
    X, y, _, _ = synth.make_poisson()
    s.trace_enable(False)
    try:
        for chains in (0, 1, 3):
            s.set_kpath_chains(chains)
            out = s.sequential_path(np.arange(1, 101), ic_type=3)
            assert_untraced_path_matches_golden(out, g, X, 2, "Poisson untraced, chains=%d" % chains, metric_rtol=1e-8)
    finally:
        s.set_kpath_chains(0)
        s.trace_enable(True)


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.fixture(scope="module")
def lmcv_full(gpu):
    X, y, support, beta = synth.make_lm()
    fold = synth.make_cv_folds(X.shape[0], 5)
    outs = {}
    for mode in (2, 1):  # covariance form, streaming form
        with gpu.Session(X, y, score_mode=mode) as s:
            s.set_cv(5, fold)
            s.trace_enable(True)
            outs[mode] = s.gs_path(1, 200, ic_type=3, is_cv=True)
            if mode == 2:
                s.trace_enable(False)
                outs["again"] = s.gs_path(1, 200, ic_type=3, is_cv=True)
                cv = bdist.FoldShardedCV(s, 5)
                outs["sharded"] = cv.gs_path(1, 200)
    return outs, support, fold


def test_lm_cv_golden_section_properties_at_full_size(lmcv_full):
    outs, support, fold = lmcv_full
    cov, stream, again, sharded = outs[2], outs[1], outs["again"], outs["sharded"]
    for k in ("cand_T0", "cand_support", "cand_beta", "cand_ic"):
        assert np.array_equal(cov[k], again[k]), k
    # the two evaluations of the score pass walk the same path: same candidates, fits, PDAS iterations, supports
    assert np.array_equal(cov["cand_T0"], stream["cand_T0"]) and cov["n_fits"] == stream["n_fits"]
    assert cov["n_pdas_iters"] == stream["n_pdas_iters"]
    assert np.array_equal(cov["cand_support"], stream["cand_support"])
    np.testing.assert_allclose(cov["cand_ic"], stream["cand_ic"], rtol=1e-9)
    ta, tb = cov["trace"]["fits"], stream["trace"]["fits"]
    assert [len(f["iters"]) for f in ta] == [len(f["iters"]) for f in tb]
    assert all(np.array_equal(x, y) for a, b in zip(ta, tb) for x, y in zip(a["iters"], b["iters"]))
    # every ic() under CV = K fold fits on floor(4n/5) training rows + (gs_path) the full-data fit before it
    n = 50000
    assert sorted(set(f["train_n"] for f in ta)) == [40000, n]
    assert_fits_stop_on_a_repeat(ta)
    # the fold-sharded driver built on bessx_session_fit (bess_amd/dist.py) reproduces the library's own path
    assert sharded["best_T0"] == cov["best_T0"] and sharded["n_fits"] == cov["n_fits"]
    assert np.array_equal(sharded["cand_T0"], cov["cand_T0"])
    np.testing.assert_allclose(sharded["cand_ic"], cov["cand_ic"], rtol=1e-12)
    np.testing.assert_allclose(sharded["beta"], cov["beta"], rtol=1e-10)
    # CV picks the planted size on this signal-to-noise ratio, and (q5) the returned beta is the LAST FOLD's fit
    assert cov["best_T0"] == 100 and np.array_equal(np.nonzero(cov["beta"])[0], support)


def test_lm_cv_golden_section_matches_compiled_reference_at_full_size(lmcv_full):
    g = _gold("fullsize_lmcv.npz")
    outs, _, _ = lmcv_full
    for mode in (2, 1):
        nfit = assert_matches_golden(outs[mode]["trace"], g, "configs[3] LM gs_path + 5-fold CV, score_mode %d" % mode)
        # the committed golden is the complete run (145 fits, 368 PDAS iterations, 8 515 s of the compiled reference); a
        # regenerated, time-boxed one must still cover the two first golden-section points, each evaluated twice (q4)
        assert nfit == len(g["fit_T0"]) if not int(g["truncated"]) else nfit >= 12
        assert_best_model(outs[mode], g)


# ------------------------------------------------------------------------------------------------ configs[4]
def test_cox_config_recipe_matches_compiled_reference_at_n4000(gpu):
    """The configs[4] recipe (synth.make_cox, SEED_COX) at the largest n the reference's dense n x n risk-set matrix
    allows: every PDAS iteration of k = 1..40 on n = 4000, p = 2000 against the compiled reference."""
    g = _gold("fullsize_cox_n4000.npz")
    n, p, ktrue, kmax = int(g["n"]), int(g["p"]), int(g["k_true"]), int(g["kmax"])
    X, _, status, support, _ = synth.make_cox(n, p, ktrue)
    for form in ("1pass", "2pass"):
        os.environ["BESSX_TEST_HOOKS"] = "cox_score=" + form
        try:
            with gpu.Session(X, status, data_type=3, model_type=4) as s:
                s.trace_enable(True)
                out = s.sequential_path(np.arange(1, kmax + 1), ic_type=3)
        finally:
            os.environ.pop("BESSX_TEST_HOOKS", None)
        assert_matches_golden(out["trace"], g, "configs[4] recipe at n=%d p=%d, %s score" % (n, p, form),
                              beta_rtol=1e-5, metric_rtol=1e-7)
        assert_best_model(out, g, rtol=1e-5)


def test_cox_benchmarked_path_matches_compiled_reference_at_n4000(gpu):
    """The same golden against the UNTRACED Cox path: one chain, and chunk chains forced (2, 3)."""
    g = _gold("fullsize_cox_n4000.npz")
    n, p, ktrue, kmax = int(g["n"]), int(g["p"]), int(g["k_true"]), int(g["kmax"])
    X, _, status, support, _ = synth.make_cox(n, p, ktrue)
    with gpu.Session(X, status, data_type=3, model_type=4) as s:
        for chains in (1, 2, 3, 0):
            s.set_kpath_chains(chains)
            out = s.sequential_path(np.arange(1, kmax + 1), ic_type=3)
            assert_untraced_path_matches_golden(out, g, X, 3, "configs[4] recipe n=%d untraced, chains=%d" % (n, chains),
                                                beta_rtol=1e-5, metric_rtol=1e-7)
            if chains in (2, 3):
                assert s.counters()["kpath_chains_last_path"] == chains


@pytest.fixture(scope="module")
def cox_full(gpu):
    n, p, kmax = 200000, 20000, 150
    X, _, status, support, beta = synth.make_cox()
    s = gpu.Session(X, status, data_type=3, model_type=4)
    s.trace_enable(True)
    out = s.sequential_path(np.arange(1, kmax + 1), ic_type=3)
    yield s, out, support, (X, status), (n, p, kmax)
    s.close()


class _Prefix:
    """A golden of k = 1..K read as a PREFIX of a longer warm-start chain (candidate k starts from candidate k - 1's model,
    src/path.cpp:60-64: the first K candidates of 1..150 are the candidates of 1..K)."""

    def __init__(self, g):
        self._g = g
        self.files = list(g.files)

    def __getitem__(self, k):
        return np.int64(1) if k == "truncated" else self._g[k]


def test_cox_config_matches_the_pinned_oracle_at_full_size(gpu, cox_full, monkeypatch):
    """configs[4] at the BENCHMARKED size, n = 200 000, p = 20 000: every PDAS iteration of k = 1..20 against the plain-C
    oracle (oracle/bess_oracle.c, kind "port" -- the reference itself needs a 320 GB n x n matrix here,
    src/Algorithm.h:1386; the port is pinned against the compiled reference in tests/test_oracle_vs_reference.py and at
    n = 4000 above).  tests/golden/make_fullsize_ref.py cox-port 20 (574 s of one host core of the GPU box).  Traced
    (1-pass and 2-pass score) and untraced (one chain, chunk chains forced), src/Algorithm.h:1377-1649."""
    g = _gold("fullsize_cox_port_prefix.npz")
    assert str(g["kind"]) == "port" and int(g["n"]) == 200000 and int(g["p"]) == 20000
    s, out, support, (X, status), (n, p, kmax) = cox_full
    # the vectors belong to exactly these inputs
    assert float(np.sum(status)) == float(g["status_sum"])
    assert np.array_equal(np.array([X[0, 0], X[n // 2, p // 2], X[n - 1, p - 1]]), g["x_probe"])
    pre = _Prefix(g)
    nfit = assert_matches_golden(out["trace"], pre, "configs[4] at full size, 1-pass score", beta_rtol=1e-5, metric_rtol=1e-7)
    assert nfit == 20
    s.trace_enable(False)
    for chains in (1, 3, 0):
        s.set_kpath_chains(chains)
        fast = s.sequential_path(np.arange(1, 41), ic_type=3)
        assert_untraced_path_matches_golden(fast, pre, X, 3, "configs[4] full size untraced, chains=%d" % chains,
                                            beta_rtol=1e-5, metric_rtol=1e-7)
        if chains == 3:
            assert s.counters()["kpath_chains_last_path"] == 3
    s.set_kpath_chains(0)
    hooks(monkeypatch, cox_score="2pass")  # read when a session is created
    with gpu.Session(X, status, data_type=3, model_type=4) as s2:
        s2.trace_enable(True)
        two = s2.sequential_path(np.arange(1, 21), ic_type=3)
    assert_matches_golden(two["trace"], pre, "configs[4] at full size, 2-pass score", beta_rtol=1e-5, metric_rtol=1e-7)


def test_cox_config_properties_at_full_size(gpu, cox_full, monkeypatch):
    s, out, support, (X, status), (n, p, kmax) = cox_full
    fits = out["trace"]["fits"]
    assert len(fits) == kmax and [f["T0"] for f in fits] == list(range(1, kmax + 1))
    assert_fits_stop_on_a_repeat(fits)
    # CoxMetric::ic, src/Metric.h:616-675: -2 loglik + log(p) log(log n) T0
    c = np.log(p) * np.log(np.log(n))
    np.testing.assert_allclose(out["cand_ic"], out["cand_train_loss"] + c * np.arange(1, kmax + 1), rtol=1e-12)
    assert np.all(np.diff(out["cand_train_loss"][:75]) < 0)
    # the 75 planted variables are recovered exactly at k = 75 and GIC selects that model
    assert np.array_equal(np.sort(out["cand_support"][74][:75]), support)
    assert out["best_T0"] == 75 and np.array_equal(np.nonzero(out["beta"])[0], support)
    # a second run of the same path: bitwise the same; the two-pass form of the score walks the same path
    s.trace_enable(False)
    again = s.sequential_path(np.arange(1, kmax + 1), ic_type=3)
    for k in ("cand_support", "cand_beta", "cand_ic", "cand_iters"):
        assert np.array_equal(out[k], again[k]), k
    hooks(monkeypatch, cox_score="2pass")  # read when a session is created
    with gpu.Session(X, status, data_type=3, model_type=4) as s2:
        two = s2.sequential_path(np.arange(1, 41), ic_type=3)
    assert np.array_equal(two["cand_support"], out["cand_support"][:40, :40])
    assert np.array_equal(two["cand_iters"], out["cand_iters"][:40])
    np.testing.assert_allclose(two["cand_ic"], out["cand_ic"][:40], rtol=1e-9)
    np.testing.assert_allclose(two["cand_beta"], out["cand_beta"][:40, :40], rtol=1e-7, atol=1e-10)
