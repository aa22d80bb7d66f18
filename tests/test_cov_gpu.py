"""GPU: the two evaluations of the LM score pass -- streaming (score_mode 1: X is read at every PDAS iteration) and
covariance updates (score_mode 2: cached Gram columns, X read only when a column is new) -- walk through exactly
the same active sets as the oracle, on warm and cold paths, CV folds, a cold start that overflows a slot's panel,
a cache small enough to be restarted, and sparsity levels beyond the register-resident solver."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import cases  # noqa: E402
from helpers import assert_same_trace, hooks  # noqa: E402
from test_lm_gpu import run_gpu  # noqa: E402
from oracle import port_ctypes as P  # noqa: E402
from bess_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = cases.all_cases()
LM_GOLD = [n for n in sorted(GOLD) if GOLD[n][2].get("model_type", 1) == 1 and "g_index" not in GOLD[n][2]]


def both_modes(gpu, X, y, kw, want, what):
    out = {}
    for mode in (1, 2):
        got = run_gpu(gpu, X, y, dict(kw, score_mode=mode))
        assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="%s mode %d" % (what, mode))
        sup = np.nonzero(want["beta"])[0]
        assert np.array_equal(np.nonzero(got["beta"])[0], sup)
        np.testing.assert_allclose(got["beta"][sup], want["beta"][sup], rtol=1e-6)
        np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                                   [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-7, atol=1e-9)
        out[mode] = got
    return out


@pytest.mark.parametrize("name", LM_GOLD)
def test_both_score_modes_match_reference_golden(gpu, name):
    X, y, kw = GOLD[name]
    both_modes(gpu, X, y, kw, cases.load_golden(name), name)


def test_cold_start_overflows_the_slot_panel(gpu):
    """No warm start at k = 100..104: the first iteration of every fit needs > 64 uncached columns."""
    X, y, _, _ = synth.make_lm(1500, 2000, 30)
    kw = dict(ic_type=3, sequence=[100, 101, 104], is_warm_start=False)
    both_modes(gpu, X, y, kw, P.trace(X, y, **kw), "cold")


def test_small_cache_is_restarted(gpu, monkeypatch):
    X, y, _, _ = synth.make_lm(1200, 3000, 25)
    kw = dict(ic_type=3, sequence=np.arange(1, 61))
    want = P.trace(X, y, **kw)
    hooks(monkeypatch, cov_cap="160")
    got = run_gpu(gpu, X, y, dict(kw, score_mode=2))
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="small cache")
    kw = dict(is_cv=True, K=3, cv_fold_id=synth.make_cv_folds(1200, 3), path_type=2, s_min=1, s_max=60)
    got = run_gpu(gpu, X, y, dict(kw, score_mode=2))
    assert_same_trace(got["trace"], P.trace(X, y, **kw), beta_rtol=1e-6, what="small cache gs cv")


def test_large_sparsity_levels_in_covariance_mode(gpu):
    X, y, _, _ = synth.make_lm(3000, 1200, 40)
    kw = dict(ic_type=3, sequence=[250, 300, 301, 420])
    both_modes(gpu, X, y, kw, P.trace(X, y, **kw), "large k")


def test_wide_matrix_two_level_topk(gpu):
    """p > 32768: the selection kernel runs in two levels and the cache lookup is its own launch."""
    X, y, _, _ = synth.make_lm(600, 40000, 12)
    kw = dict(ic_type=3, sequence=np.arange(1, 21))
    both_modes(gpu, X, y, kw, P.trace(X, y, **kw), "wide")


def test_chained_fits_on_a_lambda_grid_and_short_paths(gpu):
    """The chained warm-start path (snake order over a lambda grid, max_iter = 1 and 2, a single candidate)."""
    X, y, _, _ = synth.make_lm(900, 400, 9)
    for kw in (dict(ic_type=3, sequence=np.arange(1, 13), lambda_seq=[0.0, 0.02, 0.2, 1.0]),
               dict(ic_type=3, sequence=np.arange(1, 13), max_iter=1),
               dict(ic_type=3, sequence=np.arange(1, 13), max_iter=2),
               dict(ic_type=4, sequence=[7])):
        both_modes(gpu, X, y, kw, P.trace(X, y, **kw), "chain %r" % (sorted(kw),))


def test_conjugate_gradients_hand_ill_conditioned_systems_to_cholesky(gpu, monkeypatch):
    """Nearly collinear columns: the CG solve cannot reach its residual target and parks the fit; the Cholesky
    kernel finishes the slot.  Same active sets as the oracle either way; the test hook cov_solver=chol never uses CG."""
    rng = np.random.default_rng(11)
    n, p = 1500, 300
    z = rng.standard_normal((n, 6))
    X = np.repeat(z, 50, axis=1) + 1e-4 * rng.standard_normal((n, p))  # 6 clusters of 50 almost equal columns
    y = X[:, 0] - 2 * X[:, 60] + 1.5 * X[:, 130] + rng.standard_normal(n)
    kw = dict(ic_type=3, sequence=np.arange(1, 25))
    want = P.trace(X, y, **kw)
    s = gpu.Session(X, y, score_mode=2)
    s.trace_enable(True)
    got = s.sequential_path(kw["sequence"], (0.0,), 3, False)
    fell_back = s.counters()["cg_fallbacks"]
    s.close()
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="collinear cg")
    assert fell_back > 0
    hooks(monkeypatch, cov_solver="chol")
    s = gpu.Session(X, y, score_mode=2)
    s.trace_enable(True)
    got = s.sequential_path(kw["sequence"], (0.0,), 3, False)
    assert s.counters()["cg_fallbacks"] == 0
    s.close()
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="collinear chol")


@pytest.mark.parametrize("knob", ["chain=0", "publish=0", "cov_solver=chol", "cg_layout=tiles", "fuse=0", "cov_cs=24",
                                  "defer_publish=0", "fuse_sel=0", "cv_side_by_side=0"])
def test_runtime_knobs_do_not_change_results(gpu, monkeypatch, knob):
    """Every optional mechanism of the covariance form has a fallback the library keeps (the unfused launch sequence, the
    Cholesky behind the conjugate gradients, ...); forced through its test hook, the candidates, their supports and their
    ICs stay the same."""
    X, y, _, _ = synth.make_lm(1500, 2500, 15)
    seq = np.arange(1, 41)
    with gpu.Session(X, y, score_mode=2) as s:
        base = s.sequential_path(seq, ic_type=3)
    name, val = knob.split("=")
    hooks(monkeypatch, **{name: val})
    with gpu.Session(X, y, score_mode=2) as s:
        alt = s.sequential_path(seq, ic_type=3)
        again = s.sequential_path(seq, ic_type=3)
    for o in (alt, again):
        assert np.array_equal(o["cand_support"], base["cand_support"])
        np.testing.assert_allclose(o["cand_ic"], base["cand_ic"], rtol=1e-10)
        np.testing.assert_allclose(o["cand_beta"], base["cand_beta"], rtol=1e-8, atol=1e-12)


def test_score_mode_argument(gpu):
    X, y, _, _ = synth.make_logistic(300, 40, 3)
    with pytest.raises(gpu.BessxError) as e:
        gpu.Session(X, y, data_type=2, model_type=2, score_mode=2)
    assert e.value.code == 1
    with pytest.raises(gpu.BessxError) as e:
        gpu.Session(X, y, score_mode=7)
    assert e.value.code == 1


@pytest.mark.parametrize("n,p,K", [(1000, 300, 5), (777, 150, 4)])
def test_cv_row_sets_share_their_fills(gpu, monkeypatch, n, p, K):
    """Cross-validation in the covariance form: one unmasked pass over a fold-major copy of X fills the Gram-column
    caches of ALL K + 1 row sets (the slab partials of every fold but k sum to fold k's training rows).  Same path as with
    one masked pass per row set (test hook cv_shared=0), far fewer passes over X, and the oracle's path fit by fit."""
    X, y, _, _ = synth.make_lm(n, p, 10, seed=n)
    fold = synth.make_cv_folds(n, K, seed=3)
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, cv_shared=mode)
        with gpu.Session(X, y, score_mode=2) as s:
            s.set_cv(K, fold)
            s.trace_enable(True)
            outs[mode] = (s.gs_path(1, 30, ic_type=3, is_cv=True), s.counters()["passes_over_X"],
                          s.sequential_path(np.arange(1, 13), [0.0, 0.03], ic_type=3, is_cv=True))
    a, b = outs["1"], outs["0"]
    assert a[1] < b[1] / 2, (a[1], b[1])  # far fewer 32-column passes over X
    for i in (0, 2):
        assert np.array_equal(a[i]["cand_support"], b[i]["cand_support"]) and a[i]["n_fits"] == b[i]["n_fits"]
        np.testing.assert_allclose(a[i]["cand_ic"], b[i]["cand_ic"], rtol=1e-10)
        np.testing.assert_allclose(a[i]["beta"], b[i]["beta"], rtol=1e-9, atol=1e-13)
    want = P.trace(X, y, ic_type=3, is_cv=True, K=K, cv_fold_id=fold, path_type=2, s_min=1, s_max=30)
    assert_same_trace(a[0]["trace"], want, what="shared CV fills")


def test_panel_kernel_dp_against_numpy_and_the_default_kernels(gpu, monkeypatch):
    """k_cov_panel_dp (round 5: one 8-wave workgroup per compute unit over 128 streamed columns, two LDS tiles, one
    barrier per chunk; the default, test hook panel=lds selects k_cov_panel_lds2 / _pair): its Gram columns against NumPy X^T X_S -- one group and a pair per pass, a
    column count that is no multiple of 128, a fold mask -- and a path run with it against the default kernels' path."""
    X, y, _, _ = synth.make_lm(3000, 1100, 12, seed=5)  # 1100 columns: the last workgroup's second half is partly empty
    n, p = X.shape
    Xc = X - X.mean(axis=0)
    Xn = np.sqrt(float(n)) * Xc / np.sqrt((Xc * Xc).sum(axis=0))
    cols = ((np.arange(128) * 29 + 7) % p).astype(np.int32)
    want = Xn.T @ Xn[:, cols]
    seq = np.arange(1, 41)
    hooks(monkeypatch, panel="lds")  # (the kernels of rounds 2-4; k_cov_panel_dp is the default since round 5)
    with gpu.Session(X, y, score_mode=2) as s0:
        ref = s0.sequential_path(seq, ic_type=3)
        fold = synth.make_cv_folds(n, 4)
        s0.set_cv(4, fold)
        ref_cv = s0.gs_path(1, 30, ic_type=3, is_cv=True)
    hooks(monkeypatch, panel="dp")
    with gpu.Session(X, y, score_mode=2) as s:
        s.cov_prefill_begin(cols)
        s.cov_prefill_compute(0, 1)
        s.cov_prefill_compute(1, 1)
        s.cov_prefill_compute(2, 2)
        got = s.cov_prefill_export(0, 4).reshape(128, p).T
        s.cov_prefill_end()
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
        out = s.sequential_path(seq, ic_type=3)
        assert np.array_equal(out["cand_support"], ref["cand_support"]) and np.array_equal(out["cand_iters"], ref["cand_iters"])
        np.testing.assert_allclose(out["cand_ic"], ref["cand_ic"], rtol=1e-10)
        s.set_cv(4, fold)  # the fold-major copy: masked / multi-row-set fills through the same kernel
        cv = s.gs_path(1, 30, ic_type=3, is_cv=True)
        assert cv["best_T0"] == ref_cv["best_T0"] and np.array_equal(cv["cand_T0"], ref_cv["cand_T0"])
        np.testing.assert_allclose(cv["cand_ic"], ref_cv["cand_ic"], rtol=1e-10)
