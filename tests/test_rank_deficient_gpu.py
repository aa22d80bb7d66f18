"""Rank-deficient restricted fits: exactly dependent columns selected together (a duplicated or mirrored variable and
its original), systems with more columns than independent rows.  The reference solves them with pivoted
factorisations (ColPivHouseholderQR of the LM Gram, src/Algorithm.h:1131-1135; Eigen's LDLT for the IRLS / Newton
systems); the fast solvers here (Cholesky in registers, conjugate gradients) stand back -- k_chol's pivot test, the
dependent-pair flag of the Gram column cache -- and sym_pivoted_solve (LDL^T with diagonal pivoting, a collapsed pivot
gives a zero coefficient) takes over.

Golden vectors: tests/golden/rank_deficient_ref.npz from the COMPILED REFERENCE (make_rank_deficient.py).  Where the
reference itself is deterministic on such a system (one of the dependent coefficients exactly 0: its factorisation saw
the rank deficiency) every PDAS iteration's active set and every coefficient must agree; where its outcome is decided
by rounding (both copies non-zero: `deterministic` = 0 in the file) only a finite, error-free run is asked for."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_rank_deficient import cases  # noqa: E402  (inputs only: seeded, no reference needed)
from helpers import assert_same_trace  # noqa: E402
from test_lm_gpu import run_gpu  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "rank_deficient_ref.npz")


def _golden_trace(g, name):
    iters, betas, pos = [], [], 0
    fits = []
    lens = list(g[name + "/iter_len"])
    A, B, C = g[name + "/A_flat"], g[name + "/beta_flat"], g[name + "/coef0_flat"]
    it = 0
    for T0, tn, ni in zip(g[name + "/fit_T0"], g[name + "/fit_train_n"], g[name + "/fit_iters"]):
        f = {"T0": int(T0), "train_n": int(tn), "iters": [], "betas": [], "coef0s": []}
        for _ in range(int(ni)):
            L = int(lens[it])
            f["iters"].append(A[pos:pos + L])
            f["betas"].append(B[pos:pos + L])
            f["coef0s"].append(float(C[it]))
            pos += L
            it += 1
        fits.append(f)
    return {"fits": fits, "ic_calls": g[name + "/ic_calls"], "loss_calls": g[name + "/loss_calls"],
            "beta": g[name + "/beta"]}


def test_pivoted_solve_on_singular_and_indefinite_systems(gpu):
    rng = np.random.default_rng(0)
    for m in (6, 40, 100, 200, 254):
        B = rng.standard_normal((300, m))
        B[:, m // 2] = B[:, 1]      # a duplicated column
        B[:, m - 1] = -B[:, 3]      # and a mirror image
        A = B.T @ B
        b = A @ rng.standard_normal(m)
        x = gpu.op_chol_solve(A, b)
        assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
        # a basic solution: one copy of each dependent pair carries the coefficient, the other is exactly 0
        assert (x[m // 2] == 0.0) != (x[1] == 0.0) and (x[m - 1] == 0.0) != (x[3] == 0.0)
    A = rng.standard_normal((30, 30))
    A = A + A.T  # indefinite
    b = rng.standard_normal(30)
    np.testing.assert_allclose(gpu.op_chol_solve(A, b), np.linalg.solve(A, b), rtol=1e-9)
    # more unknowns than independent rows: a consistent singular system, any exact solution
    B = rng.standard_normal((20, 60))
    A = B.T @ B
    b = B.T @ rng.standard_normal(20)
    x = gpu.op_chol_solve(A, b)
    assert np.all(np.isfinite(x)) and np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("name", ["lm_seq", "lm_gs", "lm_cold", "lm_cv", "logistic_seq", "cox_seq"])
def test_paths_with_duplicated_true_columns(gpu, name):
    assert os.path.exists(GOLD), "golden file %s is missing (a committed fixture)" % GOLD
    g = np.load(GOLD)
    X, y, kw, _ = cases()[name]
    modes = (0, 1, 2) if name.startswith("lm") else (0,)
    for mode in modes:
        got = run_gpu(gpu, X, y, dict(kw, score_mode=mode) if name.startswith("lm") else kw)  # no BESSX_ERR_NUMERIC
        for f in got["trace"]["fits"]:
            for b in f["betas"]:
                assert np.all(np.isfinite(b))
        if int(g[name + "/deterministic"]):
            assert len(got["trace"]["fits"]) == len(g[name + "/fit_T0"])
            want = _golden_trace(g, name)
            assert_same_trace(got["trace"], want, beta_rtol=1e-5 if name == "cox_seq" else 1e-6,
                              what="%s score_mode %d vs the compiled reference" % (name, mode))
            assert np.array_equal(np.nonzero(got["beta"])[0], np.nonzero(want["beta"])[0])


def test_wide_groups_on_few_rows_no_longer_error(gpu):
    """Sum of the selected groups' sizes > training rows (tests/fuzz_parity.py used to exclude this): the normal
    equations are singular but consistent; the reference's QR returns SOME solution of them (quotients of rounding
    errors, not reproducible), this build the basic solution of the pivoted factorisation -- a fit with zero residual
    either way."""
    rng = np.random.default_rng(4)
    n, p = 24, 60
    X = rng.standard_normal((n, p))
    y = X[:, :3] @ np.array([2.0, -1.0, 1.5]) + 0.1 * rng.standard_normal(n)
    gi = np.arange(0, p, 10, dtype=np.int32)
    with gpu.Session(X, y, algorithm_type=2, g_index=gi) as s:
        out = s.sequential_path(np.array([2, 3, 4]), ic_type=3)
    assert np.all(np.isfinite(out["cand_beta"])) and np.all(np.isfinite(out["cand_ic"]))
    with gpu.Session(X, y) as s:
        out = s.sequential_path(np.array([22, 25, 30]), ic_type=3)
    assert np.all(np.isfinite(out["cand_beta"]))
    assert out["cand_train_loss"][-1] <= 1e-16 * np.var(y)  # 30 columns on 23 centred rows: an exact fit
