"""GPU: sure independence screening (screening(), src/screening.cpp:26-105; bessCpp src/bess.cpp:57-61, 186-209)
against the compiled reference's golden vectors and, iteration by iteration, against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import cases  # noqa: E402
from helpers import assert_same_trace  # noqa: E402
from test_lm_gpu import run_gpu  # noqa: E402
from oracle import port_ctypes as P  # noqa: E402
from bess_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
SCR = cases.screening_cases()


@pytest.mark.parametrize("name", sorted(SCR))
def test_screened_path_matches_reference_golden(gpu, name):
    X, y, ss, kw = SCR[name]
    want = cases.load_screening_golden(name)
    got = run_gpu(gpu, X, y, dict(kw, screening_size=ss))
    assert np.array_equal(got["screening_A"], want["A"])  # kept columns: bit-exact
    sup = np.nonzero(want["beta"])[0]
    assert got["beta"].size == X.shape[1] and np.array_equal(np.nonzero(got["beta"])[0], sup)
    np.testing.assert_allclose(got["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-7, atol=1e-9)
    # every PDAS iteration on the kept columns, against the oracle (kept-column numbering on both sides)
    ref = P.trace_screened(X, y, ss, **kw)
    assert_same_trace(got["trace"], ref, beta_rtol=1e-6, what=name)


def test_screening_through_pywrap_bess(gpu):
    X, y, ss, kw = SCR["scr_lm_seq"]
    want = cases.load_screening_golden("scr_lm_seq")
    n, p = X.shape
    r = gpu.pywrap_bess(X, y, 1, np.ones(n), True, 1, 1, 20, 0, 1, True, kw["ic_type"], False, 5, np.arange(p),
                        np.ones(n), list(kw["sequence"]), [0.0], 1, 1, 0, 1e-4, 0.0, 0.0, 100, True, ss, 1, [], 0.0, p)
    np.testing.assert_allclose(r[0], want["beta"], rtol=1e-6, atol=1e-12)
    assert abs(r[3][0] - want["ic"]) < 1e-7 * abs(want["ic"])
    sup = np.nonzero(want["beta"])[0]
    assert list(r[8][:sup.size]) == list(sup)  # A_out in the caller's column numbering


def test_screening_cv_and_wide_matrix(gpu):
    """CV after screening, and a wider matrix than the golden cases (p = 5000 -> 300 kept)."""
    X, y, _, _ = synth.make_lm(800, 5000, 10)
    fold = synth.make_cv_folds(800, 4)
    kw = dict(is_cv=True, K=4, cv_fold_id=fold, sequence=np.arange(1, 16))
    ref = P.trace_screened(X, y, 300, **kw)
    got = run_gpu(gpu, X, y, dict(kw, screening_size=300))
    assert np.array_equal(got["screening_A"], ref["screening_A"])
    assert_same_trace(got["trace"], ref, beta_rtol=1e-6, what="scr_cv")
    np.testing.assert_allclose(got["beta"], ref["beta"], rtol=1e-6, atol=1e-12)
    Xl, yl, _, _ = synth.make_logistic(1500, 2000, 6)
    kw = dict(data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, 9))
    ref = P.trace_screened(Xl, yl, 100, **kw)
    got = run_gpu(gpu, Xl, yl, dict(kw, screening_size=100))
    assert np.array_equal(got["screening_A"], ref["screening_A"])
    assert_same_trace(got["trace"], ref, beta_rtol=1e-6, what="scr_logit_wide")


def test_screening_argument_errors(gpu):
    X, y, _, _ = synth.make_lm(200, 50, 3)
    with pytest.raises(gpu.BessxError) as e:
        gpu.Session(X, y, is_screening=True, screening_size=51)
    assert e.value.code == 1
    with pytest.raises(gpu.BessxError) as e:  # Poisson: undefined behaviour in the reference, refused here
        gpu.Session(X, np.abs(np.round(y)), data_type=2, model_type=3, is_screening=True, screening_size=10)
    assert e.value.code == 3
    Xl, yl, _, _ = synth.make_logistic(30, 50, 3)
    with pytest.raises(gpu.BessxError) as e:  # logit_fit on a group as wide as the sample: UB in the reference
        gpu.Session(Xl, yl, data_type=2, model_type=2, algorithm_type=2, is_screening=True, screening_size=1,
                    g_index=[0, 40])
    assert e.value.code == 3


def _group_index(p, seed, wide=False):
    rng = np.random.default_rng(seed)
    starts, c = [], 0
    while c < p:
        starts.append(c)
        c += int(rng.choice([1, 2, 3, 5, 20] if wide else [1, 2, 3, 4]))
    return np.array(starts, dtype=np.int32)


@pytest.mark.parametrize("wide", [False, True])
def test_lm_screening_with_groups(gpu, wide):
    """Screening with groups of size > 1 (LM): the kept GROUPS are the compiled reference's screening_A; the path then
    runs on the kept groups' columns exactly like the oracle run on that sub-matrix with the kept group index; the
    coefficients come back in the caller's column numbering (the reference misplaces them here, src/bess.cpp:195-198)."""
    from oracle import ref_ctypes as R
    n, p = 500, 240
    X, y, _, _ = synth.make_lm(n, p, 8, seed=17)
    gi = _group_index(p, 5, wide)
    N, keep_n = len(gi), len(gi) // 3
    sizes = np.diff(np.append(gi, p))
    always = [int(N - 2)]
    with gpu.Session(X, y, algorithm_type=2, g_index=gi, is_screening=True, screening_size=keep_n,
                     always_select=always) as s:
        groups, cols = s.screening_groups(), s.screening()
        s.trace_enable(True)
        got = s.sequential_path(np.arange(2, 6), ic_type=3)
    if R.available():
        assert np.array_equal(groups, R.screening_groups(X, y, None, 1, keep_n, gi, always))
    assert len(groups) == keep_n and always[0] in groups
    want_cols = np.concatenate([np.arange(gi[g], gi[g] + sizes[g]) for g in groups])
    assert np.array_equal(cols, want_cols)
    # the oracle on the kept columns with the kept group index and the re-ranked always_select
    new_gi = np.concatenate([[0], np.cumsum(sizes[groups])[:-1]]).astype(np.int32)
    al = [int(np.searchsorted(groups, always[0]))]
    want = P.trace(X[:, cols], y, algorithm_type=2, g_index=new_gi, ic_type=3, sequence=np.arange(2, 6),
                   always_select=al)
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="grouped screening")
    full = np.zeros(p)
    full[cols] = want["beta"]
    np.testing.assert_allclose(got["beta"], full, rtol=1e-6, atol=1e-12)  # the right columns of the ORIGINAL numbering
    # the R-facing entry reports the kept groups as screening_A
    r = gpu.bessCpp(X, y, 1, np.ones(n), True, 2, 1, 20, 2, 1, True, 3, False, 5, np.full(10, 2.0), np.arange(2, 6), [0.0],
                    1, 1, 10, 10.0, 0.0, 0.0, 100, True, keep_n, 1, gi, always, 1.1)
    assert np.array_equal(r["screening_A"], groups)
    np.testing.assert_allclose(r["beta"], full, rtol=1e-6, atol=1e-12)


def test_logistic_screening_with_groups(gpu):
    """logit_fit on whole groups (src/logistic.cpp:60-160; groups of at most 8 columns): kept groups = the compiled
    reference's, then the path on the kept columns like the oracle's on that sub-matrix."""
    from oracle import ref_ctypes as R
    n, p = 900, 120
    X, y, _, _ = synth.make_logistic(n, p, 6, seed=23)
    gi = _group_index(p, 9)
    N, keep_n = len(gi), len(gi) // 2
    sizes = np.diff(np.append(gi, p))
    kw = dict(data_type=2, model_type=2, algorithm_type=2)
    with gpu.Session(X, y, g_index=gi, is_screening=True, screening_size=keep_n, **kw) as s:
        groups, cols = s.screening_groups(), s.screening()
        s.trace_enable(True)
        got = s.sequential_path(np.arange(1, 5), ic_type=3)
    if R.available():
        assert np.array_equal(groups, R.screening_groups(X, y, None, 2, keep_n, gi))
    new_gi = np.concatenate([[0], np.cumsum(sizes[groups])[:-1]]).astype(np.int32)
    want = P.trace(X[:, cols], y, g_index=new_gi, ic_type=3, sequence=np.arange(1, 5), **kw)
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="grouped logistic screening")
    full = np.zeros(p)
    full[cols] = want["beta"]
    np.testing.assert_allclose(got["beta"], full, rtol=1e-6, atol=1e-12)


# kept groups of the compiled reference (oracle/_ref/libbess_ref.so, recorded in the build container with exactly these
# inputs; re-checked against it wherever it is present): groups of 1..20 columns, the wide ones beyond what the
# one-block-per-group kernels hold (logistic 8, Cox 4 columns)
WIDE_KEPT = {("logistic", False): [1, 6, 8, 10, 11, 12], ("logistic", True): [1, 4, 6, 8, 10, 11],
             ("cox", False): [0, 2, 5], ("cox", True): [0, 2, 5]}


@pytest.mark.parametrize("family", ["logistic", "cox"])
@pytest.mark.parametrize("weighted", [False, True])
def test_screening_with_wide_groups_of_the_iterative_families(gpu, family, weighted):
    """logit_fit / cox_fit on groups of any width (src/screening.cpp:42-63): narrow groups by the one-block kernels,
    wider ones by the solver's own IRLS / Newton chain on a sub-session of the group's columns (no weight floor; linear
    predictor clamped at 50) -- the kept GROUPS equal the compiled reference's screening_A, then the path on the kept
    columns equals the oracle's on that sub-matrix."""
    from oracle import ref_ctypes as R
    if family == "logistic":
        n, p = 900, 120
        X, y, _, _ = synth.make_logistic(n, p, 6, seed=23)
        gi, mt = _group_index(p, 9, wide=True), 2
        kw = dict(data_type=2, model_type=2, algorithm_type=2)
    else:
        n, p = 600, 90
        X, _, y, _, _ = synth.make_cox(n, p, 5, seed=31)
        gi, mt = _group_index(p, 13, wide=True), 4
        kw = dict(data_type=3, model_type=4, algorithm_type=2)
    N, keep_n = len(gi), len(gi) // 2
    sizes = np.diff(np.append(gi, p))
    assert sizes.max() == 20
    w = np.random.default_rng(4).uniform(0.5, 2, 900)[:n] if weighted else None
    always = [] if weighted else [int(N - 1)]
    with gpu.Session(X, y, g_index=gi, is_screening=True, screening_size=keep_n, always_select=always, weight=w, **kw) as s:
        groups, cols = s.screening_groups(), s.screening()
        s.trace_enable(True)
        got = s.sequential_path(np.arange(1, 4), ic_type=3)
    assert list(groups) == WIDE_KEPT[(family, weighted)]
    if R.available():
        assert np.array_equal(groups, R.screening_groups(X, y, w, mt, keep_n, gi, always))
    want_cols = np.concatenate([np.arange(gi[g], gi[g] + sizes[g]) for g in groups])
    assert np.array_equal(cols, want_cols)
    new_gi = np.concatenate([[0], np.cumsum(sizes[groups])[:-1]]).astype(np.int32)
    al = [int(np.searchsorted(groups, a)) for a in always]
    want = P.trace(X[:, cols], y, g_index=new_gi, ic_type=3, sequence=np.arange(1, 4), always_select=al, weight=w, **kw)
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="wide-group %s screening" % family)


def test_cox_screening_with_groups(gpu):
    """cox_fit on whole groups (src/coxph.cpp:42-108; groups of at most 4 columns): kept groups = the compiled
    reference's, then the path on the kept columns like the oracle's on that sub-matrix."""
    from oracle import ref_ctypes as R
    n, p = 600, 90
    X, _, st, _, _ = synth.make_cox(n, p, 5, seed=31)
    gi = _group_index(p, 13)
    N, keep_n = len(gi), len(gi) // 2
    sizes = np.diff(np.append(gi, p))
    always = [int(N - 1)]
    kw = dict(data_type=3, model_type=4, algorithm_type=2)
    with gpu.Session(X, st, g_index=gi, is_screening=True, screening_size=keep_n, always_select=always, **kw) as s:
        groups, cols = s.screening_groups(), s.screening()
        s.trace_enable(True)
        got = s.sequential_path(np.arange(2, 5), ic_type=3)
    if R.available():
        assert np.array_equal(groups, R.screening_groups(X, st, None, 4, keep_n, gi, always))
    assert len(groups) == keep_n and always[0] in groups
    new_gi = np.concatenate([[0], np.cumsum(sizes[groups])[:-1]]).astype(np.int32)
    al = [int(np.searchsorted(groups, always[0]))]
    want = P.trace(X[:, cols], st, g_index=new_gi, ic_type=3, sequence=np.arange(2, 5), always_select=al, **kw)
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what="grouped Cox screening")
    full = np.zeros(p)
    full[cols] = want["beta"]
    np.testing.assert_allclose(got["beta"], full, rtol=1e-6, atol=1e-12)


def test_lm_group_screening_with_dependent_columns_inside_a_group(gpu):
    """A duplicated and a mirrored column INSIDE groups: the group's marginal least-squares fit is rank-deficient; the
    reference's column-pivoted QR gives the dependent copy the coefficient 0 (src/screening.cpp:44-48) and so does the
    group solve here (k_group_lsq_score drops a column whose pivot collapses).  Kept groups = the compiled reference's
    (recorded with oracle/_ref/libbess_ref.so on exactly these inputs; re-checked against it where it is present)."""
    from oracle import ref_ctypes as R
    X, y, _, _ = synth.make_lm(200, 60, 4, seed=11)
    X = np.array(X)
    gi = np.arange(0, 60, 5, dtype=np.int32)
    X[:, 7] = X[:, 5]
    X[:, 23] = -X[:, 21]
    want = {4: [1, 2, 8, 9], 6: [1, 2, 5, 8, 9, 11], 9: [0, 1, 2, 4, 5, 6, 8, 9, 11]}
    for keep, groups in want.items():
        if R.available():
            assert list(R.screening_groups(X, y, None, 1, keep, gi)) == groups
        with gpu.Session(X, y, algorithm_type=2, g_index=gi, is_screening=True, screening_size=keep) as s:
            assert list(s.screening_groups()) == groups, keep
