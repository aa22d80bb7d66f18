"""GPU vs the committed golden vectors of the compiled reference (tests/golden/ref_small.npz): every LM /
logistic / Poisson (and, once built, Cox) case -- active set of every PDAS iteration bit-exact, coefficients
within 1e-6 relative (north_star tolerance), IC and loss values within 1e-8; plus the drop-in entry points
(pywrap_bess through ctypes and through the pybind11 module, and the estimator classes)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import cases  # noqa: E402
from helpers import assert_same_trace  # noqa: E402
from test_lm_gpu import run_gpu  # noqa: E402

pytestmark = pytest.mark.gpu
CASES = cases.all_cases()
BUILT = [n for n in sorted(CASES) if CASES[n][2].get("model_type", 1) in (1, 2, 3, 4)]


@pytest.mark.parametrize("name", BUILT)
def test_gpu_matches_reference_golden(gpu, name):
    X, y, kw = CASES[name]
    want = cases.load_golden(name)
    got = run_gpu(gpu, X, y, kw)
    assert_same_trace(got["trace"], want, beta_rtol=1e-6, what=name)
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(got["beta"])[0], sup)
    np.testing.assert_allclose(got["beta"][sup], want["beta"][sup], rtol=1e-6)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-7, atol=1e-9)
    if kw.get("path_type") == 3:
        assert abs(got["lambda"] - want["lambda"]) <= 1e-12 * want["lambda"]


def test_pywrap_routes_l0l2_search_to_the_powell_path(gpu):
    """bessCpp sends path_type 2 with algorithm_type 5 to pgs_path (src/bess.cpp:174-180)."""
    X, y, kw = CASES["lm_powell_gs"]
    want = cases.load_golden("lm_powell_gs")
    n, p = X.shape
    r = gpu.pywrap_bess(X, y, 1, np.ones(n), True, 5, 1, 20, 0, 2, True, kw["ic_type"], False, 5, np.arange(p),
                        np.ones(n), [1], [0.0], kw["s_min"], kw["s_max"], 0, 1e-4, kw["lambda_min"], kw["lambda_max"],
                        kw["nlambda"], False, 1, kw["powell_path"], [], 0.0, p)
    np.testing.assert_allclose(r[0], want["beta"], rtol=1e-6, atol=1e-12)
    assert abs(r[3][0] - want["ic"]) < 1e-7 * abs(want["ic"])


def _pywrap_args(X, y, data_type=1, model_type=1, sequence=(3,), ic_type=3, path_type=1, s_min=0, s_max=0):
    n, p = X.shape
    return (X, y, data_type, np.ones(n), True, 1, model_type, 20, 0, path_type, True, ic_type, False, 5, np.arange(p),
            np.ones(n), list(sequence), [0.0], s_min, s_max, 0, 1e-4, 0.0, 0.0, 100, False, 1, 1, [], 0.0, p, 1, 1, 1,
            1, 1, 1, p)


def test_pywrap_bess_ctypes_and_pybind(gpu):
    X, y = cases.prostate()
    want = cases.load_golden("prostate_one_k3")
    from bess_amd import _cbess
    for fn in (gpu.pywrap_bess, _cbess.pywrap_bess):
        r = fn(*_pywrap_args(X, y))
        beta, coef0, loss, ic, _, _, _, _, a_out, l_out = r
        np.testing.assert_allclose(beta, want["beta"], rtol=1e-6, atol=1e-12)
        assert abs(coef0[0] - want["coef0"]) < 1e-7 and abs(ic[0] - want["ic"]) < 1e-7
        assert abs(loss[0] - want["train_loss"]) < 1e-9
        assert list(a_out[:3]) == [0, 1, 4] and l_out == 3
    r = gpu.pywrap_bess(*_pywrap_args(X, y, path_type=2, s_min=1, s_max=8))
    assert list(np.nonzero(r[0])[0]) == [0, 1, 4]


def test_estimator_classes_readme_example(gpu):
    import bess_amd
    X, y = cases.readme_lm()
    want = cases.load_golden("readme_seq5")
    m = bess_amd.PdasLm(path_type="seq", sequence=[5])
    m.fit(X=X, y=y)
    assert list(np.nonzero(m.beta)[0]) == [0, 1, 2, 3, 4]
    np.testing.assert_allclose(m.beta, want["beta"], rtol=1e-6, atol=1e-12)
    assert abs(m.coef0[0] - want["coef0"]) < 1e-8 and abs(m.ic[0] - want["ic"]) < 1e-8
    np.testing.assert_allclose(m.predict(X), X @ want["beta"] + want["coef0"], rtol=1e-6)
    m = bess_amd.PdasLm(path_type="seq")  # default sequence 1..min(p, n/log n)
    m.fit(X=X, y=y)
    assert list(np.nonzero(m.beta)[0]) == [0, 1, 2, 3, 4]
    m = bess_amd.PdasLm(path_type="pgs", s_max=20)
    m.fit(X=X, y=y)
    assert list(np.nonzero(m.beta)[0]) == [0, 1, 2, 3, 4] and abs(m.ic[0] - want["ic"]) < 1e-8


def test_group_estimator_class(gpu):
    """GroupPdasLm with a `group` label per column (python/bess/linear.py:238-253 turns labels into g_index)."""
    import bess_amd
    X, y, _, _, gi = cases.group_data()
    want = cases.load_golden("grp_lm_seq")
    labels = np.zeros(X.shape[1], dtype=int)
    for g, lo in enumerate(gi):
        labels[lo:] = g
    m = bess_amd.GroupPdasLm(path_type="seq", sequence=list(range(1, 9)), ic_type="gic")
    m.fit(X, y, group=labels)
    np.testing.assert_allclose(m.beta, want["beta"], rtol=1e-6, atol=1e-12)
    assert abs(m.ic[0] - want["ic"]) < 1e-7 * abs(want["ic"])


def test_group_cox_estimator_class(gpu):
    """GroupPdasCox: y = (time, status); the wrapper sorts the rows by time (python/bess/linear.py:257-263)."""
    import bess_amd
    X, status, gi = cases.cox_group_data()
    want = cases.load_golden("grp_cox_seq")
    labels = np.zeros(X.shape[1], dtype=int)
    for g, lo in enumerate(gi):
        labels[lo:] = g
    n = X.shape[0]
    perm = np.random.default_rng(3).permutation(n)  # hand the rows over unsorted, with their times
    y = np.column_stack([np.arange(n, dtype=float), status])[perm]
    m = bess_amd.GroupPdasCox(path_type="seq", sequence=list(range(1, 8)), ic_type="gic")
    m.fit(X[perm], y, group=labels)
    np.testing.assert_allclose(m.beta, want["beta"], rtol=1e-6, atol=1e-12)
    assert abs(m.ic[0] - want["ic"]) < 1e-7 * abs(want["ic"])


def test_unsupported_and_invalid_requests_fail_loudly(gpu):
    X, y = cases.prostate()
    a = list(_pywrap_args(X, y))
    bad = list(a)
    bad[25], bad[26] = True, 9  # is_screening with screening_size > p
    with pytest.raises(gpu.BessxError) as e:
        gpu.pywrap_bess(*bad)
    assert e.value.code == 1
    bad = list(a)
    bad[14] = np.array([0, 4, 2, 6])  # group index must increase
    with pytest.raises(gpu.BessxError) as e:
        gpu.pywrap_bess(*bad)
    assert e.value.code == 1
    Xw = np.random.default_rng(0).standard_normal((60, 40))
    wide = list(_pywrap_args(Xw, Xw[:, 0]))
    wide[14] = np.array([0, 20])  # two groups of 20 columns: wider than the register-resident group path, accepted
    wide[16] = [1]
    r = gpu.pywrap_bess(*wide)
    assert np.count_nonzero(r[0]) == 20 and np.all(r[0][:20] != 0)  # y = first column: the first group is chosen
    bad = list(a)
    bad[16] = [9]  # sparsity level > p
    with pytest.raises(gpu.BessxError) as e:
        gpu.pywrap_bess(*bad)
    assert e.value.code == 1
