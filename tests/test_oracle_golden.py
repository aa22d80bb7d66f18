"""CPU: the plain-C oracle replays the committed golden vectors (generated from the compiled reference by
tests/golden/make_golden.py): identical active sets at every PDAS iteration of every fit, coefficients,
loss and IC values within 1e-8."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import cases  # noqa: E402
from helpers import assert_same_trace  # noqa: E402
from oracle import port_ctypes as P  # noqa: E402

CASES = cases.all_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    X, y, kw = CASES[name]
    want = cases.load_golden(name)
    got = P.trace(X, y, **kw)
    assert_same_trace(got, want, beta_rtol=1e-8, what=name)
    np.testing.assert_allclose(got["beta"], want["beta"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-9, atol=1e-11)


SCR = cases.screening_cases()


@pytest.mark.parametrize("name", sorted(SCR))
def test_oracle_screening_matches_golden(name):
    """screening() + bessCpp's un-screening of beta (src/screening.cpp:26-105, src/bess.cpp:57-61, 186-209)."""
    X, y, ss, kw = SCR[name]
    want = cases.load_screening_golden(name)
    got = P.trace_screened(X, y, ss, **kw)
    assert np.array_equal(got["screening_A"], want["A"])
    np.testing.assert_allclose(got["beta"], want["beta"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose([got["coef0"], got["train_loss"], got["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-9, atol=1e-11)


def test_known_answers_from_survey():
    """Values recorded independently in SURVEY.md section 8c (prostate, README example)."""
    g = cases.load_golden("prostate_one_k3")
    assert list(np.nonzero(g["beta"])[0]) == [0, 1, 4]
    np.testing.assert_allclose(g["beta"][[0, 1, 4]], [0.52585188, 0.66176991, 0.66566656], rtol=1e-7)
    assert abs(g["coef0"] + 0.77715664) < 1e-7 and abs(g["ic"] + 61.691791663411266) < 1e-9
    g = cases.load_golden("readme_seq5")
    assert list(np.nonzero(g["beta"])[0]) == [0, 1, 2, 3, 4]
    assert abs(g["ic"] - 84.324826013868) < 1e-9 and abs(g["train_loss"] - 1.1184303414093817) < 1e-12
    g = cases.load_golden("prostate_nonorm")  # quirk q7: no intercept without normalisation
    assert list(np.nonzero(g["beta"])[0]) == [1, 2, 6] and g["coef0"] == 0.0


def test_oracle_building_blocks():
    rng = np.random.default_rng(0)
    s = rng.standard_normal(500) ** 2
    k = 17
    want = np.sort(np.argsort(-s, kind="stable")[:k])
    assert np.array_equal(P.max_k(s, k), want)
    s2 = np.array([1.0, 3.0, 3.0, 3.0, 0.5])
    # ties: whatever libstdc++'s nth_element leaves in the first k positions (restated move by move in the oracle; on
    # this input its partition keeps the first and the LAST of the three tied indices)
    assert list(P.max_k(s2, 2)) == [1, 3]
    a = rng.standard_normal((60, 12))
    g = a.T @ a
    b = rng.standard_normal(12)
    np.testing.assert_allclose(P.sym_solve(g, b), np.linalg.solve(g, b), rtol=1e-10)


def test_oracle_rejects_bad_arguments():
    X, y, _ = CASES["lm_seq"]
    with pytest.raises(ValueError):
        P.trace(X, y, sequence=[X.shape[1] + 1])
    with pytest.raises(ValueError):
        P.trace(X, y, is_cv=True, K=5, cv_fold_id=None, sequence=[1])


def test_max_k_heap_select_branch_matches_the_reference_golden():
    """std::nth_element runs out of its depth limit on score vectors a PDAS iteration really produces (63 distinct values,
    k = 10) and falls to __heap_select; with EQUAL scores at the boundary the selected set is what its heap moves leave
    in front.  Golden selections: the compiled reference's max_k (tests/golden/make_heap_select.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "heap_select_ties.npz"))
    before = P.nth_heap_selects()
    for sc, k, want in zip(g["scores"], g["k"], g["selected"]):
        assert np.array_equal(P.max_k(sc, int(k)), want[:k]), int(k)
    assert P.nth_heap_selects() - before == len(g["k"])  # every vector takes the branch
