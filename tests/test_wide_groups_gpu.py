"""GPU: group selection with groups wider than 16 columns (round-1 limit; the reference takes any width,
src/utilities.cpp:142-177, src/Algorithm.h:1112-1123).  Wide groups take the tiled moment kernel and the Cholesky
form of the sacrifice, || L^T b + L^{-1} d ||^2 with M = L L^T, which equals || Phi b + Phi^{-1} d ||^2 for
Phi = sqrtm(M).  Every PDAS iteration against the oracle (itself checked against the compiled reference on wide groups,
tests/test_oracle_vs_reference.py::test_wide_groups_random)."""
import numpy as np
import pytest

from bess_amd import synth
from test_glm_gpu import check
from test_oracle_vs_reference import _wide_groups

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(ic_type=3, sequence=np.arange(1, 6))),
    ("gs", dict(ic_type=3, path_type=2, s_min=1, s_max=6)),
    ("l0l2", dict(algorithm_type=3, ic_type=4, sequence=np.arange(1, 4), lambda_seq=[0.0, 0.05])),
    ("cv", dict(is_cv=True, K=4, sequence=np.arange(1, 4))),
])
def test_lm_wide_groups(gpu, name, kw):
    X, y, _, _ = synth.make_lm(600, 200, 6, seed=3)
    gi = _wide_groups(200, 5)
    assert np.max(np.diff(np.append(gi, 200))) > 16
    kw = dict(dict(algorithm_type=2), **kw, g_index=gi)
    if kw.get("is_cv"):
        kw["cv_fold_id"] = synth.make_cv_folds(600, 4)
    check(gpu, X, y, kw, "lm wide groups " + name)


def test_logistic_and_poisson_wide_groups(gpu):
    X, y, _, _ = synth.make_logistic(1200, 160, 5, seed=8)
    gi = _wide_groups(160, 9)
    check(gpu, X, y, dict(algorithm_type=2, g_index=gi, data_type=2, model_type=2, ic_type=3, sequence=np.arange(1, 5)),
          "logistic wide groups")
    rng = np.random.default_rng(4)
    Xp = rng.standard_normal((900, 90))
    b = np.zeros(90)
    b[[3, 30, 31, 60]] = [0.3, -0.3, 0.2, 0.25]
    yp = rng.poisson(np.exp(Xp @ b)).astype(float)
    gp = _wide_groups(90, 2)
    check(gpu, Xp, yp, dict(algorithm_type=2, g_index=gp, data_type=2, model_type=3, ic_type=3, sequence=np.arange(1, 4)),
          "poisson wide groups")


def test_cox_wide_groups(gpu):
    X, _, status, _, _ = synth.make_cox(700, 120, 5, seed=6)
    gi = _wide_groups(120, 7)
    check(gpu, X, status, dict(algorithm_type=2, g_index=gi, data_type=3, model_type=4, ic_type=3,
                               sequence=np.arange(1, 4)), "cox wide groups", beta_rtol=1e-5)


def test_one_very_wide_group(gpu):
    """a 300-column group among narrow ones: the selected columns exceed the register-resident solver too"""
    X, y, _, _ = synth.make_lm(2000, 400, 8, seed=12)
    gi = np.concatenate([np.arange(0, 100, 4), [100], np.arange(400 - 0, 400, 1)]).astype(np.int32)
    gi = np.concatenate([np.arange(0, 100, 4), np.arange(100, 400, 300)]).astype(np.int32)  # 25 groups of 4, one of 300
    check(gpu, X, y, dict(algorithm_type=2, g_index=gi, ic_type=3, sequence=np.arange(1, 4)), "one 300-column group")
