"""Cox PDAS on the GPU (two-pass suffix-scan score, device-gated Newton with step halving) against the plain-C
oracle.  Supports must agree at every PDAS iteration; coefficients within 1e-6 relative -- the Newton iterates are
compared after the same number of steps (the trace would differ in length otherwise)."""
import numpy as np
import pytest

from bess_amd import synth
from test_glm_gpu import check

pytestmark = pytest.mark.gpu
COX = dict(data_type=3, model_type=4)


@pytest.mark.parametrize("name,kw", [
    ("seq", dict(ic_type=3, sequence=np.arange(1, 13))),
    ("gs", dict(ic_type=4, path_type=2, s_min=1, s_max=20)),
    ("nowarm", dict(ic_type=2, sequence=np.arange(1, 9), is_warm_start=False)),
])
def test_cox_paths(gpu, name, kw):
    X, _, status, _, _ = synth.make_cox(600, 100, 6)
    check(gpu, X, status, dict(COX, **kw), "cox " + name)


def test_cox_cv_and_weights(gpu):
    X, _, status, _, _ = synth.make_cox(600, 100, 6)
    check(gpu, X, status, dict(COX, is_cv=True, K=5, cv_fold_id=synth.make_cv_folds(600, 5), sequence=np.arange(1, 9)),
          "cox cv")
    w = np.random.default_rng(1).uniform(0.5, 2, 600)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 9), weight=w), "cox weighted")


@pytest.mark.parametrize("n,p", [(2500, 300), (5000, 130)])
def test_cox_bigger(gpu, n, p):
    X, _, status, _, _ = synth.make_cox(n, p, 8, seed=n)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=np.arange(1, 17)), "cox %dx%d" % (n, p))


def test_cox_sparsity_levels_above_254(gpu):
    """Beyond the register-resident solver: the n x k work space grows on demand and the Newton system is solved by
    the blocked Cholesky (the reference's default s.list reaches min(p, n / log n))."""
    X, _, status, _, _ = synth.make_cox(1200, 400, 10, seed=77)
    check(gpu, X, status, dict(COX, ic_type=3, sequence=[250, 256, 300]), "cox k>254")


def test_cox_score_pass_forms_agree(gpu, monkeypatch):
    """BESSX_COX_SCORE=2pass (block totals, carry, rescan: X read twice) and the default one-pass form (summation
    order exchanged, X read once) give the same path: supports at every iteration, coefficients, IC values."""
    X, _, status, _, _ = synth.make_cox(3000, 260, 8, seed=12)
    fold = synth.make_cv_folds(3000, 5)
    outs = []
    for form in ("1pass", "2pass"):
        monkeypatch.setenv("BESSX_COX_SCORE", form)
        with gpu.Session(X, status, data_type=3, model_type=4) as s:
            s.set_cv(5, fold)
            s.trace_enable(True)
            outs.append((s.sequential_path(np.arange(1, 21), ic_type=3),
                         s.sequential_path(np.arange(1, 7), ic_type=3, is_cv=True)))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
        np.testing.assert_array_equal(a["cand_iters"], b["cand_iters"])
        assert a["n_pdas_iters"] == b["n_pdas_iters"]
        np.testing.assert_allclose(a["cand_beta"], b["cand_beta"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-10)
