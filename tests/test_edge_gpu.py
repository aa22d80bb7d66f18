"""Edge cases of the hot path on the GPU against the plain-C oracle: tiny and ragged shapes, every column selected,
one PDAS iteration only, unsorted / repeated sparsity sequences, two folds, more than 32768 columns (two-level
top-k), always_select filling the whole active set, zero weights."""
import numpy as np
import pytest

from bess_amd import synth
from test_lm_gpu import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p", [(5, 3), (17, 2), (129, 5), (1025, 9)])
def test_tiny_and_ragged_shapes(gpu, n, p):
    rng = np.random.default_rng(n * 10 + p)
    X = rng.standard_normal((n, p))
    y = X[:, 0] * 2 - X[:, p - 1] + 0.1 * rng.standard_normal(n)
    check(gpu, X, y, dict(ic_type=1, sequence=np.arange(1, p + 1)), "tiny %dx%d" % (n, p))  # includes T0 == p


def test_single_iteration_and_odd_sequences(gpu):
    X, y, _, _ = synth.make_lm(700, 120, 6, seed=21)
    check(gpu, X, y, dict(ic_type=3, sequence=[5, 3, 3, 8, 1, 20], max_iter=1), "max_iter=1, unsorted sequence")
    check(gpu, X, y, dict(ic_type=2, sequence=[7, 7, 7]), "repeated size")
    check(gpu, X, y, dict(ic_type=3, path_type=2, s_min=4, s_max=4), "gs on a single size")
    check(gpu, X, y, dict(ic_type=9, sequence=[2, 3]), "unknown ic_type -> ic = 0 (src/Metric.h:227-228)")


def test_two_folds_and_zero_weights(gpu):
    X, y, _, _ = synth.make_lm(600, 80, 5, seed=22)
    check(gpu, X, y, dict(is_cv=True, K=2, cv_fold_id=synth.make_cv_folds(600, 2), sequence=np.arange(1, 9)), "K=2")
    w = np.ones(600)
    w[::7] = 0.0
    check(gpu, X, y, dict(ic_type=3, sequence=np.arange(1, 9), weight=w), "zero weights")


def test_always_select_fills_the_active_set(gpu):
    X, y, _, _ = synth.make_lm(500, 60, 5, seed=23)
    check(gpu, X, y, dict(ic_type=3, sequence=[3, 4], always_select=[10, 20, 30]), "always_select == T0")


def test_two_level_topk_many_columns(gpu):
    X, y, _, _ = synth.make_lm(150, 40000, 4, seed=24)
    check(gpu, X, y, dict(ic_type=4, sequence=np.arange(1, 7)), "p = 40000")


def test_logistic_and_cox_small(gpu):
    from test_glm_gpu import check as gcheck
    X, y, _, _ = synth.make_logistic(60, 7, 2, seed=25)
    gcheck(gpu, X, y, dict(data_type=2, model_type=2, ic_type=1, sequence=np.arange(1, 6)), "logistic 60x7")
    Xc, _, st, _, _ = synth.make_cox(80, 6, 2, seed=26)
    gcheck(gpu, Xc, st, dict(data_type=3, model_type=4, ic_type=1, sequence=np.arange(1, 5)), "cox 80x6")


@pytest.mark.parametrize("fam", ["lm", "lm-streaming", "lm-cv", "logistic", "cox"])
def test_poisoned_device_allocations_change_nothing(gpu, monkeypatch, fam):
    """hipMalloc returns whatever the memory last held -- zeros on a fresh box, earlier sessions' bytes in a long-lived
    process.  BESSX_TEST_HOOKS=poison=1 fills every device allocation of the library with 0xFF bytes (NaN / -1): a
    kernel that reads a buffer nobody wrote would show here.  (The whole -m gpu suite was run once under the hook in
    round 5: 375 passed.)"""
    from helpers import hooks
    from bess_amd import synth

    def run():
        if fam in ("lm", "lm-streaming", "lm-cv"):
            X, y, _, _ = synth.make_lm(1500, 400, 8, seed=4)
            with gpu.Session(X, y, score_mode=1 if fam == "lm-streaming" else 2) as s:
                if fam == "lm-cv":
                    s.set_cv(4, synth.make_cv_folds(1500, 4))
                    return s.gs_path(1, 24, ic_type=3, is_cv=True)
                return s.sequential_path(np.arange(1, 31), ic_type=3)
        if fam == "logistic":
            X, y, _, _ = synth.make_logistic(1500, 300, 8, seed=4)
            with gpu.Session(X, y, data_type=2, model_type=2) as s:
                return s.sequential_path(np.arange(1, 21), ic_type=3)
        X, _, st, _, _ = synth.make_cox(1500, 300, 8)
        with gpu.Session(X, st, data_type=3, model_type=4) as s:
            return s.sequential_path(np.arange(1, 16), ic_type=3)

    want = run()
    hooks(monkeypatch, poison=1)
    got = run()
    for k in ("cand_T0", "cand_support", "cand_iters"):
        assert np.array_equal(got[k], want[k]), k
    for k in ("cand_beta", "cand_ic", "cand_train_loss"):
        assert np.array_equal(got[k], want[k]), k  # bitwise: the same arithmetic on the same data
