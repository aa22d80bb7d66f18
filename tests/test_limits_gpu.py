"""GPU: shape limits the round-1 review found.

* top-k for p just above a multiple of the 32768-wide selection chunk (the chunks are balanced now; whether a fit was
  accepted used to depend on p mod 32768);
* sparsity levels beyond 2046 (bessx_problem.max_sparsity sizes the k x k work space; bessx_pywrap_bess derives it
  from the path, so the reference's default sequence 1..min(p, n / log n) runs as it is);
* screening scores of an all-zero column (must rank where the compiled reference ranks it, never by a NaN key).
"""
import numpy as np
import pytest

from bess_amd import synth
from bess_amd import linear
from oracle import port_ctypes as P
from oracle import ref_ctypes as R
from helpers import assert_same_trace
from test_lm_gpu import run_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("length", [32769, 33000, 65537, 98305, 131073])
@pytest.mark.parametrize("k", [1, 2, 232, 233, 1000, 2046])
def test_topk_lengths_just_above_a_chunk_multiple(gpu, length, k):
    rng = np.random.default_rng(length * 7 + k)
    s = rng.standard_normal(length) ** 2
    s[rng.choice(length, 50, replace=False)] = s[0]  # ties across chunk borders
    assert np.array_equal(gpu.op_topk(s, k), P.max_k(s, k))


@pytest.mark.parametrize("p", [32769, 33000, 65537])
def test_lm_path_with_p_just_above_a_chunk_multiple(gpu, p):
    X, y, _, _ = synth.make_lm(200, p, 4, seed=p)
    kw = dict(ic_type=3, sequence=[1, 2, 3, 8, 40, 150] if p == 33000 else [1, 2, 3, 8])
    want = P.trace(X, y, **kw)
    got = run_gpu(gpu, X, y, kw)
    assert_same_trace(got["trace"], want, what="p=%d" % p)
    np.testing.assert_allclose(got["beta"], want["beta"], rtol=1e-6, atol=1e-12)


def test_sparsity_levels_beyond_2046(gpu):
    """A session created with max_sparsity holds the work space for larger active sets (blocked Cholesky in global
    memory); without it the limit is named in the error.  Expected values: tests/golden/ref_bigk.npz, every PDAS
    iteration of k = 2040, 2100, 2300 on an 8000 x 2600 problem from the COMPILED REFERENCE
    (tests/golden/make_fullsize_ref.py bigk; the plain-C oracle needs minutes per iteration at these sizes)."""
    import os
    from test_fullsize_families_gpu import assert_matches_golden, assert_best_model
    path = os.path.join(os.path.dirname(__file__), "golden", "ref_bigk.npz")
    assert os.path.exists(path), "golden file %s is missing (a committed fixture, not optional)" % path
    g = np.load(path)
    X, y, _, _ = synth.make_lm(int(g["n"]), int(g["p"]), 10, seed=int(g["seed"]))
    seq = [int(v) for v in g["sequence"]]
    for mode in (1, 2):
        with gpu.Session(X, y, max_sparsity=max(seq), score_mode=mode) as s:
            s.trace_enable(True)
            got = s.sequential_path(seq, ic_type=3)
        assert_matches_golden(got["trace"], g, "k > 2046, score_mode %d" % mode)
        assert_best_model(got, g)
    with gpu.Session(X, y) as s:
        with pytest.raises(gpu.BessxError) as e:
            s.sequential_path([2100], ic_type=3)
        assert e.value.code == 1 and "max_sparsity" in str(e.value)
    # the pywrap_bess entry sizes the session from the path itself
    n, p = X.shape
    r = gpu.pywrap_bess(X, y, 1, np.ones(n), True, 1, 1, 20, 0, 1, True, 3, False, 5, np.arange(p), np.ones(n),
                        seq, [0.0], 1, 1, 0, 1e-4, 0.0, 0.0, 100, False, 1, 1, [], 0.0, p)
    assert np.array_equal(np.nonzero(r[0])[0], g["best_beta_idx"])
    np.testing.assert_allclose(r[0][g["best_beta_idx"]], g["best_beta_val"], rtol=1e-6)


@pytest.mark.timeout(600)
def test_estimator_with_default_arguments_above_the_old_cap(gpu):
    """PdasLm().fit(X, y) with DEFAULT arguments: sequence = 1..min(p, n / log n) (python/bess/linear.py:285-287),
    here 1..2468 > 2046 -- used to fail with BESSX_ERR_ARG.  2468 candidates are far beyond what the
    CPU checkers finish, so this is a property test: EBIC selects exactly the planted variables and the coefficients
    are the least-squares fit on them."""
    n, p = 25000, 3000
    assert min(p, int(n / np.log(n))) == 2468
    X, y, support, _ = synth.make_lm(n, p, 12, seed=9)
    m = linear.PdasLm()
    m.fit(X, y)
    assert m.sequence == list(range(1, 2469))
    assert np.array_equal(np.nonzero(m.beta)[0], support)
    Xs = np.column_stack([np.ones(n), X[:, support]])
    ls = np.linalg.lstsq(Xs, y, rcond=None)[0]
    np.testing.assert_allclose(m.beta[support], ls[1:], rtol=1e-8)
    np.testing.assert_allclose(m.coef0, ls[0], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError, match="16382"):
        linear.PdasLm(sequence=[17000]).fit(np.zeros((4, 17001)), np.zeros(4))


@pytest.mark.parametrize("family", ["lm", "logistic", "cox"])
def test_screening_ranks_an_all_zero_column_last(gpu, family):
    rng = np.random.default_rng(3)
    if family == "lm":
        X, y, _, _ = synth.make_lm(300, 40, 4, seed=2)
        kw, mt = dict(), 1
    elif family == "logistic":
        X, y, _, _ = synth.make_logistic(400, 40, 4, seed=2)
        kw, mt = dict(data_type=2, model_type=2), 2
    else:
        X, _, y, _, _ = synth.make_cox(400, 40, 4, seed=2)
        kw, mt = dict(data_type=3, model_type=4), 4
    X = np.array(X)
    X[:, [5, 17]] = 0.0
    # the compiled reference's `screening_A` on exactly these inputs (recorded with oracle/_ref/libbess_ref.so in the
    # build container; re-checked against it wherever it is present): LM drops {12, 14}, the other two {5, 17}
    dropped = {"lm": (12, 14), "logistic": (5, 17), "cox": (5, 17)}[family]
    keep = np.array([j for j in range(40) if j not in dropped])
    if R.available():
        assert np.array_equal(R.screening(X, y, None, mt, 38), keep)
    # LM: Eigen's colPivHouseholderQr divides by the zero pivot (beta = +-inf): the reference KEEPS such columns;
    # logistic / Cox: the LDLT solve zeroes them, they rank last
    assert (5 in keep and 17 in keep) if family == "lm" else (5 not in keep and 17 not in keep)
    with gpu.Session(X, y, is_screening=True, screening_size=38, **kw) as s:
        assert np.array_equal(s.screening(), keep)
