"""BASELINE configs[1] at FULL size (n=50000, p=10000) on the GPU.

1. Size-independent properties of the path (no oracle needed): run-to-run bitwise determinism, every fit stopped
   because its active set repeated (nearly always a fixed point of the PDAS map), the training loss
   decreases along the nested part of the path, the IC matches its formula, the true support is recovered at
   k = k_true, and the golden-section path selects the same model as the exhaustive sequential path.
2. tests/golden/fullsize_lm.npz (generated in the build container from the COMPILED REFERENCE by
   tests/golden/make_fullsize_ref.py: 10038 s of one CPU core for the 200 candidates): the active set of every
   PDAS iteration (421) of all 200 candidates, the coefficients, losses and ICs are compared with it, in both
   score-pass forms.
"""
import os

import numpy as np
import pytest

from bess_amd import synth

from helpers import assert_untraced_path_matches_golden  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fullsize_lm.npz")


@pytest.fixture(scope="module", params=[2, 1], ids=["covariance", "streaming"])
def full(gpu, request):
    """Both evaluations of the score pass (include/bessx.h, score_mode) at full size."""
    X, y, support, beta = synth.make_lm()
    s = gpu.Session(X, y, score_mode=request.param)
    s.trace_enable(True)
    out = s.sequential_path(np.arange(1, 201), ic_type=3)
    yield gpu, s, out, support, (X, y)
    s.close()


def test_properties_at_full_size(full):
    gpu, s, out, support, _ = full
    n, p = 50000, 10000
    again = s.sequential_path(np.arange(1, 201), ic_type=3)
    assert np.array_equal(out["cand_support"], again["cand_support"])
    assert np.array_equal(out["cand_beta"], again["cand_beta"]) and np.array_equal(out["cand_ic"], again["cand_ic"])
    fits = out["trace"]["fits"]
    assert len(fits) == 200
    for f in fits:  # Algorithm::fit stopped because the last active set had been seen before in this fit
        assert 2 <= len(f["iters"]) <= 20
        assert any(np.array_equal(f["iters"][-1], prev) for prev in f["iters"][:-1])
    assert sum(np.array_equal(f["iters"][-1], f["iters"][-2]) for f in fits) >= 150  # mostly a fixed point; the rest are 2-cycles
    loss = out["cand_train_loss"]
    assert np.all(np.diff(loss[:100]) < 0)  # every true variable lowers the loss
    c = np.log(p) * np.log(np.log(n))
    np.testing.assert_allclose(out["cand_ic"], n * np.log(loss) + c * np.arange(1, 201), rtol=1e-12)
    assert np.array_equal(np.sort(out["cand_support"][99][:100]), support)  # k = k_true recovers the truth
    assert out["best_T0"] == 100
    gs = s.gs_path(1, 200, ic_type=3)
    assert gs["best_T0"] == 100 and np.array_equal(np.nonzero(gs["beta"])[0], support)
    np.testing.assert_allclose(gs["beta"][support], out["beta"][support], rtol=1e-9)


def test_matches_compiled_reference_at_full_size(full):
    _, _, out, _, _ = full
    assert os.path.exists(GOLD), "golden file %s is missing (a committed fixture, not optional)" % GOLD
    g = np.load(GOLD)
    kmax = int(g["kmax"])
    fits = out["trace"]["fits"][:kmax]
    assert list(g["fit_iters"]) == [len(f["iters"]) for f in fits]
    got_A = np.concatenate([a for f in fits for a in f["iters"]])
    assert np.array_equal(got_A, g["A_flat"])  # bit-exact supports, every iteration of every candidate
    got_b = np.concatenate([b for f in fits for b in f["betas"]])
    np.testing.assert_allclose(got_b, g["beta_flat"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out["trace"]["ic_calls"][:kmax], g["ic_calls"], rtol=1e-9)
    np.testing.assert_allclose(out["trace"]["loss_calls"][:kmax], g["loss_calls"], rtol=1e-9)
    assert np.array_equal(np.nonzero(out["beta"])[0], g["best_beta_idx"])
    np.testing.assert_allclose(out["beta"][g["best_beta_idx"]], g["best_beta_val"], rtol=1e-6)


@pytest.fixture(scope="module")
def untraced(gpu):
    """The shipped configurations of configs[1], as bench.py runs them: no trace."""
    X, y, _, _ = synth.make_lm()
    outs = {}
    with gpu.Session(X, y, score_mode=2) as s:
        for chains in (0, 1, 4, 2):
            s.set_kpath_chains(chains)
            before = s.counters()["kpath_chunked_paths"]
            out = s.sequential_path(np.arange(1, 201), ic_type=3)
            cnt = s.counters()
            cnt["chunked"] = cnt["kpath_chunked_paths"] - before  # did THIS path run as chunk chains?
            outs["covariance, chains=%s" % ("auto" if chains == 0 else chains)] = (out, cnt)
    with gpu.Session(X, y, score_mode=1) as s:
        for chains in (0, 1, 4):  # automatic (round 6: 8 chunk chains that share their passes over X), one chain, four
            s.set_kpath_chains(chains)
            before = s.counters()["kpath_chunked_paths"]
            out = s.sequential_path(np.arange(1, 201), ic_type=3)
            cnt = s.counters()
            cnt["chunked"] = cnt["kpath_chunked_paths"] - before
            outs["streaming, chains=%s" % ("auto" if chains == 0 else chains)] = (out, cnt)
    return X, outs


@pytest.mark.parametrize("which", ["covariance, chains=auto", "covariance, chains=1", "covariance, chains=4",
                                   "covariance, chains=2", "streaming, chains=auto", "streaming, chains=1",
                                   "streaming, chains=4"])
def test_benchmarked_path_matches_compiled_reference_at_full_size(untraced, which):
    X, outs = untraced
    out, counters = outs[which]
    assert out["trace"] is None or len(out["trace"]["fits"]) == 0  # untraced: nothing was recorded
    g = np.load(GOLD)
    n = assert_untraced_path_matches_golden(out, g, X, 1, "configs[1] untraced, " + which)
    assert n == 200 and out["n_fits"] == 200 and out["n_pdas_iters"] == int(np.sum(g["fit_iters"]))
    if which in ("covariance, chains=1", "streaming, chains=1"):
        assert counters["chunked"] == 0
    if which in ("covariance, chains=auto", "streaming, chains=auto"):
        assert counters["chunked"] == 1 and counters["kpath_chains_last_path"] == (4 if which.startswith("cov") else 8)
    if which.startswith("streaming") and which[-1] != "1":  # the chains shared their passes
        assert counters["shared_pass_launches"] > 0  # (the chain slots are only counted with the kernel timing on)
    if which in ("covariance, chains=4", "covariance, chains=2", "streaming, chains=4"):
        assert counters["kpath_chains_last_path"] == int(which[-1]) and counters["chunked"] == 1
