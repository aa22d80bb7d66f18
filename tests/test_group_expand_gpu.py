"""Group selection with groups of one width: the selected groups are expanded to columns on the device
(k_group_expand; find_ind, src/utilities.cpp:113-130) and the PDAS iterations of a fit are queued as gated slots, one
host round trip per batch.  Same fits as the host-side expansion (test hook group_expand=host: two synchronisations per
iteration, round 3's form) and as the oracle: every candidate's groups, iteration counts, coefficients, criteria."""
import numpy as np
import pytest

from bess_amd import synth
from oracle import port_ctypes as P

from helpers import hooks  # noqa: E402

pytestmark = pytest.mark.gpu


def _grouped_lm(n, G, gs, k_true, seed):
    rng = np.random.default_rng(seed)
    p = G * gs
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    for g in rng.choice(G, k_true, replace=False):
        beta[g * gs:(g + 1) * gs] = rng.uniform(0.3, 1.5, gs) * rng.choice([-1.0, 1.0], gs)
    y = X @ beta + rng.standard_normal(n)
    return X, y, np.arange(0, p, gs).astype(np.int32)


@pytest.mark.parametrize("gs,G,T", [(5, 120, 14), (2, 200, 30), (20, 40, 6), (5, 150, 60), (3, 40, 40)])
def test_uniform_groups_expand_on_the_device(gpu, monkeypatch, gs, G, T):
    X, y, g_index = _grouped_lm(1500, G, gs, 6, seed=gs + G)
    outs = {}
    for mode in ("device", "host"):
        hooks(monkeypatch, group_expand=mode)
        with gpu.Session(X, y, g_index=g_index, algorithm_type=2) as s:
            outs[mode] = (s.sequential_path(np.arange(1, T + 1), ic_type=3), s.gs_path(1, min(T, 20), ic_type=3))
    for a, b in zip(outs["device"], outs["host"]):
        np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
        np.testing.assert_array_equal(a["cand_iters"], b["cand_iters"])
        np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-12)
        np.testing.assert_allclose(a["cand_beta"], b["cand_beta"], rtol=1e-10, atol=1e-14)
        assert a["best_T0"] == b["best_T0"] and a["n_pdas_iters"] == b["n_pdas_iters"]
    if G * gs <= 600 and T <= 30:
        want = P.trace(X, y, ic_type=3, sequence=np.arange(1, T + 1), g_index=g_index, algorithm_type=2)
        got = outs["device"][0]
        sup = np.nonzero(want["beta"])[0]
        assert np.array_equal(np.nonzero(got["beta"])[0], sup)
        np.testing.assert_allclose(got["beta"][sup], want["beta"][sup], rtol=1e-6)
        np.testing.assert_allclose(got["cand_ic"], want["ic_calls"], rtol=1e-8)
        assert list(got["cand_iters"]) == [len(f["iters"]) for f in want["fits"]]


def test_grouped_cv_and_ridge_paths_on_the_device_path(gpu, monkeypatch):
    X, y, g_index = _grouped_lm(1200, 80, 4, 5, seed=11)
    fold = synth.make_cv_folds(1200, 4, seed=2)
    outs = {}
    for mode in ("device", "host"):
        hooks(monkeypatch, group_expand=mode)
        with gpu.Session(X, y, g_index=g_index, algorithm_type=3) as s:
            s.set_cv(4, fold)
            outs[mode] = (s.sequential_path(np.arange(1, 11), [0.0, 0.05], ic_type=3, is_cv=True),
                          s.gs_path(1, 12, ic_type=3, is_cv=True))
    for a, b in zip(outs["device"], outs["host"]):
        np.testing.assert_array_equal(a["cand_support"], b["cand_support"])
        np.testing.assert_allclose(a["cand_ic"], b["cand_ic"], rtol=1e-12)
        assert a["n_fits"] == b["n_fits"] and a["n_pdas_iters"] == b["n_pdas_iters"]


@pytest.mark.parametrize("gs,ragged", [(4, False), (3, True)])
def test_group_blocks_diagonalised_once_per_row_set_and_lambda(gpu, monkeypatch, gs, ragged):
    """The eigenvectors of every group's block are kept per (row set, lambda) and reused by every iteration, every
    candidate and every fold fit; a new lambda or new folds form them again.  Same fits as forming them every time."""
    X, y, g_index = _grouped_lm(900, 60, gs, 5, seed=31 + gs)
    if ragged:  # (merge two groups: ragged widths take the host-side expansion, the same cached blocks)
        g_index = np.delete(g_index, 7)
    outs = {}
    for mode in ("1", "0"):
        hooks(monkeypatch, group_eig=mode)
        with gpu.Session(X, y, g_index=g_index, algorithm_type=3) as s:
            o = [s.sequential_path(np.arange(1, 9), [0.0, 0.3, 0.0], ic_type=3)]
            for seed in (2, 5):
                s.set_cv(3, synth.make_cv_folds(900, 3, seed=seed))
                o.append(s.sequential_path(np.arange(1, 7), [0.1, 0.0], ic_type=3, is_cv=True))
            o.append(s.fit(4, lam=0.3))
            o.append(s.fit(4, lam=0.0))
            outs[mode] = o
    for a, b in zip(outs["1"], outs["0"]):
        for key in ("cand_support", "cand_iters", "support", "iters"):
            if key in a:
                np.testing.assert_array_equal(a[key], b[key])
        for key in ("cand_ic", "cand_beta", "beta"):
            if key in a:
                np.testing.assert_allclose(a[key], b[key], rtol=1e-12, atol=1e-14)
    want = P.trace(X, y, ic_type=3, sequence=np.arange(1, 9), lambda_seq=[0.0, 0.3, 0.0], g_index=g_index,
                   algorithm_type=3)
    np.testing.assert_allclose(outs["1"][0]["cand_ic"], want["ic_calls"], rtol=1e-8)


@pytest.mark.parametrize("gs,G,T,lam", [(5, 120, 20, 0.0), (2, 300, 40, 0.2), (8, 60, 12, 0.0), (3, 40, 40, 0.0)])
def test_grouped_lm_in_the_covariance_form(gpu, gs, G, T, lam):
    """Groups of one width, all rows: d = X^T y - G_A beta_A from cached Gram columns (no pass over X per PDAS
    iteration), the selected groups' columns looked up in the cache, fills of whole groups listed by the host when a fit
    parks.  Same candidates as the streaming form (score_mode = 1) and the oracle; a handful of passes over X per path."""
    X, y, g_index = _grouped_lm(1500, G, gs, 6, seed=3 * gs + G)
    seq, lams = np.arange(1, T + 1), [lam]
    outs, passes = {}, {}
    for mode in (2, 1):
        with gpu.Session(X, y, g_index=g_index, algorithm_type=3 if lam else 2, score_mode=mode) as s:
            assert s.score_mode() == mode
            outs[mode] = [s.sequential_path(seq, lams, ic_type=3)]
            passes[mode] = s.counters()["passes_over_X"]
            outs[mode] += [s.gs_path(1, min(T, 20), ic_type=3), s.fit(min(T, 7), lam=lam),
                           s.fit(min(T, 7), lam=lam, init_idx=np.arange(2 * gs), init_val=np.ones(2 * gs))]
            s.set_cv(3, synth.make_cv_folds(1500, 3, seed=1))  # fold fits stay in the streaming form
            outs[mode].append(s.sequential_path(np.arange(1, 6), lams, ic_type=3, is_cv=True))
    for a, b in zip(outs[2], outs[1]):
        for key in ("cand_support", "cand_iters", "support", "iters"):
            if key in a:
                np.testing.assert_array_equal(a[key], b[key])
        for key in ("cand_ic", "cand_beta", "beta", "train_loss"):
            if key in a:
                np.testing.assert_allclose(a[key], b[key], rtol=1e-9, atol=1e-12)
    # every column is formed at most once per path; only when fewer than 32 uncached columns are left do the last groups
    # come in a few columns per pass (the p = 120 case here)
    assert passes[2] <= gs * G // 32 + 10
    if G * gs <= 600 and T <= 30 and lam == 0.0:
        want = P.trace(X, y, ic_type=3, sequence=seq, g_index=g_index, algorithm_type=2)
        np.testing.assert_allclose(outs[2][0]["cand_ic"], want["ic_calls"], rtol=1e-8)
        assert list(outs[2][0]["cand_iters"]) == [len(f["iters"]) for f in want["fits"]]


def test_covariance_form_insisted_on_where_it_does_not_exist(gpu):
    X, y, g_index = _grouped_lm(600, 30, 4, 3, seed=5)
    ragged = np.delete(g_index, 3)
    with pytest.raises(Exception):
        gpu.Session(X, y, g_index=ragged, algorithm_type=2, score_mode=2)
    with gpu.Session(X, y, g_index=ragged, algorithm_type=2) as s:
        assert s.score_mode() == 1
