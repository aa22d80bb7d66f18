"""CPU, world_size 2 over gloo: the multi-GPU host logic (unit partition + all-gather of the IC curve + best
model selection).  The per-rank solver is stood in for by the plain-C oracle (test infrastructure); on the GPU
box the same code path runs with backend nccl and the HIP library (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, seq, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bess_amd import dist as bdist, synth
    from oracle import port_ctypes as P
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    X, y, _, _ = synth.make_lm(400, 80, 5)
    lo, hi = bdist.partition(len(seq), world, rank)
    # each rank solves its contiguous chunk of the k-path as one warm-start chain
    t = P.trace(X, y, ic_type=3, sequence=seq[lo:hi])
    curve = bdist.gather_curve(t["ic_calls"], len(seq), world, rank)
    if rank == 0:
        np.save(out_path, curve)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_all_units():
    from bess_amd import dist as bdist
    for n in (1, 7, 200):
        for w in (1, 2, 3, 8):
            spans = [bdist.partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.timeout(300)
def test_kpath_sharded_over_two_ranks(tmp_path):
    sys.path.insert(0, ROOT)
    from bess_amd import dist as bdist, synth
    from oracle import port_ctypes as P
    seq = np.arange(1, 12)
    out = str(tmp_path / "curve.npy")
    mp.spawn(_worker, args=(2, 29517, seq, out), nprocs=2, join=True)
    curve = np.load(out)
    X, y, _, _ = synth.make_lm(400, 80, 5)
    # expected: the same two chains run one after the other in a single process
    lo, hi = bdist.partition(len(seq), 2, 0)
    want = np.concatenate([P.trace(X, y, ic_type=3, sequence=seq[lo:hi])["ic_calls"],
                           P.trace(X, y, ic_type=3, sequence=seq[hi:])["ic_calls"]])
    np.testing.assert_allclose(curve, want, rtol=0, atol=0)
    assert bdist.select_best(curve) == int(np.argmin(want))


# ---- cross-validated paths with the fold fits dealt to ranks (bess_amd.dist.FoldShardedCV) ----------------
def _cv_problem():
    from bess_amd import synth
    X, y, _, _ = synth.make_lm(300, 60, 5)
    return X, y, synth.make_cv_folds(300, 5)


def _cv_worker(rank, world, port, path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bess_amd import dist as bdist
    from helpers import NumpyLmSession
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    X, y, fold = _cv_problem()
    cv = bdist.FoldShardedCV(NumpyLmSession(X, y, fold, 5), 5, world, rank, is_warm_start=not path.endswith("cold"))
    if path in ("gs", "gscold"):
        out = cv.gs_path(1, 20)
    elif path == "seqlam":  # lambda grid in snake order (src/path.cpp:50)
        out = cv.sequential_path(np.arange(1, 7), [0.0, 0.02, 0.1])
    else:
        out = cv.sequential_path(np.arange(1, 13))
    np.savez(out_path + ".%d.npz" % rank, beta=out["beta"], scal=[out["coef0"], out["train_loss"], out["ic"]],
             cand_ic=out["cand_ic"], cand_T0=out["cand_T0"], n_fits=out["n_fits"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("path,world", [("gs", 2), ("seq", 2), ("gs", 3), ("seqlam", 2), ("seqcold", 3),
                                        ("gscold", 8), ("seqcold", 8)])
def test_cv_folds_sharded_over_ranks(tmp_path, path, world):
    """K fold chains + the full-data chain on `world` gloo ranks: every rank ends with the same model, and it is the
    one the pinned oracle's single-process gs_path / sequential_path under CV selects (same folds)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import port_ctypes as P
    out = str(tmp_path / "cv")
    port = 29531 + world + {"gs": 7, "seq": 0, "seqlam": 13, "seqcold": 19, "gscold": 29}[path]
    mp.spawn(_cv_worker, args=(world, port, path, out), nprocs=world, join=True)
    X, y, fold = _cv_problem()
    kw = {"gs": dict(path_type=2, s_min=1, s_max=20), "seq": dict(sequence=np.arange(1, 13)),
          "seqlam": dict(sequence=np.arange(1, 7), lambda_seq=[0.0, 0.02, 0.1]),
          "seqcold": dict(sequence=np.arange(1, 13), is_warm_start=False),
          # (fold x s) pairs of the golden-section evaluations and of the final sweep over 8 ranks
          "gscold": dict(path_type=2, s_min=1, s_max=20, is_warm_start=False)}[path]
    want = P.trace(X, y, is_cv=True, K=5, cv_fold_id=fold, **kw)
    got = [np.load(out + ".%d.npz" % r) for r in range(world)]
    for g in got[1:]:
        for k in ("beta", "scal", "cand_ic", "cand_T0"):
            np.testing.assert_array_equal(g[k], got[0][k])
    sup = np.nonzero(want["beta"])[0]
    assert np.array_equal(np.nonzero(got[0]["beta"])[0], sup)
    np.testing.assert_allclose(got[0]["beta"][sup], want["beta"][sup], rtol=1e-8)
    np.testing.assert_allclose(got[0]["scal"], [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-9)
    # one CV value per candidate stored, in evaluation order, equal to the oracle's first ic() of that point
    assert int(got[0]["n_fits"]) == len(want["fits"])


def test_cv_sharding_is_independent_of_world_size():
    """world = 1 (no process group) gives the same path as the oracle too: the unit -> rank map changes nothing."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bess_amd import dist as bdist
    from helpers import NumpyLmSession
    from oracle import port_ctypes as P
    X, y, fold = _cv_problem()
    out = bdist.FoldShardedCV(NumpyLmSession(X, y, fold, 5), 5).gs_path(1, 20)
    want = P.trace(X, y, is_cv=True, K=5, cv_fold_id=fold, path_type=2, s_min=1, s_max=20)
    assert np.array_equal(np.nonzero(out["beta"])[0], np.nonzero(want["beta"])[0])
    np.testing.assert_allclose([out["coef0"], out["train_loss"], out["ic"]],
                               [want["coef0"], want["train_loss"], want["ic"]], rtol=1e-9)


# ---- k-path chunks stitched into the single warm-start chain (bess_amd.dist.StitchedKPath) -------------------
def _hard_lm(n=300, p=80, seed=3):
    """Correlated columns and a weak signal: a chunk started cold lands in other local fixed points than the warm
    chain, so the stitching has real work (half of the candidates of the cold chunks differ from the single chain)."""
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, p))
    X = Z.copy()
    for j in range(1, p):
        X[:, j] = 0.8 * X[:, j - 1] + 0.6 * Z[:, j]
    beta = np.zeros(p)
    beta[rng.choice(p, 12, replace=False)] = rng.uniform(0.3, 1.0, 12) * rng.choice([-1.0, 1.0], 12)
    return X, X @ beta + rng.standard_normal(n)


def _stitch_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bess_amd import dist as bdist
    from helpers import NumpyLmSession
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    X, y = _hard_lm()
    seq = np.arange(1, 25)
    sk = bdist.StitchedKPath(NumpyLmSession(X, y, np.zeros(len(y), int), 1), seq, world, rank, rebalance=True)
    # (the lead fits of round 6 -- every rank walks the one-GPU path's coarse levels below its chunk first -- change where
    # a chunk starts, never the stitched path: same supports and curve below)
    lead = bdist.StitchedKPath(NumpyLmSession(X, y, np.zeros(len(y), int), 1), seq, world, rank, coarse_lead=True)
    rep_lead = lead.step()
    rep = sk.step()
    assert np.allclose(rep_lead["ic_curve"], rep["ic_curve"], rtol=1e-12) and rep_lead["best_k"] == rep["best_k"]
    assert np.array_equal(rep_lead["chunk"]["cand_support"], rep["chunk"]["cand_support"])
    assert rank == 0 or (lead.lead_levels().size >= 1 and int(lead.lead_levels()[-1]) == int(lead.seq[0]) - 1)
    sup = np.full((len(seq), 24), -1)
    lo, hi = bdist.partition(len(seq), world, rank)
    assert rep["bounds"][rank] == lo and rep["bounds"][rank + 1] == hi  # the first step runs the equal split
    sup[lo:hi, :rep["chunk"]["cand_support"].shape[1]] = rep["chunk"]["cand_support"]
    # a second step on boundaries moved by hand to a very uneven split (what rebalancing may arrive at): same path
    sk.bounds = [0] + [len(seq) - world + r for r in range(1, world)] + [len(seq)]
    sk.lo, sk.hi = sk.bounds[rank], sk.bounds[rank + 1]
    sk.seq = sk.full_seq[sk.lo:sk.hi]
    rep2 = sk.step()
    sup2 = np.full((len(seq), 24), -1)
    sup2[rep2["bounds"][rank]:rep2["bounds"][rank + 1], :rep2["chunk"]["cand_support"].shape[1]] = rep2["chunk"]["cand_support"]
    np.savez(out_path + ".%d.npz" % rank, sup=sup, lo=lo, hi=hi, curve=rep["ic_curve"], best=rep["best_k"],
             refits=rep["stitch_refits_per_rank"], rounds=rep["stitch_rounds"], sup2=sup2, curve2=rep2["ic_curve"],
             bounds2=np.asarray(rep2["bounds"]), bounds3=np.asarray(sk.bounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_stitched_kpath_equals_the_single_chain(tmp_path, world):
    """Chunks of the k-path on `world` gloo ranks, stitched: every candidate's support equals the single warm-start
    chain's (the pinned oracle's sequential_path), although the cold chunks alone differ from it."""
    sys.path.insert(0, ROOT)
    from bess_amd import dist as bdist
    from oracle import port_ctypes as P
    out = str(tmp_path / "st")
    mp.spawn(_stitch_worker, args=(world, 29600 + world, out), nprocs=world, join=True)
    X, y = _hard_lm()
    seq = np.arange(1, 25)
    want = P.trace(X, y, ic_type=3, sequence=seq)
    got = [np.load(out + ".%d.npz" % r) for r in range(world)]
    for k, f in enumerate(want["fits"]):
        r = [i for i in range(world) if got[i]["lo"] <= k < got[i]["hi"]][0]
        assert np.array_equal(got[r]["sup"][k, :k + 1], f["iters"][-1]), k + 1
    for g in got:
        np.testing.assert_allclose(g["curve"], want["ic_calls"], rtol=1e-9)
        assert int(g["best"]) == int(seq[int(np.argmin(want["ic_calls"]))])
        np.testing.assert_allclose(g["curve2"], want["ic_calls"], rtol=1e-9)
        assert np.array_equal(g["bounds2"], got[0]["bounds2"]) and np.array_equal(g["bounds3"], got[0]["bounds3"])
    b2 = got[0]["bounds2"]
    for k, f in enumerate(want["fits"]):  # the uneven split of the second step: the same single chain
        r = [i for i in range(world) if b2[i] <= k < b2[i + 1]][0]
        assert np.array_equal(got[r]["sup2"][k, :k + 1], f["iters"][-1]), k + 1
    # the cold chunks alone would NOT have been the single chain (the stitch replaced candidates)
    assert int(np.sum(got[0]["refits"])) > world - 1
    cold = P.trace(X, y, ic_type=3, sequence=seq[bdist.partition(len(seq), world, world - 1)[0]:])
    assert any(not np.array_equal(c["iters"][-1], w["iters"][-1])
               for c, w in zip(cold["fits"], want["fits"][bdist.partition(len(seq), world, world - 1)[0]:]))


def test_rebalance_bounds():
    """Chunk boundaries move towards equal time per rank, half way per step; outliers are clipped, small imbalances and
    degenerate inputs leave the boundaries alone, every chunk keeps a candidate."""
    sys.path.insert(0, ROOT)
    from bess_amd.dist import rebalance_bounds
    assert rebalance_bounds([0, 100, 200], [4.5, 9.3], damping=1.0) == [0, 126, 200]
    assert rebalance_bounds([0, 100, 200], [4.5, 9.3]) == [0, 113, 200]
    assert rebalance_bounds([0, 100, 200], [4.5, 4.8]) == [0, 100, 200]                    # inside the dead band
    assert rebalance_bounds([0, 100, 200], [4.5, 9.3], fixed_seconds=100.0) == [0, 100, 200]
    b = rebalance_bounds([0, 50, 100, 150, 200], [2.8, 460.0, 3.6, 4.1])                    # one rank hit a 0.4 s stall
    assert b[0] == 0 and b[-1] == 200 and all(y > x for x, y in zip(b, b[1:])) and abs(b[2] - 100) <= 15
    assert rebalance_bounds([0, 1, 2, 3], [1.0, 50.0, 1.0], damping=1.0) == [0, 1, 2, 3]   # nothing left to give
    assert rebalance_bounds([0, 7], [3.0]) == [0, 7]
    assert rebalance_bounds([0, 3, 6], [0.0, 0.0]) == [0, 3, 6]
    b = [0, 50, 100]
    for _ in range(12):  # candidates of the upper half cost three times as much: converges to equal time and stays
        cost = [1.0 * (b[1] - b[0]), 3.0 * (b[2] - b[1])]
        b = rebalance_bounds(b, cost)
    assert b == [0, 72, 100] or b == [0, 73, 100] or b == [0, 74, 100]
