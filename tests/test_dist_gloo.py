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
