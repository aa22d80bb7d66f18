"""Timing of the other BASELINE configs on one GPU (not the driver's bench.py, which is configs[1]):
  python tools/bench_family.py logistic [n p kmax]     configs[2]  default 100000 5000 100
  python tools/bench_family.py cox [n p kmax]          configs[4]  default 200000 20000 150
  python tools/bench_family.py lmcv [n p smax]         configs[3]  default 50000 10000 200  (gs_path + 5-fold CV)
  python tools/bench_family.py poisson [n p kmax]      SURVEY 8f   default 100000 5000 100
  python tools/bench_family.py grouped [n p kmax]      SURVEY 8f   default 50000 10000 40   (p / 5 groups of 5 columns)
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_family.py lmcv-sharded [n p smax]
      configs[3] with the 5 fold chains + the full-data chain dealt to N ranks (bess_amd.dist.FoldShardedCV);
      BESSX_BENCH_BACKEND=gloo BESSX_BENCH_ONE_DEVICE=1 rehearses the N-rank path on ONE GPU (<= 6 ranks).
Prints one JSON line: candidates/s, PDAS iterations, score-pass timing (HIP events inside the library)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bess_amd import capi, synth  # noqa: E402


def main():
    fam = sys.argv[1]
    a = [int(v) for v in sys.argv[2:]]
    t0 = time.time()
    if fam == "logistic":
        n, p, kmax = (a + [100000, 5000, 100][len(a):])[:3]
        X, y, _, _ = synth.make_logistic(n, p, 50)
        sess = capi.Session(X, y, data_type=2, model_type=2)
        run = lambda: sess.sequential_path(np.arange(1, kmax + 1), ic_type=3)  # noqa: E731
    elif fam == "cox":
        n, p, kmax = (a + [200000, 20000, 150][len(a):])[:3]
        X, _, st, _, _ = synth.make_cox(n, p, 75)
        sess = capi.Session(X, st, data_type=3, model_type=4)
        run = lambda: sess.sequential_path(np.arange(1, kmax + 1), ic_type=3)  # noqa: E731
    elif fam == "lmcv":
        n, p, kmax = (a + [50000, 10000, 200][len(a):])[:3]
        X, y, _, _ = synth.make_lm(n, p, 100)
        sess = capi.Session(X, y, data_type=1, model_type=1)
        sess.set_cv(5, synth.make_cv_folds(n, 5))
        run = lambda: sess.gs_path(1, kmax, ic_type=3, is_cv=True)  # noqa: E731
    elif fam == "poisson":  # SURVEY 8f rank 1 at the configs[2] shape
        n, p, kmax = (a + [100000, 5000, 100][len(a):])[:3]
        X, y, _, _ = synth.make_poisson(n, p, 50)
        sess = capi.Session(X, y, data_type=2, model_type=3)
        run = lambda: sess.sequential_path(np.arange(1, kmax + 1), ic_type=3)  # noqa: E731
    elif fam == "grouped":  # the configs[1] data as p / 5 groups of 5 columns
        n, p, kmax = (a + [50000, 10000, 40][len(a):])[:3]
        X, y, _, _ = synth.make_lm(n, p, 100)
        sess = capi.Session(X, y, data_type=1, model_type=1, algorithm_type=2, g_index=np.arange(0, p, 5, dtype=np.int32))
        run = lambda: sess.sequential_path(np.arange(1, kmax + 1), ic_type=3)  # noqa: E731
    elif fam == "lmcv-sharded":
        return sharded_cv(a)
    else:
        raise SystemExit("unknown family")
    setup = time.time() - t0
    del X
    import os
    for _ in range(int(os.environ.get("BENCH_FAMILY_WARMUP", "0"))):  # warm paths first (every path starts with empty caches)
        t0 = time.time()
        run()
        print("warm-up path: %.4f s" % (time.time() - t0), file=sys.stderr)
    sess.enable_kernel_timing(os.environ.get("BENCH_FAMILY_TIMING", "1") != "0")
    sess.score_pass_stats(reset=True)
    t0 = time.time()
    out = run()
    dt = time.time() - t0
    k1 = sess.score_pass_stats()
    ncand = out["n_candidates"]
    print(json.dumps({"family": fam, "n": n, "p": p, "kmax": kmax, "seconds": dt, "candidates": ncand,
                      "candidates_per_s": ncand / dt, "fits": out["n_fits"], "pdas_iters": out["n_pdas_iters"],
                      "score_pass_launches": k1["launches"], "score_pass_avg_ms": 1e3 * k1["seconds"] / max(k1["launches"], 1),
                      "score_pass_alg_GBps": (k1["algorithmic_bytes"] / k1["seconds"] / 1e9) if k1["seconds"] else None,
                      "selected_k": out["best_T0"], "ic": out["ic"], "setup_seconds": setup}))


def sharded_cv(a):
    import os
    import torch
    import torch.distributed as dist
    from bess_amd import dist as bdist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    one_dev = os.environ.get("BESSX_BENCH_ONE_DEVICE") == "1"
    dev = 0 if one_dev else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    backend = os.environ.get("BESSX_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    n, p, kmax = (a + [50000, 10000, 200][len(a):])[:3]
    X, y, _, _ = synth.make_lm(n, p, 100)
    sess = capi.Session(X, y, data_type=1, model_type=1, device=dev)
    sess.set_cv(5, synth.make_cv_folds(n, 5))
    del X
    comm_dev = "cuda" if backend == "nccl" else None
    times = []
    for rep in range(3):  # first repetition = warm-up
        cv = bdist.FoldShardedCV(sess, 5, world, rank, device=comm_dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        out = cv.gs_path(1, kmax)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        times.append(time.time() - t0)
    dt = min(times[1:])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"family": "lmcv-sharded", "n": n, "p": p, "smax": kmax, "world": world, "backend": backend,
                          "one_device": one_dev, "seconds": dt, "candidates": out["n_candidates"],
                          "candidates_per_s": out["n_candidates"] / dt, "fits": out["n_fits"],
                          "fits_per_s": out["n_fits"] / dt, "pdas_iters": out["n_pdas_iters"],
                          "evaluation_rounds": out["evaluations"], "selected_k": out["best_T0"], "cv_loss": out["ic"],
                          "units": "5 fold chains + the full-data chain, unit u on rank u % world"}))
    sess.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
