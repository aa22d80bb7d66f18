#!/usr/bin/env python3
"""bench.py -- candidate subsets solved per second on BASELINE configs[1]:
LM sequential path, synthetic Gaussian n=50000, p=10000, s.list = 1..200, GIC, warm start, max_iter 20.

A "step" is one pass of the hot path over one batch: the full 200-candidate warm-start chain
(Algorithm::fit to PDAS convergence + train_loss + ic per candidate) on data that is already
resident in HBM (upload + normalisation are untimed and reported separately).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n N --p P --kmax KMAX] [--no-cpu-baseline]

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling.  The units that shard are
independent candidate chains: every rank holds a replica of X and solves the same 200-candidate
path for its OWN response vector (same X beta, rank-specific noise seed), i.e. N independent
best-subset problems per step.  There is no data-path collective; the per-candidate IC curves are
all-gathered over RCCL at the end of each step (8 B per candidate), as north_star prescribes.
--shard kpath switches to strong scaling instead: ONE problem, s.list cut into N contiguous warm-start
chains (bess_amd/dist.py partition), IC curve all-gathered, best k picked on every rank.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_MFMA_PEAK_TFLOPS = 78.6  # dense fp64 matrix peak (same guide)


def make_problem(n, p, k_true, rank):
    """configs[1] inputs (bess_amd/synth.py).  rank 0 is exactly the BASELINE problem; other ranks
    replace the noise vector by a rank-seeded one (an independent problem on the same design)."""
    from bess_amd import synth
    X, y, support, beta = synth.make_lm(n, p, k_true)
    if rank > 0:
        rng = np.random.Generator(np.random.PCG64(synth.SEED_LM + 1000 * rank))
        y = X[:, support] @ beta[support] + rng.standard_normal(n)
    return X, y


def cpu_baseline(X, y, budget_s=30.0):
    """Time the CPU path on this host, one thread, on a bounded sample of the same workload: the first
    candidates (k = 1, 2, 3) of the same warm-start chain on the same full-size data.
    Preferred: the reference's own Eigen build (oracle/_ref/libbess_ref.so, compiled from the reference sources
    with the package flags -O2 -DNDEBUG -std=c++11; single-threaded by construction) -> kind "reference".
    Fallback when that library did not travel: the plain-C oracle -> kind "port".
    value = steady-state candidates/s = (kmax - 1) / (t[k=1..kmax] - t[k=1]), i.e. without the one-time copy /
    normalisation the GPU number also excludes."""
    from oracle import ref_ctypes as R
    from oracle import port_ctypes as P
    use_ref = R.available()
    run = (lambda seq: R.trace(X, y, ic_type=3, sequence=seq)) if use_ref else \
          (lambda seq: P.trace(X, y, ic_type=3, sequence=seq))
    t0 = time.time()
    run([1])
    t1 = time.time() - t0
    kmax = 3 if 3.5 * t1 < budget_s else 2
    t0 = time.time()
    run(list(range(1, kmax + 1)))
    tk = time.time() - t0
    per = max((tk - t1) / (kmax - 1), 1e-9)
    return {"value": 1.0 / per, "unit": "candidates/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "k=1..%d of the same sequential path on the full n=%d p=%d data: %.1f s (k=1 alone, incl. one-time "
                      "copy+normalise: %.1f s); value = steady-state rate, incl. set-up it is %.3f candidates/s; host has "
                      "%d cores, 1 used" % (kmax, X.shape[0], X.shape[1], tk, t1, kmax / tk, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--p", type=int, default=10000)
    ap.add_argument("--kmax", type=int, default=200)
    ap.add_argument("--k-true", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=["replica", "kpath"], default="replica")
    ap.add_argument("--score-mode", choices=["auto", "streaming", "covariance"], default="auto",
                    help="evaluation of the LM score pass (include/bessx.h, bessx_problem.score_mode)")
    ap.add_argument("--no-streaming-leg", action="store_true",
                    help="skip the extra (untimed-by-contract) measurement of the streaming score pass at N=1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bess_amd import capi
    from bess_amd import dist as bdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libbessx has no CPU path")
    # rehearsal of the N-rank path on a box with ONE GPU (never what the driver runs): all ranks share device 0 and
    # the IC curves travel over gloo -- BESSX_BENCH_ONE_DEVICE=1 BESSX_BENCH_BACKEND=gloo
    if os.environ.get("BESSX_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BESSX_BENCH_BACKEND", "nccl")
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    kpath = args.shard == "kpath" and distributed
    X, y = make_problem(args.n, args.p, args.k_true, 0 if kpath else rank)
    seq = np.arange(1, args.kmax + 1)
    if kpath:
        lo, hi = bdist.partition(args.kmax, world, rank)
        seq = seq[lo:hi]
    t0 = time.time()
    mode = {"auto": 0, "streaming": 1, "covariance": 2}[args.score_mode]
    sess = capi.Session(X, y, data_type=1, is_normal=True, model_type=1, max_iter=20, is_warm_start=True,
                        score_mode=mode, device=local_rank)
    covariance = sess.score_mode() == 2
    torch.cuda.synchronize()
    upload_s = time.time() - t0

    ic_curves = None
    out = None
    for _ in range(args.warmup):
        out = sess.sequential_path(seq, ic_type=3)
    sess.enable_kernel_timing(True)
    sess.score_pass_stats(reset=True)
    barrier()
    t0 = time.time()
    pdas_iters = 0
    for _ in range(args.steps):
        out = sess.sequential_path(seq, ic_type=3)
        pdas_iters += out["n_pdas_iters"]
        if kpath:  # gather the IC curve: the only collective of the path
            ic_curves = bdist.gather_curve(out["cand_ic"], args.kmax, world, rank, device=comm_dev)[None, :]
        elif distributed:
            ic_curves = bdist.gather_rows(out["cand_ic"], world, device=comm_dev)
    barrier()
    dt = time.time() - t0
    if distributed:
        tmax = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    k1 = sess.score_pass_stats()
    sess.enable_kernel_timing(False)

    def roofline_of(stats, cov):
        """HBM roofline of the kernel that streams X: algorithmic bytes (8 n p per pass over X) / its HIP-event time."""
        passes = stats["algorithmic_bytes"] / (8.0 * args.n * args.p)
        per_pass = stats["seconds"] / passes if passes else 0.0
        achieved = 8.0 * args.n * args.p / per_pass / 1e9 if passes else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and (args.n, args.p) == (50000, 10000):
            try:
                traffic = json.load(open(tpath)).get("k_cov_panel_hbm_bytes_per_pass" if cov else
                                                     "k_xtv_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kern = ("k_cov_panel_lds (X^T diag(m) X_S on the fp64 matrix cores: 32 new Gram columns per pass over X)"
                if cov else "k_xtv<8,16,false> (X^T r score pass)")
        roof = {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": 8.0 * args.n * args.p, "avg_launch_ms": 1e3 * per_pass,
                "launches_timed": stats["launches"], "passes_over_X_timed": passes}
        if cov and per_pass:
            # the same kernel against the other roof: 2 n p 32 flop per pass on the fp64 matrix cores
            tf = 2.0 * args.n * args.p * 32 / per_pass / 1e12
            roof["mfma_fp64"] = {"achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / FP64_MFMA_PEAK_TFLOPS, "flop_per_pass": 2.0 * args.n * args.p * 32}
        return roof

    if rank == 0:
        n_cand = args.kmax * args.steps * (1 if kpath else world)
        value = n_cand / dt
        roof = roofline_of(k1, covariance)
        try:  # the spec peak next to a ceiling measured on this very device (read + write of a D2D copy)
            roof["measured_stream_copy_GBps"] = capi.op_stream_copy_gbps(1 << 30, 10)
        except Exception:
            roof["measured_stream_copy_GBps"] = None
        line = {
            "metric": "candidate subsets solved/sec (n=50k,p=10k,k<=200 LM)", "value": value,
            "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if kpath else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: LM sequential path, Gaussian X n=%d p=%d, s.list=1..%d, GIC, "
                                   "warm start, max_iter=20, is_normal" % (args.n, args.p, args.kmax),
                       "candidates_per_step_per_gpu": args.kmax, "units_sharded": "independent candidate chains "
                       "(one response vector per rank on a replicated X)", "collective": "all_gather of the IC curve",
                       "score_pass": "covariance updates (cached Gram columns)" if covariance else "streaming"},
            "roofline": roof,
            "passes_over_X_per_candidate": roof["passes_over_X_timed"] / float(len(seq) * args.steps),
            "pdas_iterations_per_candidate": pdas_iters / float(len(seq) * args.steps),
            # I_k of SURVEY 8d: PDAS iterations Algorithm::fit took per candidate (identical to the reference's,
            # tests/test_fullsize_gpu.py), as a histogram {iterations: candidates}
            "pdas_iterations_histogram": {str(int(k)): int(v) for k, v in
                                          zip(*np.unique(out["cand_iters"], return_counts=True))},
            "upload_and_normalise_seconds": upload_s,
            "selected_k": int(out["best_T0"]), "selected_ic": float(out["ic"]),
        }
        if covariance and world == 1 and not args.no_streaming_leg:
            # the other evaluation of the same path (every PDAS iteration reads X once), for comparison
            sess.close()
            s2 = capi.Session(X, y, data_type=1, is_normal=True, model_type=1, max_iter=20, is_warm_start=True,
                              score_mode=1, device=local_rank)
            s2.sequential_path(seq, ic_type=3)
            s2.enable_kernel_timing(True)
            s2.score_pass_stats(reset=True)
            torch.cuda.synchronize()
            t1 = time.time()
            o2 = s2.sequential_path(seq, ic_type=3)
            torch.cuda.synchronize()
            d2 = time.time() - t1
            st2 = s2.score_pass_stats()
            s2.close()
            line["streaming_score_pass"] = {
                "value": len(seq) / d2, "unit": "candidates/s", "steps": 1, "roofline": roofline_of(st2, False),
                "passes_over_X_per_candidate": st2["launches"] / float(len(seq)),
                "same_selection": bool(int(o2["best_T0"]) == int(out["best_T0"]) and
                                       np.array_equal(np.nonzero(o2["beta"])[0], np.nonzero(out["beta"])[0]))}
        if ic_curves is not None:
            line["ic_curves_gathered"] = int(ic_curves.shape[0])
            line["best_k_per_problem"] = [int(bdist.select_best(c)) + 1 for c in ic_curves]
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(X, y)
            except Exception as e:  # the checker is optional for the measurement itself
                line["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": 1, "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        print(json.dumps(line))
    sess.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
