#!/usr/bin/env python3
"""bench.py -- candidate subsets solved per second on BASELINE configs[1]:
LM sequential path, synthetic Gaussian n=50000, p=10000, s.list = 1..200, GIC, warm start, max_iter 20.

A "step" is one pass of the hot path over one batch: the full 200-candidate path (the all-rows group_XTX pass of
src/path.cpp:37, then Algorithm::fit to PDAS convergence + train_loss + ic per candidate) on data that is already resident
in HBM (upload + normalisation are untimed and reported separately).

OUTPUT.  Rank 0 prints ONE line of strict JSON as the LAST line of stdout, at most 6 KB: the contract keys, `roofline`
(HIP-event timing of the kernel that streams X; covariance form: both roofs per launch width, the binding one on top),
`cpu_baseline`, the streaming leg's flat keys `roofline.streaming_*`, and a few figures per other config.  The full record
(every leg, segments of the CPU timing, histograms, the k-path report) goes to gpurun_out/bench_detail.json
(BESSX_BENCH_DETAIL_STDOUT=1: also as an EARLIER stdout line).

Two evaluations of the score pass are timed with the same --steps / --warmup at N = 1: the headline is the covariance
form (cached Gram columns; `config.headline_mode`), and the STREAMING formulation north_star prescribes -- every PDAS
iteration reads X once -- is in `roofline.streaming_*` (candidates/s, ms per step, k_xtv against the HBM roof, the whole
step against the HBM roof) and in `streaming_score_pass`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload lm-seq|lm-cv-gs] [--shard kpath|replica]
                  [--n N --p P --kmax KMAX] [--no-cpu-baseline]

--gpus N > 1 without a launcher (WORLD_SIZE unset): this process starts N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU)
BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the launcher's code.  Under a launcher,
WORLD_SIZE must equal --gpus.

N > 1, default (--shard kpath): STRONG scaling of the ONE configs[1] problem, partitioned as north_star says.  X is
replicated; s.list = 1..200 is cut into N contiguous chunks, each chunk one warm-start chain (bess_amd/dist.py
partition), stitched into the single chain (dist.StitchedKPath); NO data-path collective: RCCL carries the IC curve
(8 B per candidate) and the chunks' last models (a few KB per stitch round).  After the timed region rank 0 runs the
single chain and the line reports for how many k the supports agree (all of them, by construction of the stitch).
--coop-variant (opt-in): the same steps once more with the cooperative prefill / pilot policy (Gram column blocks
all-gathered between the ranks -- a data-path collective north_star does not have), as a second figure
`cooperative_prefill_variant` in the same line; --prefill / --pilot make that variant the timed one.
--shard replica: WEAK scaling, N independent problems (rank-specific response on the same design).
--workload cox-seq: BASELINE configs[4] (Cox PDAS, n=200000 p=20000 k<=150), the same k-path split: every candidate pays
its own passes over the 32 GB design, so this is the path whose chunks scale; every rank generates and uploads the
whole design (32 GB of host memory per rank while it does).
--chunk-start cold|ladder: how a chunk that begins at k0 > 1 gets there (see --help).
--workload lm-cv-gs: BASELINE configs[3] (gs_path on [1, kmax] under 5-fold CV), the K fold chains + the full-data
chain and, in the final sweep, (fold x s) pairs dealt to the ranks (bess_amd.dist.FoldShardedCV).

BESSX_BENCH_ONE_DEVICE=1 rehearses the N-rank path on a box with ONE GPU (ranks share device 0, collectives over
gloo) -- never what the driver runs.

At N = 1 with the default sizes the record also carries `other_configs`: one timed path each of BASELINE configs[2]
(logistic n=100k p=5k k<=100), configs[3] (LM gs_path + 5-fold CV) and configs[4] (Cox n=200k p=20k k<=150) with the
kernel that streams X against the HBM roof and the time per IRLS / Newton step (--no-other-configs skips them; they
add about two minutes, most of it generating and uploading the 32 GB Cox design).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_MFMA_PEAK_TFLOPS = 78.6  # dense fp64 matrix peak (same guide)
# the one full-path run of the compiled reference that exists (it produced tests/golden/fullsize_lm.npz)
RECORDED_FULL_PATH = {"seconds": 10038.0, "candidates": 200, "value": 200 / 10038.0, "unit": "candidates/s",
                      "cores": 1, "host": "build container (8 vCPU), not the GPU box",
                      "source": "tests/golden/fullsize_lm.npz: ref_wall_seconds (tests/golden/make_fullsize_ref.py lm)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=None, help="default: 50000 (lm-*), 200000 (cox-seq)")
    ap.add_argument("--p", type=int, default=None, help="default: 10000 (lm-*), 20000 (cox-seq)")
    ap.add_argument("--kmax", type=int, default=None, help="default: 200 (lm-*), 150 (cox-seq)")
    ap.add_argument("--k-true", type=int, default=None, help="default: 100 (lm-*), 75 (cox-seq)")
    ap.add_argument("--no-shared-design", action="store_true",
                    help="cox-seq with N > 1: every rank draws its own (identical, seeded) copy of the design instead of "
                         "mapping the one rank 0 leaves in /dev/shm")
    ap.add_argument("--workload", choices=["lm-seq", "lm-cv-gs", "cox-seq"], default="lm-seq",
                    help="lm-seq = BASELINE configs[1] (the metric); lm-cv-gs = configs[3]; cox-seq = configs[4]")
    ap.add_argument("--chunk-start", choices=["auto", "cold", "ladder", "lead"], default="auto",
                    help="N > 1, --shard kpath: how a chunk that does not begin at k = 1 reaches its first sparsity "
                         "level: cold = Algorithm::fit from the empty model at k0; ladder = a warm-start chain up the "
                         "levels k0/8, k0/4, k0/2 first (their candidates are discarded); auto = ladder from k0 = 128 "
                         "(measured on configs[1], tools/coldstart.py: 11.2 vs 12.2 ms at k0 = 176, but 7.9 vs 7.2 ms at "
                         "k0 = 101); lead (round 6, LM; what auto means for LM) = every rank first walks the ONE-GPU "
                         "path's coarse levels below its chunk and the level just below it as lead fits "
                         "(bessx_path_chain.lead_levels: the ranks repeat each other's few coarse fits, no communication), "
                         "then its chunk warm from the last lead model as chunk chains")
    ap.add_argument("--prefill", default="0",
                    help="N > 1, --shard kpath, LM covariance form: columns of the cooperative prefill of the Gram column "
                         "caches in front of the chunks (bess_amd.dist.cooperative_prefill: the ranks share the passes "
                         "over X their cold starts would repeat; ONE data-path all-gather of p x 32 blocks).  0 (default) "
                         "= replicas only, as north_star partitions the path: the collectives carry the IC curve and the "
                         "chunks' last models, nothing else; auto = the measured policy (320 columns from 3 ranks on)")
    ap.add_argument("--coop-variant", action="store_true",
                    help="N > 1, lm-seq: after the timed region of the default partition, time the SAME steps once more "
                         "with --prefill auto --pilot auto (Gram column blocks all-gathered between the ranks) and report "
                         "it as a second figure, `cooperative_prefill_variant`, in the same line")
    ap.add_argument("--pilot", default="none",
                    help="with the prefill: 'K,M2[,W]' = every rank runs the same pilot fit of sparsity level K on the "
                         "prefilled cache -- with W its own fills are shared too (W columns per fill: the missing ones and "
                         "the best uncached ones by that iteration's scores, one 32-column group per rank; "
                         "bessx_session_set_fill_hook) --, the M2 uncached columns its final scores rank highest are shared "
                         "as a second list and the chunks beyond K start warm from the pilot's model "
                         "(bess_amd.dist.pilot_prefill); 'none' (default); auto = K = 0.64 kmax rounded to 32, prefill K, M2 = 0, "
                         "W = 32 per rank (tools/coop_prefill.py, 8 ranks: slowest 5.2-5.6 ms against 7.5 ms without the "
                         "shared fills in the pilot, 9.5 ms with the marginal list alone, 12.8 ms without a prefill)")
    ap.add_argument("--rebalance", choices=["auto", "on", "off"], default="auto",
                    help="N > 1, k-path: after every step the chunk boundaries move towards equal time per rank "
                         "(bess_amd.dist.rebalance_bounds); auto = on up to 4 ranks (2 ranks: 13.7 -> 11.9 ms; no gain at 8)")
    ap.add_argument("--comm", choices=["torch", "bessx"], default="torch",
                    help="N > 1, k-path: the collectives of the step (all-gathers of the chunks' last models and of the IC "
                         "curve) through torch.distributed (default; backend nccl = RCCL) or through the library's own "
                         "communicator (bessx_comm_*: RCCL directly, what a C or R host has; its unique id is broadcast once "
                         "through torch.distributed at start-up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work per timed segment")
    ap.add_argument("--shard", choices=["auto", "replica", "kpath"], default="auto",
                    help="N > 1: kpath (default) = strong scaling of one problem; replica = weak scaling")
    ap.add_argument("--score-mode", choices=["auto", "streaming", "covariance"], default="auto",
                    help="evaluation of the LM score pass (include/bessx.h, bessx_problem.score_mode)")
    ap.add_argument("--no-streaming-leg", action="store_true",
                    help="skip the extra (untimed-by-contract) measurement of the streaming score pass at N=1")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the N=1 measurements of BASELINE configs[2], [3], [4] (other_configs in the line)")
    args = ap.parse_args(argv)
    args.chunk_start_option = args.chunk_start
    dn, dp, dk, dt = (200000, 20000, 150, 75) if args.workload == "cox-seq" else (50000, 10000, 200, 100)
    args.n = dn if args.n is None else args.n
    args.p = dp if args.p is None else args.p
    args.kmax = dk if args.kmax is None else args.kmax
    args.k_true = dt if args.k_true is None else args.k_true
    return args


def visible_gpus():
    """GPUs this process could use, found without the HIP / HSA runtime (the launching process stays clear of the GPU):
    none without /dev/kfd; the *_VISIBLE_DEVICES list if one is set; else the GPU agents in the kernel driver's topology
    files (an upper bound inside a container that exposes fewer).  None when nothing can be read -- the ranks then fail by
    themselves on a missing device."""
    if not os.path.exists("/dev/kfd"):
        return 0  # no compute driver node: no GPU for this process
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            return len([v for v in os.environ[var].split(",") if v.strip()])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        count = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
        return count
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """--gpus N > 1 and no launcher: start N fresh ranks as CHILD processes (never exec).  The parent does not import
    torch and makes no HIP call; the device count comes from the driver's topology files."""
    one_dev = os.environ.get("BESSX_BENCH_ONE_DEVICE") == "1"
    if not one_dev:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            print("bench.py: --gpus %d but %d visible (BESSX_BENCH_ONE_DEVICE=1 rehearses on one device)"
                  % (args.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # the ranks read their arguments from the environment: torchrun's own parser trips over script options that
    # are prefixes of its own (--n / --nnodes); --gpus is repeated on the command line for readability only
    env["BESSX_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus)]
    return subprocess.call(cmd, env=env)


def make_problem(n, p, k_true, rank):
    """configs[1] inputs (bess_amd/synth.py).  rank 0 is exactly the BASELINE problem; in the weak-scaling mode the
    other ranks replace the noise vector by a rank-seeded one (an independent problem on the same design)."""
    from bess_amd import synth
    X, y, support, beta = synth.make_lm(n, p, k_true)
    if rank > 0:
        rng = np.random.Generator(np.random.PCG64(synth.SEED_LM + 1000 * rank))
        y = X[:, support] @ beta[support] + rng.standard_normal(n)
    return X, y


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(X, y, gpu_out, sess_norm, kmax, budget_s):
    """The reference's Eigen path (oracle/_ref/libbess_ref.so: the reference sources compiled with the package's own
    flags -O2 -DNDEBUG -std=c++11, single-threaded by construction) timed on THIS host, pinned to one core, on a
    bounded sample of the same workload -- two segments of the same warm-start chain on the same full-size data:
    the head (k = 1, 2, ...) and the far end (k = 181, ... started from the k = 180 model, which is taken from the
    GPU path: same support, coefficients within 1e-6), each for about `budget_s` seconds.  A candidate = fit +
    train_loss + ic, as in sequential_path (src/path.cpp:48-74).  value = candidates timed / their seconds over
    both segments; set-up (copy + normalise) is excluded like the GPU figure excludes upload + normalise.
    Fallback when the compiled reference did not travel: the plain-C oracle on the head segment -> kind "port"."""
    from oracle import ref_ctypes as R
    from oracle import port_ctypes as P
    n, p = X.shape
    pinned = None
    try:
        cores = sorted(os.sched_getaffinity(0))
        pinned = cores[len(cores) // 2]
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        cores = []
    try:
        base = {"unit": "candidates/s", "cores": 1, "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
                "pinned_to_core": pinned, "recorded_full_path": RECORDED_FULL_PATH}
        if not R.available():
            t0 = time.time()
            P.trace(X, y, ic_type=3, sequence=[1])
            t1 = time.time() - t0
            t0 = time.time()
            P.trace(X, y, ic_type=3, sequence=[1, 2, 3])
            t3 = time.time() - t0
            base.update({"value": 2.0 / max(t3 - t1, 1e-9), "kind": "port",
                         "sample": "plain-C oracle, k=2..3 of the chain (k=1..3: %.1f s, k=1: %.1f s)" % (t3, t1)})
            return base
        head = R.time_chain(X, y, np.arange(1, kmax + 1), budget_s=budget_s)
        segs = [("k=1..%d" % len(head["seconds"]), head)]
        k0 = min(180, kmax - 2)
        if k0 >= 4:
            xm, xn, ym = sess_norm
            sup = gpu_out["cand_support"][k0 - 1][:k0]
            init = gpu_out["cand_beta"][k0 - 1][:k0] * xn[sup] / np.sqrt(float(n))  # back to the normalised scale
            tail = R.time_chain(X, y, np.arange(k0 + 1, kmax + 1), init_idx=sup, init_val=init, budget_s=budget_s)
            ok = list(tail["iters"]) == list(gpu_out["cand_iters"][k0:k0 + len(tail["iters"])])
            segs.append(("k=%d..%d (started from the k=%d model)" % (k0 + 1, k0 + len(tail["seconds"]), k0), tail))
        else:
            ok = None
        cnt = sum(len(s["seconds"]) for _, s in segs)
        sec = sum(float(np.sum(s["seconds"])) for _, s in segs)
        base.update({
            "value": cnt / sec, "kind": "reference",
            "segments": [{"candidates": name, "seconds_per_candidate": [round(float(v), 3) for v in s["seconds"]],
                          "pdas_iterations": [int(v) for v in s["iters"]],
                          "setup_seconds": round(s["setup_seconds"], 2)} for name, s in segs],
            "tail_iterations_match_gpu": ok,
            "sample": "%d candidates of the same chain on the full n=%d p=%d data in %.1f s: %s; package build "
                      "(-O2, 1 thread); an -O3 -march=x86-64-v3 -fopenmp build of the same sources is timed by "
                      "oracle/ref_cpu_timing.py (profiles/README.md)" % (cnt, n, p, sec, " + ".join(s for s, _ in segs))})
        return base
    finally:
        if cores:
            os.sched_setaffinity(0, set(cores))


def _pin_one_core():
    try:
        cores = sorted(os.sched_getaffinity(0))
        pinned = cores[len(cores) // 2]
        os.sched_setaffinity(0, {pinned})
        return cores, pinned
    except (AttributeError, OSError):
        return [], None


def _mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return float(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline_logistic(X, y, gpu_out, sess_norm, kmax, budget_s):
    """configs[2] beside its GPU figure: the compiled reference (GroupPdasLogistic + LogisticMetric, src/Algorithm.h:
    1148-1263, src/Metric.h:259-417) on the full n x p data, one core, two segments of the same warm-start chain -- the
    head (k = 1, ...) and the far end (k = kmax-1, ... started from the GPU path's k = kmax-2 model: same support) --
    each until the first candidate that ends beyond budget_s.  BASELINE.md section 3."""
    from oracle import ref_ctypes as R
    n, p = X.shape
    base = {"unit": "candidates/s", "cores": 1, "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
    if not R.available():
        return dict(base, value=None, kind="reference", sample="oracle/_ref/libbess_ref.so did not travel")
    cores, pinned = _pin_one_core()
    try:
        head = R.time_chain(X, y, np.arange(1, 4), budget_s=budget_s, data_type=2, model_type=2)
        segs = [("k=1..%d" % len(head["seconds"]), head)]
        k0 = kmax - 2
        xm, xn, _ = sess_norm
        sup = gpu_out["cand_support"][k0 - 1][:k0]
        bden = gpu_out["cand_beta"][k0 - 1][:k0]
        init = bden * xn[sup] / np.sqrt(float(n))                 # back to the normalised scale
        c0 = float(gpu_out["cand_coef0"][k0 - 1]) + float(np.dot(bden, xm[sup]))  # src/path.cpp:99-101 undone
        tail = R.time_chain(X, y, np.arange(k0 + 1, kmax + 1), init_idx=sup, init_val=init, init_coef0=c0,
                            budget_s=budget_s, data_type=2, model_type=2)
        segs.append(("k=%d..%d (started from the GPU path's k=%d model)" % (k0 + 1, k0 + len(tail["seconds"]), k0), tail))
        ok = list(tail["iters"]) == list(gpu_out["cand_iters"][k0:k0 + len(tail["iters"])])
        cnt = sum(len(sg["seconds"]) for _, sg in segs)
        sec = sum(float(np.sum(sg["seconds"])) for _, sg in segs)
        return dict(base, value=cnt / sec, kind="reference", pinned_to_core=pinned, tail_iterations_match_gpu=ok,
                    segments=[{"candidates": nm, "seconds_per_candidate": [round(float(v), 3) for v in sg["seconds"]],
                               "pdas_iterations": [int(v) for v in sg["iters"]],
                               "setup_seconds": round(sg["setup_seconds"], 2)} for nm, sg in segs],
                    sample="%d candidates of the same chain on the full n=%d p=%d data in %.1f s (%s); package build "
                           "(-O2, 1 thread)" % (cnt, n, p, sec, " + ".join(nm for nm, _ in segs)),
                    recorded_full_path={"seconds": 4038.0, "candidates": 100, "value": 100 / 4038.0,
                                        "host": "build container (8 vCPU), not the GPU box",
                                        "source": "tests/golden/fullsize_logistic.npz: ref_wall_seconds"})
    finally:
        if cores:
            os.sched_setaffinity(0, set(cores))


def cpu_baseline_cox(X_full, status_full, budget_s):
    """configs[4] beside its GPU figure.  The reference cannot run it (dense n x n risk-set matrix, src/Algorithm.h:1386:
    320 GB at n = 200000), so: (a) the compiled reference at n = 1000 / 2000 / 4000 (p = 400, k = 1..8, the recipe of
    configs[4]) -- documents its O(n^2); (b) the plain-C oracle (the build's own O(n k^2) restatement, kind "port") on
    the FULL n = 200000, p = 20000 data for the first candidates of the chain, when the host has the memory for its
    column-major copy of X (32 GB more)."""
    from bess_amd import synth
    from oracle import ref_ctypes as R
    from oracle import port_ctypes as P
    base = {"unit": "candidates/s", "cores": 1, "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
    cores, pinned = _pin_one_core()
    try:
        out = dict(base, pinned_to_core=pinned)
        if R.available():
            rows = []
            for n in (1000, 2000, 4000):
                Xs, _, st, _, _ = synth.make_cox(n, 400, 8)
                tc = R.time_chain(Xs, st, np.arange(1, 9), budget_s=budget_s, data_type=3, model_type=4)
                rows.append({"n": n, "p": 400, "candidates": int(len(tc["seconds"])),
                             "seconds": round(float(np.sum(tc["seconds"])), 3),
                             "candidates_per_s": float(len(tc["seconds"]) / max(np.sum(tc["seconds"]), 1e-9))})
            out["reference_small_n"] = {"kind": "reference", "rows": rows,
                                        "note": "compiled reference, k = 1..8 (or as many as fit %.0f s), p = 400: seconds grow "
                                                "~ n^2 (the n x n matrix of src/Algorithm.h:1386); n = 200000 would need "
                                                "320 GB" % budget_s}
        n, p = X_full.shape
        need = 8.0 * n * p / 1e9 + 8.0
        if _mem_available_gb() > need:
            t0 = time.time()
            P.trace(X_full, status_full, data_type=3, model_type=4, ic_type=3, sequence=[1])
            t1 = time.time() - t0
            setup_s, path_s = P.last_timing()
            out.update({"value": 1.0 / path_s, "kind": "port", "setup_seconds": round(setup_s, 1),
                        "sample": "plain-C oracle (oracle/bess_oracle.c, the O(n k^2) restatement, 1 thread) on the full "
                                  "n=%d p=%d data: candidate k = 1 of the chain in %.1f s (its column-major copy + "
                                  "normalisation of X, %.1f s, excluded like the GPU figure excludes upload + normalise; "
                                  "call %.1f s in all)" % (n, p, path_s, setup_s, t1)})
        else:
            out.update({"value": None, "kind": "port",
                        "sample": "skipped: the oracle's own copy of X needs %.0f GB of host memory, %.0f available"
                                  % (need, _mem_available_gb())})
        return out
    finally:
        if cores:
            os.sched_setaffinity(0, set(cores))


def measure_other_configs(local_rank, X_lm, y_lm, cpu_budget=None):
    """BASELINE configs[2] (logistic n=100k p=5k k<=100), configs[3] (LM gs_path + 5-fold CV on the configs[1] data)
    and configs[4] (Cox n=200k p=20k k<=150) on ONE GPU, one path each after one warm-up path (a path call starts from
    empty caches, so the repeat does the same work).  Per config: candidates/s, the kernel that streams X against the
    HBM roof (HIP events on the session stream), the whole path against the HBM roof, and where the rest goes (steps
    of the IRLS / Newton chains, time per step)."""
    import torch
    from bess_amd import capi, synth

    def timed(sess, run, n, p, traffic_key=None):
        run()  # warm-up (first use of a kernel loads its code object)
        sess.enable_kernel_timing(True)
        sess.score_pass_stats(reset=True)
        sess.submodel_steps(reset=True)
        cnt0 = sess.counters()
        torch.cuda.synchronize()
        t0 = time.time()
        out = run()
        torch.cuda.synchronize()
        dt = time.time() - t0
        k1 = sess.score_pass_stats()
        steps = sess.submodel_steps()
        sess.enable_kernel_timing(False)
        passes = k1["algorithmic_bytes"] / (8.0 * n * p)
        per_pass = k1["seconds"] / passes if passes else 0.0
        gbps = 8.0 * n * p / per_pass / 1e9 if per_pass else 0.0
        traffic = None
        if traffic_key:  # HBM bytes per pass from the counter passes of an earlier run of the same kernel at this size
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(traffic_key)
            except Exception:
                traffic = None
        rec = {
            "candidates": int(out["n_candidates"]), "candidates_per_s": out["n_candidates"] / dt, "ms_per_path": 1e3 * dt,
            "fits": int(out["n_fits"]), "pdas_iterations": int(out["n_pdas_iters"]),
            "passes_over_X": passes,
            "score_kernel": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": gbps / HBM_PEAK_GBPS, "avg_launch_ms": 1e3 * per_pass,
                             "algorithmic_bytes_per_pass": 8.0 * n * p, "measured_by": "HIP events on the session stream",
                             "traffic": traffic,
                             "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                "an earlier run of this kernel at this size, not measured in this run)"
                                                if traffic else None)},
            "whole_path_frac_of_hbm": passes * 8.0 * n * p / dt / 1e9 / HBM_PEAK_GBPS,
            # kernels of different chains overlap: the SUM of the score kernel's launch durations can exceed the wall time
            # of the path (round 4 printed 1 - sum / wall as a "share" and got negative time).  Reported as what was
            # measured: the wall time, the summed kernel time, their ratio; a share only where nothing overlaps.
            "time": dict({"wall_ms": 1e3 * dt, "summed_score_kernel_ms": 1e3 * k1["seconds"],
                          "summed_score_kernel_over_wall": k1["seconds"] / dt}),
            "selected_k": int(out["best_T0"]), "criterion": float(out["ic"]),
        }
        cnt = sess.counters()
        sp_l = cnt.get("shared_pass_launches", 0) - cnt0.get("shared_pass_launches", 0)
        if sp_l > 0:
            # round 6: the chunk chains share their passes over X (one multi-chain launch serves every chain at its score
            # pass): passes_over_X counts those launches, chain slots the vector sets they served
            rec["shared_passes"] = {"chains_served_per_pass": (cnt.get("shared_pass_chain_slots", 0) -
                                                              cnt0.get("shared_pass_chain_slots", 0)) / max(passes, 1.0),
                                    "launches_incl_fall_through": sp_l,
                                    "batches_cut_short": cnt.get("shared_pass_partial_batches", 0) -
                                    cnt0.get("shared_pass_partial_batches", 0)}
        overlapped = cnt.get("kpath_chunked_paths", 0) > 0 or cnt.get("cv_side_by_side_rounds", 0) > 0
        rec["time"]["chains_side_by_side"] = bool(overlapped)
        if not overlapped:
            rec["time"]["rest_of_the_chain_and_host_share"] = max(0.0, 1.0 - k1["seconds"] / dt)
        if steps:
            rec["submodel_steps"] = int(steps)
            if not overlapped:
                rec["us_per_submodel_step_incl_iteration_overheads"] = 1e6 * max(0.0, dt - k1["seconds"]) / steps
        if cnt.get("kpath_chunked_paths", 0) > 0:
            # the path ran as chunk chains side by side (bessx_kchunks.cpp): the same path as ONE chain on the same session
            # beside it (the kernel statistics above are the chunked run's: passes of several chains share the device)
            sess.set_kpath_chains(1)
            run()
            sess.enable_kernel_timing(True)
            sess.score_pass_stats(reset=True)
            sess.submodel_steps(reset=True)
            torch.cuda.synchronize()
            t1 = time.time()
            o1 = run()
            torch.cuda.synchronize()
            d1 = time.time() - t1
            k11 = sess.score_pass_stats()
            steps1 = sess.submodel_steps()
            sess.enable_kernel_timing(False)
            sess.set_kpath_chains(0)
            p1 = k11["algorithmic_bytes"] / (8.0 * n * p)
            if steps1:  # one chain: nothing overlaps, the remainder of the wall time is the sub-model chain + host
                rec["us_per_submodel_step_incl_iteration_overheads"] = 1e6 * max(0.0, d1 - k11["seconds"]) / steps1
                rec["us_per_submodel_step_measured_on"] = "the single chain (%d steps)" % steps1
            rec["chunk_chains"] = {
                "chains": cnt["kpath_chains_last_path"],
                "stitch_refits_per_path": cnt["kpath_stitch_refits"] / float(cnt["kpath_chunked_paths"]),
                "single_chain": {"candidates_per_s": o1["n_candidates"] / d1, "ms_per_path": 1e3 * d1, "passes_over_X": p1,
                                 # the score kernel with the device to itself (beside other chains' kernels a pass takes
                                 # longer: score_kernel.frac above is the chunked run's)
                                 "score_kernel_frac_of_hbm": (8.0 * n * p * p1 / k11["seconds"] / 1e9 / HBM_PEAK_GBPS)
                                 if k11["seconds"] else None},
                "same_candidates_as_the_single_chain": bool(
                    np.array_equal(o1["cand_support"], out["cand_support"]) and
                    np.array_equal(o1["cand_iters"], out["cand_iters"]) and
                    np.allclose(o1["cand_ic"], out["cand_ic"], rtol=1e-9, atol=0.0))}
        return rec

    res = {}
    # configs[3] first: it shares the configs[1] data
    n, p = X_lm.shape
    t0 = time.time()
    with capi.Session(X_lm, y_lm, data_type=1, model_type=1, device=local_rank) as sess:
        sess.set_cv(5, synth.make_cv_folds(n, 5))
        setup = time.time() - t0
        rec = timed(sess, lambda: sess.gs_path(1, 200, ic_type=3, is_cv=True), n, p)
        # the fold-sharded driver (bess_amd.dist.FoldShardedCV on bessx_session_cv_eval) at ONE rank, same session: what a
        # rank of `--workload lm-cv-gs --gpus N` runs, beside the in-library path it must not be slower than
        from bess_amd import dist as bdist
        bdist.FoldShardedCV(sess, 5).gs_path(1, 200)
        torch.cuda.synchronize()
        t1 = time.time()
        sh = bdist.FoldShardedCV(sess, 5).gs_path(1, 200)
        torch.cuda.synchronize()
        sharded = {"ms_per_path": 1e3 * (time.time() - t1), "fits": int(sh["n_fits"]),
                   "pdas_iterations": int(sh["n_pdas_iters"]), "selected_k": int(sh["best_T0"]),
                   "criterion": float(sh["ic"]), "evaluation_rounds": int(sh["evaluations"])}
    rec.update({"workload": "configs[3]: LM gs_path on [1,200] + 5-fold CV (fixed folds), n=%d p=%d; a candidate = the "
                            "full-data fit + 5 fold fits" % (n, p),
                "fits_per_s": rec["fits"] / (rec["ms_per_path"] / 1e3), "score_kernel_name": "k_cov_panel (32 Gram "
                "columns of every row set per pass over the fold-major copy of X)", "setup_seconds": setup})
    sharded["ratio_to_the_in_library_path"] = sharded["ms_per_path"] / rec["ms_per_path"]
    sharded["same_result"] = bool(sharded["selected_k"] == rec["selected_k"] and
                                  abs(sharded["criterion"] - rec["criterion"]) <= 1e-12 * abs(rec["criterion"]))
    rec["lmcv_sharded_world1"] = sharded
    res["lmcv"] = rec
    # SURVEY 8f rank 3, group selection: the configs[1] design as 2000 groups of 5 columns, 1..40 groups
    t0 = time.time()
    with capi.Session(X_lm, y_lm, data_type=1, model_type=1, algorithm_type=2, g_index=np.arange(0, p, 5, dtype=np.int32),
                      device=local_rank) as sess:
        setup = time.time() - t0
        rec = timed(sess, lambda: sess.sequential_path(np.arange(1, 41), ic_type=3), n, p)
    rec.update({"workload": "SURVEY 8f: grouped LM (algorithm_type 2), the configs[1] data as %d groups of 5 columns, "
                            "sequential path over 1..40 groups, GIC" % (p // 5),
                "score_kernel_name": "k_cov_panel (covariance form: d = X^T y - G_A beta_A from cached Gram columns, no "
                                     "pass over X per PDAS iteration; a parked fit fills 64 columns -- the missing ones and "
                                     "whole groups by this iteration's sacrifices -- in one pass; the per-group sacrifices "
                                     "follow in k_group_score, the blocks' diagonalisation cached per lambda)",
                "host_round_trips": "one per batch of two PDAS iterations (the selected groups are expanded to columns on "
                                    "the device, k_group_expand) + one per fill", "setup_seconds": setup})
    # the streaming form of the same path (score_mode = 1: X^T r by k_xtv, one pass over X per PDAS iteration; round 4's
    # figure before the covariance form reached the grouped fits)
    with capi.Session(X_lm, y_lm, data_type=1, model_type=1, algorithm_type=2, g_index=np.arange(0, p, 5, dtype=np.int32),
                      device=local_rank, score_mode=1) as sess:
        st = timed(sess, lambda: sess.sequential_path(np.arange(1, 41), ic_type=3), n, p)
    rec["streaming_form"] = {"candidates_per_s": st["candidates_per_s"], "ms_per_path": st["ms_per_path"],
                             "passes_over_X": st["passes_over_X"], "score_kernel": st["score_kernel"],
                             "same_selection": bool(st["selected_k"] == rec["selected_k"] and
                                                    abs(st["criterion"] - rec["criterion"]) <= 1e-9 * abs(rec["criterion"]))}
    res["grouped_lm"] = rec
    # L0L2 / bsrr -- the Powell path over (s, log lambda) with golden-section line searches
    with capi.Session(X_lm, y_lm, data_type=1, model_type=1, algorithm_type=5, device=local_rank) as sess:
        rec = timed(sess, lambda: sess.pgs_path(1, 200, 0.01, 100.0, n_lambda=100, powell_path=1, ic_type=3), n, p)
    rec.update({"workload": "SURVEY 8f: L0L2 (algorithm_type 5) Powell path pgs_path on s in [1,200] x lambda in [0.01,100], "
                            "golden-section line searches, configs[1] data, GIC; candidates = line searches + the final fit",
                "score_kernel_name": "k_cov_panel (covariance form)"})
    res["powell_l0l2"] = rec
    # sure independence screening in front of the path (10000 -> 2000 columns), then k = 1..100
    t0 = time.time()
    with capi.Session(X_lm, y_lm, data_type=1, model_type=1, is_screening=True, screening_size=2000,
                      device=local_rank) as sess:
        setup = time.time() - t0
        rec = timed(sess, lambda: sess.sequential_path(np.arange(1, 101), ic_type=3), n, 2000)
    rec.update({"workload": "SURVEY 8f: screening (SIS, src/screening.cpp:26-105) of the configs[1] design to its 2000 best "
                            "columns, then the sequential path k=1..100 on them; the screening itself is one two-accumulator "
                            "pass over the raw X at session creation",
                "session_creation_seconds_incl_upload_and_screening": setup,
                "score_kernel_name": "k_cov_panel on the 2000 kept columns (0.8 GB per pass)"})
    res["screened_lm"] = rec
    del X_lm
    # the reference's DEFAULT call: sequence = 1..min(p, n / log n) (python/bess/linear.py:285-287) at n = 25000, p = 3000
    nd, pd = 25000, 3000
    kd = min(pd, int(nd / np.log(nd)))
    Xd, yd, sup_d, _ = synth.make_lm(nd, pd, 12, seed=9)
    t0 = time.time()
    with capi.Session(Xd, yd, data_type=1, model_type=1, max_sparsity=kd, device=local_rank) as sess:
        setup = time.time() - t0
        del Xd
        rec = timed(sess, lambda: sess.sequential_path(np.arange(1, kd + 1), ic_type=4), nd, pd)
        cnt = sess.counters()
    rec.update({"workload": "the reference's default sequence 1..min(p, n / log n) = 1..%d (python/bess/linear.py:285-287), "
                            "LM n=%d p=%d, EBIC: sparsity levels far beyond the register-resident solvers" % (kd, nd, pd),
                "score_kernel_name": "k_cov_panel (32 Gram columns per pass over X; every column of the design is formed "
                                     "once: the cache holds all p columns)",
                "submodel": "k > 254: conjugate gradients on the dense copy of the cached Gram entries, one launch per "
                            "step over the whole chip (bessx_cgbig.hip); Cholesky hand-overs: %d" % cnt["cg_fallbacks"],
                "round_3_seconds": 24.0, "setup_seconds": setup})
    res["default_sequence"] = rec
    # SURVEY 8f rank 1, Poisson at the shape of configs[2]
    t0 = time.time()
    X, y, _, _ = synth.make_poisson(100000, 5000, 50)
    with capi.Session(X, y, data_type=2, model_type=3, device=local_rank) as sess:
        setup = time.time() - t0
        del X
        rec = timed(sess, lambda: sess.sequential_path(np.arange(1, 101), ic_type=3), 100000, 5000,
                    "k_xtv_two_accumulators_hbm_bytes_per_pass")
    rec.update({"workload": "SURVEY 8f: Poisson PDAS + IRLS (warm-started), sequential path k=1..100, n=100000 p=5000, GIC; "
                            "pinned by tests/golden/fullsize_poisson.npz (compiled reference)",
                "score_kernel_name": "k_xtv_mc<8,true> (X^T g and X^2^T h of every chunk chain at its score pass, one pass over X; "
                                     "single chain: k_xtv<8,16,true>)",
                "submodel": "IRLS step = k_irls_gram + k_gram_reduce + k_chol (3 launches)", "setup_seconds": setup})
    res["poisson"] = rec
    t0 = time.time()
    X, y, _, _ = synth.make_logistic(100000, 5000, 50)
    with capi.Session(X, y, data_type=2, model_type=2, device=local_rank) as sess:
        setup = time.time() - t0
        keep = {}
        rec = timed(sess, lambda: keep.setdefault("out", sess.sequential_path(np.arange(1, 101), ic_type=3)), 100000, 5000,
                    "k_xtv_two_accumulators_hbm_bytes_per_pass")
        norm = sess.normalization()
    if cpu_budget:
        try:
            rec["cpu_baseline"] = cpu_baseline_logistic(X, y, keep["out"], norm, 100, cpu_budget)
        except Exception as e:
            rec["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": 1, "kind": "reference",
                                   "sample": "failed: %r" % (e,)}
    del X, keep
    rec.update({"workload": "configs[2]: logistic PDAS + IRLS, sequential path k=1..100, n=100000 p=5000, GIC",
                "score_kernel_name": "k_xtv_mc<8,true> (X^T g and X^2^T h of every chunk chain at its score pass, one pass over X; "
                                     "single chain: k_xtv<8,16,true>)",
                "submodel": "IRLS step = k_irls_gram + k_gram_reduce + k_chol (3 launches)", "setup_seconds": setup})
    res["logistic"] = rec
    t0 = time.time()
    X, _, st, _, _ = synth.make_cox(200000, 20000, 75)
    with capi.Session(X, st, data_type=3, model_type=4, device=local_rank) as sess:
        setup = time.time() - t0
        if not cpu_budget:
            del X
        rec = timed(sess, lambda: sess.sequential_path(np.arange(1, 151), ic_type=3), 200000, 20000,
                    "k_cox_score1p_hbm_bytes_per_pass")
    if cpu_budget:
        try:
            rec["cpu_baseline"] = cpu_baseline_cox(X, st, cpu_budget)
        except Exception as e:
            rec["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": 1, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
        del X
    rec.update({"workload": "configs[4]: Cox PDAS, sequential path k=1..150, n=200000 p=20000 (32 GB X), GIC",
                "score_kernel_name": "k_cox_score1p_mc (risk-set scores of every chunk chain at its score pass, X read once; "
                                     "single chain: k_cox_score1p)",
                "submodel": "Newton step = linear predictor update, 4 scan launches, k_cox_hess (both Grams + gradient in "
                            "one pass over the active columns), carries, reduction, k_chol, direction, 3 line-search "
                            "launches", "setup_seconds": setup})
    res["cox"] = rec
    return res


LINE_LIMIT = 6000  # bytes: the driver keeps an 8 KB tail of stdout and parses the LAST line (round 5's 20 KB line: parsed null)
DETAIL_PATH = os.environ.get("BESSX_BENCH_DETAIL_PATH") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _round(v, nd=6):
    """Floats to `nd` significant digits, recursively (the final line is a report, not a checkpoint)."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None  # strict JSON has no NaN / Infinity
        return float("%.*g" % (nd, v))
    if isinstance(v, dict):
        return {k: _round(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_round(x, nd) for x in v]
    if isinstance(v, (np.floating,)):
        return _round(float(v), nd)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.bool_,)):
        return bool(v)
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full):
    """The ONE line the driver parses: the contract keys + `roofline` + `cpu_baseline` + a few flat figures per other
    config, at most LINE_LIMIT bytes of strict JSON.  Everything else (segments of the CPU timing, histograms, the per-config
    records, the k-path report) goes to gpurun_out/bench_detail.json."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data"]
    line = {k: full[k] for k in keep}
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ["workload", "candidates_per_step", "score_pass", "headline_mode", "units_sharded",
                                 "collective", "communicator", "ranks_with_work", "streaming_mode"])
    roof = dict(full.get("roofline", {}))
    roof.pop("traffic_source", None)
    line["roofline"] = roof
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ["value", "unit", "cores", "kind", "sample", "cpu_model", "host_cores",
                                         "tail_iterations_match_gpu"])
        for k in ("value", "unit", "cores", "kind", "sample"):
            line["cpu_baseline"].setdefault(k, cb.get(k))
        if cb.get("value"):
            line["gpu_over_cpu_baseline"] = full["value"] / cb["value"]
    for k in ("group_XTX_ms_inside_step", "value_excl_group_XTX", "ms_per_step_excl_group_XTX",
              "passes_over_X_per_candidate", "pdas_iterations_per_candidate", "selected_k", "selected_ic",
              "upload_and_normalise_seconds", "kpath_expected_speedup", "fits_per_s", "fits_per_step",
              "evaluation_rounds", "cv_loss", "pdas_iterations_per_step"):
        if k in full:
            line[k] = full[k]
    ws = full.get("whole_step")
    if ws:
        line["whole_step"] = {"frac_of_hbm_peak": ws.get("frac_of_hbm_peak"),
                              "kernel_streaming_X_share": ws.get("time", {}).get("kernel_streaming_X_share")}
    cc = full.get("chunk_chains")
    if cc:
        line["chunk_chains"] = {"chains": cc.get("chains"), "same_candidates_as_the_single_chain":
                                cc.get("same_candidates_as_the_single_chain"),
                                "single_chain_ms_per_path": cc.get("single_chain", {}).get("ms_per_path")}
    sp = full.get("streaming_score_pass")
    if sp:
        line["streaming_score_pass"] = _pick(sp, ["value", "unit", "steps", "warmup", "ms_per_step", "same_selection",
                                                  "same_candidates", "whole_step_frac_of_hbm_peak",
                                                  "passes_over_X_per_candidate"])
    kp = full.get("kpath_chunks_vs_single_chain")
    if kp:
        line["kpath"] = _pick(kp, ["chunks", "supports_equal_to_single_chain", "of", "best_k_chunked",
                                   "best_k_single_chain", "stitch_refits", "stitch_rounds", "prefill_columns",
                                   "ic_curve_max_rel_diff_to_single_chain", "one_gpu_ms_per_path_same_run",
                                   "speedup_over_one_gpu_same_run", "lead_levels_per_rank", "chunk_start"])
    cv = full.get("cooperative_prefill_variant")
    if cv:
        line["cooperative_prefill_variant"] = _pick(cv, ["value", "ms_per_step", "prefill_columns", "pilot"])
    oc = full.get("other_configs")
    if isinstance(oc, dict):
        small = {}
        for name, rec in oc.items():
            if not isinstance(rec, dict):
                small[name] = rec
                continue
            e = _pick(rec, ["candidates_per_s", "ms_per_path", "passes_over_X", "whole_path_frac_of_hbm", "fits_per_s"])
            sk = rec.get("score_kernel") or {}
            if sk.get("frac") is not None:
                e["score_kernel_frac_of_hbm"] = sk["frac"]
            sc = (rec.get("chunk_chains") or {}).get("single_chain") or {}
            if sc.get("score_kernel_frac_of_hbm") is not None:
                e["score_kernel_frac_alone"] = sc["score_kernel_frac_of_hbm"]
            if (rec.get("chunk_chains") or {}).get("chains") is not None:
                e["chains"] = rec["chunk_chains"]["chains"]
            if (rec.get("shared_passes") or {}).get("chains_served_per_pass") is not None:
                e["chains_per_pass"] = rec["shared_passes"]["chains_served_per_pass"]
            c2 = rec.get("cpu_baseline") or {}
            if c2.get("value"):
                e["cpu_baseline"] = {"value": c2["value"], "kind": c2.get("kind"), "cores": c2.get("cores")}
            small[name] = e
        line["other_configs"] = small
    line["detail"] = "gpurun_out/bench_detail.json (the full record of this run; also the FIRST stdout line of rank 0 when " \
                     "BESSX_BENCH_DETAIL_STDOUT=1)"
    line = _round(line)
    # never lose the line to its size: drop the least important parts until it fits
    for victim in ("detail", "other_configs", "streaming_score_pass", "chunk_chains", "kpath", "whole_step"):
        if len(json.dumps(line, allow_nan=False)) <= LINE_LIMIT:
            break
        if victim == "other_configs" and isinstance(line.get(victim), dict):
            line[victim] = {k: _pick(v, ["candidates_per_s", "score_kernel_frac_of_hbm"]) if isinstance(v, dict) else None
                            for k, v in line[victim].items()}
            if len(json.dumps(line, allow_nan=False)) <= LINE_LIMIT:
                break
        line.pop(victim, None)
    return line


def emit(full):
    """Rank 0's report: the full record to gpurun_out/bench_detail.json, the compact line as the LAST line of stdout."""
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as f:
            json.dump(_round(full, 9), f, indent=1, allow_nan=False)
    except (OSError, ValueError) as e:
        print("bench.py: could not write %s: %r" % (DETAIL_PATH, e), file=sys.stderr)
    if os.environ.get("BESSX_BENCH_DETAIL_STDOUT") == "1":
        print(json.dumps({"bench_detail": _round(full, 9)}, allow_nan=False))
    text = json.dumps(compact_line(full), allow_nan=False)
    assert len(text) <= LINE_LIMIT + 2000, len(text)
    print(text)
    sys.stdout.flush()


def main():
    inherited = os.environ.pop("BESSX_BENCH_ARGV", None)  # set by launch_ranks() for its children
    args = parse_args(json.loads(inherited) if inherited else None)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(world_env or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from bess_amd import capi, synth
    from bess_amd import dist as bdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libbessx has no CPU path")
    one_dev = os.environ.get("BESSX_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    backend = os.environ.get("BESSX_BENCH_BACKEND", "gloo" if one_dev else "nccl")
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

    def max_over_ranks(dt):
        if not distributed:
            return dt
        t = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.workload == "lm-cv-gs":
        return bench_cv(args, world, rank, local_rank, comm_dev, barrier, max_over_ranks)

    shard = args.shard if args.shard != "auto" else "kpath"
    kpath = shard == "kpath" and distributed
    cox = args.workload == "cox-seq"
    shared_files = []
    if cox and distributed and not args.no_shared_design:
        # configs[4], N ranks of one node: ONE host copy of the (replicated) 32 GB design.  Local rank 0 draws it and
        # leaves it in shared memory, the others map it read-only and upload from there (round 3: every rank drew
        # its own copy, 8 x 32 GB of host memory and 8 x the generator's time on the same cores)
        tag = "/dev/shm/bessx_cox_%s_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), args.n, args.p, args.k_true)
        shared_files = [tag + "_X.npy", tag + "_y.npy"]
        if rank == 0:
            import atexit
            atexit.register(lambda: [os.unlink(f) for f in shared_files if os.path.exists(f)])
            X, _, y, _, _ = synth.make_cox(args.n, args.p, args.k_true)
            np.save(shared_files[0], X)
            np.save(shared_files[1], y)
        dist.barrier()
        if rank != 0:
            X = np.load(shared_files[0], mmap_mode="r")
            y = np.load(shared_files[1])
    elif cox:
        # configs[4]: the whole (replicated) design; rows sorted by time, y = status
        X, _, y, _, _ = synth.make_cox(args.n, args.p, args.k_true)
    else:
        X, y = make_problem(args.n, args.p, args.k_true, 0 if (kpath or not distributed) else rank)
    full_seq = np.arange(1, args.kmax + 1)
    seq = full_seq
    lead = []  # sparsity levels in front of the chunk that only lead up to it (ladder start): run, timed, discarded
    if kpath:
        lo, hi = bdist.partition(args.kmax, world, rank)
        seq = full_seq[lo:hi]
        k0 = int(seq[0]) if len(seq) else 0
    t0 = time.time()
    mode = {"auto": 0, "streaming": 1, "covariance": 2}[args.score_mode]
    if cox:
        sess = capi.Session(X, y, data_type=3, model_type=4, max_iter=20, is_warm_start=True, device=local_rank)
    else:
        sess = capi.Session(X, y, data_type=1, is_normal=True, model_type=1, max_iter=20, is_warm_start=True,
                            score_mode=mode, device=local_rank)
    covariance = sess.score_mode() == 2
    torch.cuda.synchronize()
    upload_s = time.time() - t0
    if shared_files:  # every rank has uploaded: the shared host copy goes
        dist.barrier()
        if rank == 0:
            for f in shared_files:
                if os.path.exists(f):
                    os.unlink(f)

    ic_curves = None
    out = None
    stitch = None
    # N > 1, k-path: the chunks are stitched into the single warm-start chain every step (bess_amd.dist.StitchedKPath)
    def prefill_policy(opt_prefill, opt_pilot):
        """(prefill columns, pilot) of the k-path's optional cooperative prefill.  The DEFAULT (--prefill 0 --pilot none)
        is north_star's partition: replicas only.  "auto" = the policy measured on configs[1] (tools/coop_prefill.py,
        profiles/r04_lm_kpath_*_one_gpu.jsonl; one-device rehearsals, never on N devices): 2 ranks lose with any prefill
        (15.0 ms without, 16.1 with 320 columns); 4 ranks: pilot 10.0 ms, marginal list 11.8, none 13.4; 8 ranks: pilot
        7.5 ms, marginal list 9.5, none 11.6."""
        pf, pl = 0, None
        if not (kpath and covariance and not cox):
            return pf, pl
        kp = int(round(0.64 * args.kmax / 32.0)) * 32
        wide = 32 * world
        if opt_pilot not in ("auto", "none"):
            v = [int(q) for q in opt_pilot.split(",")]
            pl = (v[0], v[1] // 32 * 32) + ((v[2] // 32 * 32,) if len(v) > 2 else ())
        elif opt_pilot == "auto" and world >= 4 and 32 <= kp <= args.kmax - 8 and args.p >= 4 * (kp + 32 + 2 * wide):
            pl = (kp, 0, wide)
        if opt_prefill == "auto":
            pf = pl[0] if pl else (320 if world >= 3 else 0)
        else:
            pf = int(opt_prefill)
        pf = max(0, min(pf, (args.p // 64) * 32, 1024)) // 32 * 32
        if not pf and not (pl and len(pl) > 2):
            pl = None
        return pf, pl

    prefill, pilot = prefill_policy(args.prefill, args.pilot)
    any_ladder = False
    coarse_lead = False
    if kpath and not cox and not prefill and args.chunk_start in ("auto", "lead"):
        # LM (either score form): lead fits instead of a cold or ladder start (bess_amd.dist.StitchedKPath coarse_lead)
        coarse_lead = True
        args.chunk_start = "lead"
    if kpath and args.chunk_start == "lead" and not coarse_lead:
        args.chunk_start = "auto"  # (Cox, or with a cooperative prefill: the older rules)
    if kpath:
        k0_last = int(full_seq[bdist.partition(args.kmax, world, world - 1)[0]]) if args.kmax >= world else 0
        any_ladder = args.chunk_start == "ladder" or (args.chunk_start == "auto" and k0_last >= 128 and not cox
                                                      and not prefill)
        if args.chunk_start == "auto":
            # measured on configs[1], 8 chunks on one GPU (tools/coldstart.py, tools/coop_prefill.py, profiles/r04_*):
            # without the prefill the ladder saves Gram-column passes at large k0 (11.5 vs 12.5 ms for the slowest chunk),
            # on the prefilled cache the cold start is faster (9.5 vs 10.1 ms); Cox: every rung pays its own passes over
            # X and the cold start is faster at every k0 (tools/coldstart_cox.py)
            args.chunk_start = "ladder" if (k0 >= 128 and not cox and not prefill) else "cold"
        if args.chunk_start == "ladder" and lo > 0:
            lead = sorted({k for k in (k0 // 8, k0 // 4, k0 // 2) if 1 <= k < k0})
    # (the same decision on every rank: a ladder start anywhere rules the moving boundaries out)
    rebalance = kpath and not any_ladder and (args.rebalance == "on" or
                                              (args.rebalance == "auto" and world <= 4 and not cox))
    own_comm = None
    if kpath and args.comm == "bessx":
        ident = [bdist.BessxComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        own_comm = bdist.BessxComm(rank, world, ident[0], device=local_rank)
    stitched = bdist.StitchedKPath(sess, full_seq, world, rank, ic_type=3, lead=lead, device=comm_dev,
                                   prefill=prefill, pilot=pilot, rebalance=rebalance and not coarse_lead,
                                   coarse_lead=coarse_lead, comm=own_comm) if kpath else None
    for _ in range(args.warmup):
        out = stitched.step() if kpath else sess.sequential_path(seq, ic_type=3)
    sess.enable_kernel_timing(True)
    sess.score_pass_stats(reset=True)
    barrier()
    t0 = time.time()
    pdas_iters = 0
    for _ in range(args.steps):
        if kpath:
            stitch = stitched.step()  # chunk, stitch rounds, all-gather of the IC curve: the whole step of this rank
            out = stitch["chunk"] if stitch["chunk"] is not None else {"cand_iters": np.zeros(0, np.int32)}
            pdas_iters += int(np.sum(out["cand_iters"]))
            ic_curves = stitch["ic_curve"][None, :]
            continue
        out = sess.sequential_path(seq, ic_type=3)
        pdas_iters += out["n_pdas_iters"]
        if distributed:
            ic_curves = bdist.gather_rows(out["cand_ic"], world, device=comm_dev)
    barrier()
    dt = max_over_ranks(time.time() - t0)
    k1 = sess.score_pass_stats()
    cw = sess.counters()  # (panel launches by width, device time of the last group_XTX pass: read before any reset)
    sess.enable_kernel_timing(False)

    coop_variant = None
    if kpath and args.coop_variant and covariance and not cox:
        # opt-in second figure: the same steps with the cooperative prefill / pilot policy ("auto"), i.e. WITH the
        # data-path all-gathers of Gram column blocks north_star's partition does not have
        pf2, pl2 = prefill_policy("auto", "auto")
        if pf2 or pl2:
            st2 = bdist.StitchedKPath(sess, full_seq, world, rank, ic_type=3, lead=[], device=comm_dev, prefill=pf2,
                                      pilot=pl2, rebalance=False)
            for _ in range(args.warmup):
                st2.step()
            barrier()
            t2 = time.time()
            for _ in range(args.steps):
                r2 = st2.step()
            barrier()
            dt2 = max_over_ranks(time.time() - t2)
            coop_variant = {"value": args.kmax * args.steps / dt2, "unit": "candidates/s",
                            "ms_per_step": 1e3 * dt2 / args.steps, "steps": args.steps, "prefill_columns": pf2,
                            "pilot": list(pl2) if pl2 else None,
                            "ic_curve_equal_to_default_partition": bool(np.allclose(r2["ic_curve"], ic_curves[0],
                                                                                    rtol=1e-10, atol=0.0)),
                            "collective": "as the default + all_gathers of p x 32 Gram column blocks (%d bytes per rank "
                                          "and list)" % (max(pf2, 32) * args.p * 8)}
        else:
            coop_variant = {"value": None, "what": "the measured policy asks for no prefill at this N / size"}

    chunk_report = None
    if kpath:
        # compare the chunked chains with the single warm-start chain, after the timed region
        bounds = stitch["bounds"]  # (of the last timed step: the boundaries move when --rebalance is on)
        lo, hi = bounds[rank], bounds[rank + 1]
        sup = np.full((args.kmax, args.kmax), -1.0)
        if hi > lo:
            sup[lo:hi, :out["cand_support"].shape[1]] = out["cand_support"]
        allsup = bdist.gather_rows(sup.ravel(), world, device=comm_dev)
        if rank == 0:
            chunked = np.full((args.kmax, args.kmax), -1, dtype=np.int64)
            for r in range(world):
                a, b = bounds[r], bounds[r + 1]
                chunked[a:b] = allsup[r].reshape(args.kmax, args.kmax)[a:b]
            single = sess.sequential_path(full_seq, ic_type=3)
            # the same path on ONE GPU of this run (rank 0's device, the other ranks idle): what N ranks are measured against
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(3):
                single = sess.sequential_path(full_seq, ic_type=3)
            torch.cuda.synchronize()
            one_gpu_s = (time.time() - t1) / 3
            same = [bool(np.array_equal(chunked[k, :k + 1], single["cand_support"][k, :k + 1])) for k in range(args.kmax)]
            chunk_report = {
                "chunks": [[int(bounds[r]) + 1, int(bounds[r + 1])] for r in range(world)],
                "rebalance": bool(rebalance),
                "supports_equal_to_single_chain": int(np.sum(same)), "of": args.kmax,
                "differing_k": [int(k + 1) for k in range(args.kmax) if not same[k]][:40],
                "ic_curve_max_rel_diff_to_single_chain": float(np.max(np.abs(ic_curves[0] - single["cand_ic"]) /
                                                                      np.maximum(np.abs(single["cand_ic"]), 1e-300))),
                "best_k_chunked": int(stitch["best_k"]), "best_k_single_chain": int(single["best_T0"]),
                # last timed step: what every rank spent in its chunk (the parallel part) and in the stitch rounds
                "chunk_seconds_per_rank": [round(v, 5) for v in stitch["chunk_seconds_per_rank"]],
                "stitch_seconds_per_rank": [round(v, 5) for v in stitch["stitch_seconds_per_rank"]],
                "stitch_refits": stitch["stitch_refits"], "stitch_refits_per_rank": stitch["stitch_refits_per_rank"],
                "stitch_rounds": stitch["stitch_rounds"],
                "prefill_columns": prefill, "pilot": list(pilot) if pilot else None, "prefill_seconds_per_rank": [round(v, 5) for v in stitch["prefill_seconds_per_rank"]],
                "prefill": ("cooperative prefill: the %d columns with the largest marginal scores are formed once, their "
                            "32-column groups dealt to the ranks, the p x 32 blocks all-gathered (%d bytes per rank: a "
                            "data-path collective north_star's partitioning does not have; --prefill 0 = replicas only)"
                            % (prefill, prefill * args.p * 8) + ("; then a pilot fit of level %d on every rank, the %d columns its "
                            "scores rank highest shared the same way, chunks beyond it started warm from its model"
                            % (pilot[0], pilot[1]) if pilot else "") + ("; the pilot fit's own fills shared too, %d columns "
                            "per fill (the missing ones + the best uncached ones by that iteration's scores), one group per rank"
                            % pilot[2] if (pilot and len(pilot) > 2) else "")) if (prefill or pilot) else "none (replicas only)",
                "stitching": "after its chunk rank r re-fits its first candidates warm from rank r-1's last model until "
                             "a candidate coincides with its chunk's (same support, coefficients to 1e-9); the "
                             "candidates before that point are replaced: the gathered path IS the single chain's",
                "one_gpu_ms_per_path_same_run": 1e3 * one_gpu_s,
                "speedup_over_one_gpu_same_run": one_gpu_s / (dt / args.steps),
                "lead_levels_per_rank": [[int(v) for v in bdist.StitchedKPath(None, full_seq, world, r, coarse_lead=coarse_lead)
                                          .lead_levels()] for r in range(world)] if coarse_lead else None,
                "chunk_start": args.chunk_start_option,
                "chunk_start_meaning": "auto: LM -- lead fits (the one-GPU path's coarse levels below the chunk, then the level below it, on every rank; no communication); Cox -- ladder for chunks beginning at k0 >= 128, else cold; ladder: a chunk beginning at k0 > 1 first climbs the warm-start chain k0/8, k0/4, "
                                       "k0/2 (timed, candidates discarded); cold: Algorithm::fit from the empty model at k0"}

    def roofline_of(stats, cov, widths=None):
        """Roofline of the kernel that streams X, from its HIP-event time.  Algorithmic bytes = 8 n p per LAUNCH (X is read
        once per launch whatever the launch computes).  Streaming form (k_xtv / k_cox_score1p): the HBM roof.  Covariance
        form (k_cov_panel_dp): a launch forms one or two groups of 32 Gram columns, 2 n p 32 flop per group on the fp64
        matrix cores -- 8 flop/B for one group (below the 9.8 flop/B ridge: HBM binds), 16 flop/B for two (MFMA binds).
        `widths` = {groups per launch: (launches, seconds)}; both roofs are reported per width, the top-level roof is
        the one that needs more time over all the launches, and frac = achieved / peak against THAT roof."""
        launches = stats["algorithmic_bytes"] / (8.0 * args.n * args.p)
        per_launch = stats["seconds"] / launches if launches else 0.0
        gbps = 8.0 * args.n * args.p / per_launch / 1e9 if launches else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and (args.n, args.p) == (50000, 10000):
            try:
                tj = json.load(open(tpath))
                shared = (not cov) and stats.get("shared", False)
                traffic = tj.get("k_cov_panel_hbm_bytes_per_launch" if cov else
                                 ("k_xtv_mc_hbm_bytes_per_launch" if shared else "k_xtv_hbm_bytes_per_launch"))
                traffic_src = "profiles/pmc_traffic.json: " + tj.get("round6", {}).get(
                    "source", "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes") + " (counter passes are runs of " \
                    "their own, not part of this one)"

            except Exception:
                traffic = None
        kern = ("k_cov_panel_dp" if cov else ("k_cox_score1p" if cox else "k_xtv<8,16,false>"))
        roof = {"bound": "hbm", "kernel": kern, "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": 8.0 * args.n * args.p, "avg_launch_ms": 1e3 * per_launch,
                "launches_timed": stats["launches"], "hbm_frac": gbps / HBM_PEAK_GBPS}
        if cov and widths and sum(w[0] for w in widths.values()):
            flop_g = 2.0 * args.n * args.p * 32
            t_all = sum(w[1] for w in widths.values())
            f_all = sum(g * w[0] * flop_g for g, w in widths.items())
            t_hbm = sum(w[0] for w in widths.values()) * 8.0 * args.n * args.p / (HBM_PEAK_GBPS * 1e9)
            t_mfma = f_all / (FP64_MFMA_PEAK_TFLOPS * 1e12)
            binding = 0.0
            by = {}
            for g, (cnt, sec) in sorted(widths.items()):
                if not cnt:
                    continue
                avg = sec / cnt
                hf = 8.0 * args.n * args.p / avg / 1e9 / HBM_PEAK_GBPS
                mf = g * flop_g / avg / 1e12 / FP64_MFMA_PEAK_TFLOPS
                binding += cnt * max(8.0 * args.n * args.p / (HBM_PEAK_GBPS * 1e9), g * flop_g / (FP64_MFMA_PEAK_TFLOPS * 1e12))
                by["%d_group%s" % (g, "s" if g > 1 else "")] = {
                    "launches": int(cnt), "avg_launch_ms": 1e3 * avg, "hbm_frac": hf, "mfma_fp64_frac": mf,
                    "bound": "mfma" if mf > hf else "hbm", "flop_per_byte": g * flop_g / (8.0 * args.n * args.p)}
            tf = f_all / t_all / 1e12
            roof["by_width"] = by
            roof["mfma_fp64_frac"] = tf / FP64_MFMA_PEAK_TFLOPS
            roof["mfma_fp64_TFLOPs"] = tf
            roof["hbm_GBps"] = gbps
            roof["frac_sum_of_binding_roofs"] = binding / t_all  # sum over launches of max(t_HBM, t_MFMA) / sum of times
            roof["flop_per_group"] = flop_g
            if t_mfma > t_hbm:  # the matrix cores are the roof that needs more time over these launches
                roof.update({"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / FP64_MFMA_PEAK_TFLOPS, "dtype": "f64"})
        return roof

    def widths_of(cnt):
        return {1: (cnt.get("panel_launches_one_group", 0), cnt.get("panel_ns_one_group", 0) / 1e9),
                2: (cnt.get("panel_launches_two_groups", 0), cnt.get("panel_ns_two_groups", 0) / 1e9)}

    if rank == 0:
        n_cand = args.kmax * args.steps * (1 if (kpath or not distributed) else world)
        value = n_cand / dt
        roof = roofline_of(k1, covariance, widths_of(cw))
        roof["passes_over_X_timed"] = float(k1["launches"])
        try:  # the spec peak next to a ceiling measured on this very device (read + write of a D2D copy)
            roof["measured_stream_copy_GBps"] = capi.op_stream_copy_gbps(1 << 30, 10)
        except Exception:
            roof["measured_stream_copy_GBps"] = None
        step_s = dt / args.steps
        x_seconds = k1["seconds"] / args.steps
        # passes over X per step: the timed launches of the kernel that streams X + (LM) the group_XTX pass of every path
        x_passes = k1["launches"] / float(args.steps) + (0.0 if cox else 1.0)
        line = {
            "metric": ("candidate subsets solved/sec (Cox n=%dk,p=%dk,k<=%d)" % (args.n // 1000, args.p // 1000, args.kmax)
                       if cox else "candidate subsets solved/sec (n=50k,p=10k,k<=200 LM)"), "value": value,
            "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s, "higher_is_better": True,
            "scaling": "strong" if (kpath or not distributed) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("configs[4]: Cox PDAS sequential path, Gaussian X n=%d p=%d (rows sorted by time), "
                                    "s.list=1..%d, GIC, warm start, max_iter=20" % (args.n, args.p, args.kmax)) if cox else
                                   ("configs[1]: LM sequential path, Gaussian X n=%d p=%d, s.list=1..%d, GIC, "
                                    "warm start, max_iter=20, is_normal" % (args.n, args.p, args.kmax)),
                       "candidates_per_step": args.kmax * (1 if (kpath or not distributed) else world),
                       "units_sharded": ("contiguous chunks of s.list, one warm-start chain per rank, X replicated"
                                         if kpath else "independent problems (one response vector per rank on a "
                                         "replicated X)" if distributed else "none (one GPU)"),
                       "collective": ("all_gather of the IC curve + per stitch round one all_gather of the chunks' last "
                                      "models" + (" + one all_gather of Gram column blocks (cooperative prefill)"
                                                  if (kpath and prefill) else "")) if kpath else
                                     ("all_gather of the IC curve" if distributed else "none"),
                       "communicator": ("bessx_comm (RCCL directly, behind the C ABI)" if (kpath and args.comm == "bessx") else
                                        ("torch.distributed" if distributed else None)),
                       "score_pass": "covariance updates (cached Gram columns)" if covariance else "streaming",
                       "headline_mode": ("covariance form (Gram-cached: an exact re-formulation of the score pass, same "
                                         "candidates; NOT the streaming formulation SURVEY 8d's algorithmic bytes are "
                                         "written for -- that one is in roofline.streaming_*)") if covariance else
                                        ("streaming (every PDAS iteration reads X once)" if not cox else "Cox score pass"),
                       # k-path chunks: every rank owns a chunk of at least one candidate as long as N <= kmax
                       "ranks_with_work": min(world, args.kmax) if kpath else world,
                       "chunk_start": (args.chunk_start_option if kpath else None)},
            "roofline": roof,
            # the WHOLE step against the HBM roof: bytes the step streams from X / step time, and where the time goes
            "whole_step": {"bytes_streamed_from_X": x_passes * 8.0 * args.n * args.p,
                           "achieved_GBps": x_passes * 8.0 * args.n * args.p / step_s / 1e9,
                           "frac_of_hbm_peak": x_passes * 8.0 * args.n * args.p / step_s / 1e9 / HBM_PEAK_GBPS,
                           # (passes over X never overlap each other: one chain at a time holds the fill right -- staged
                           # fills run beside other chains' selection / solve kernels, not beside another pass -- so the
                           # summed launch time of the kernel that streams X is a true share of the step)
                           "time": {"wall_ms": 1e3 * step_s, "summed_kernel_ms_streaming_X": 1e3 * x_seconds,
                                    "kernel_streaming_X_share": min(1.0, x_seconds / step_s),
                                    "candidate_chain_and_host_share (selection, k x k solve, p x k GEMV, publish)":
                                    max(0.0, 1.0 - x_seconds / step_s)}},
            "passes_over_X_per_candidate": x_passes / float(max(len(seq), 1)),
            "pdas_iterations_per_candidate": pdas_iters / float(max(len(seq), 1) * args.steps),
            # I_k of SURVEY 8d: PDAS iterations Algorithm::fit took per candidate (identical to the reference's,
            # tests/test_fullsize_gpu.py), as a histogram {iterations: candidates}
            "pdas_iterations_histogram": {str(int(k)): int(v) for k, v in
                                          zip(*np.unique(out["cand_iters"], return_counts=True))},
            "upload_and_normalise_seconds": upload_s,
            "selected_k": int(out["best_T0"]) if not kpath else chunk_report["best_k_chunked"],
            "selected_ic": float(out["ic"]) if not kpath else float(np.min(ic_curves[0])),
        }
        if chunk_report:
            line["kpath_chunks_vs_single_chain"] = chunk_report
            # Amdahl's arithmetic of the partition north_star names (X replicated, contiguous chunks of s.list, no data-path
            # collective): the lead fits' passes over X are repeated by every rank, only the chunk phase divides by N
            line["kpath_expected_speedup"] = {
                "measured_this_run": chunk_report["speedup_over_one_gpu_same_run"],
                # every rank's step timed ALONE on one GPU, no communication (profiles/r06_kpath_lead_fits_one_gpu_probe.jsonl,
                # r06_cox_kpath_one_gpu_probe.jsonl): what the partition can reach on N devices
                "one_gpu_rehearsal_by_N": ({"2": 1.72, "4": 1.78, "8": 3.25} if cox else {"2": 0.98, "4": 1.04, "8": 1.10}),
                "bound": "T1 / (T_replicated + T_chunks / N): the passes over X that fill the Gram column cache are "
                         "repeated on every rank (DESIGN.md section 6)"}
        if coop_variant:
            line["cooperative_prefill_variant"] = coop_variant
        norm = sess.normalization() if world == 1 else None
        cnt = sess.counters() if world == 1 else {}
        if not cox:
            # the reference forms X^T y / diag(X^T X) inside every sequential_path call (group_XTX, src/path.cpp:37) and so
            # does every path call here (bessx_paths.cpp run_path): that pass over X IS in the step.  Its device time (HIP
            # events, last path) and the step without it as side keys (rounds 1-5 had it at session creation):
            gx_ms = cw.get("group_XTX_ns", 0) / 1e6
            line["group_XTX_ms_inside_step"] = gx_ms
            line["ms_per_step_excl_group_XTX"] = 1e3 * step_s - gx_ms
            line["value_excl_group_XTX"] = (n_cand / args.steps) / max(step_s - gx_ms / 1e3, 1e-12)
        if world == 1 and cnt.get("kpath_chunked_paths", 0) > 0:
            # the path ran as chunk chains side by side on one Gram column cache, stitched into the single warm-start chain
            # (bessx_kchunks.cpp); the same path as ONE chain on the same session, timed the same way, beside it
            runs = float(cnt["kpath_chunked_paths"])
            sess.set_kpath_chains(1)
            for _ in range(2):
                sess.sequential_path(seq, ic_type=3)
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(5):
                o1 = sess.sequential_path(seq, ic_type=3)
            torch.cuda.synchronize()
            d1 = (time.time() - t1) / 5
            sess.set_kpath_chains(0)
            line["chunk_chains"] = {
                "chains": cnt["kpath_chains_last_path"], "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                "stitch_refits_per_path": cnt["kpath_stitch_refits"] / runs,
                "fills_in_the_chunk_phase_per_path": cnt["kpath_chunk_fills"] / runs,
                "single_chain": {"value": len(seq) / d1, "unit": "candidates/s", "ms_per_path": 1e3 * d1},
                "same_candidates_as_the_single_chain": bool(
                    np.array_equal(o1["cand_support"], out["cand_support"]) and
                    np.array_equal(o1["cand_iters"], out["cand_iters"]) and
                    np.allclose(o1["cand_ic"], out["cand_ic"], rtol=1e-9, atol=0.0)),
                "what": "a coarse warm-start chain over the chunk boundaries fills the cache and starts every chunk; the "
                        "chunks run side by side on fit contexts of their own (a stream and a host thread each), fills "
                        "of the shared cache while every other chain stands still; stitched like the multi-GPU k-path"}
        if covariance and world == 1 and not args.no_streaming_leg and not cox:
            # the other evaluation of the same path (every PDAS iteration reads X once), for comparison
            sess.close()
            s2 = capi.Session(X, y, data_type=1, is_normal=True, model_type=1, max_iter=20, is_warm_start=True,
                              score_mode=1, device=local_rank)
            # ... timed exactly like the headline: the same --warmup and --steps, wall clock around the steps, HIP events
            # on every launch of the kernel that streams X.  This is the formulation north_star prescribes (SURVEY 7/8d:
            # every PDAS iteration reads X once, src/Algorithm.h:1109); the headline's covariance form is a separate mode.
            for _ in range(args.warmup):
                s2.sequential_path(seq, ic_type=3)
            s2.enable_kernel_timing(True)
            s2.score_pass_stats(reset=True)
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(args.steps):
                o2 = s2.sequential_path(seq, ic_type=3)
            torch.cuda.synchronize()
            d2 = (time.time() - t1) / args.steps
            st2 = s2.score_pass_stats()
            cnt2 = s2.counters()
            single2 = None
            if cnt2.get("kpath_chunked_paths", 0) > 0:
                # the streaming path ran as chunk chains (round 5: one chain's selection, Gram panel, solve and residual
                # beside another's pass over X -- the passes of several chains share the device, so a launch of the kernel
                # that streams X takes longer than alone).  The same path as ONE chain on the same session beside it: the
                # kernel with the device to itself.
                s2.set_kpath_chains(1)
                s2.sequential_path(seq, ic_type=3)
                s2.score_pass_stats(reset=True)
                torch.cuda.synchronize()
                t1 = time.time()
                n1 = min(args.steps, 3)
                for _ in range(n1):
                    o1 = s2.sequential_path(seq, ic_type=3)
                torch.cuda.synchronize()
                d1 = (time.time() - t1) / n1
                st1 = s2.score_pass_stats()
                single2 = {"value": len(seq) / d1, "unit": "candidates/s", "ms_per_step": 1e3 * d1, "steps": n1,
                           "roofline": roofline_of(st1, False),
                           "whole_step_frac_of_hbm_peak": st1["launches"] / float(n1) * 8.0 * args.n * args.p / d1 / 1e9
                           / HBM_PEAK_GBPS,
                           "same_candidates_as_the_chunk_chains": bool(
                               np.array_equal(o1["cand_support"], o2["cand_support"]) and
                               np.array_equal(o1["cand_iters"], o2["cand_iters"]))}
            s2.close()
            shared = cnt2.get("shared_pass_launches", 0) > 0
            st2["shared"] = shared
            roof2 = roofline_of(st2, False)
            if shared:
                # round 6: the chunk chains share their passes (k_xtv_mc: one launch streams X for every chain that is at its
                # score pass).  Algorithmic bytes per launch: 8 n p + 16 n C (the C chains' vectors in, their row-block sums
                # are p-vectors); passes per candidate count LAUNCHES, chain slots the vector sets served by them
                slots = cnt2.get("shared_pass_chain_slots", 0) / float(max(st2["launches"], 1))
                roof2["kernel"] = "k_xtv_mc<8,false> (X^T r of every chunk chain that is at its score pass, X read once)"
                roof2["chains_served_per_launch"] = slots
                roof2["algorithmic_bytes_per_launch"] = 8.0 * args.n * args.p + 16.0 * args.n * slots
            whole2 = st2["launches"] / float(args.steps) * 8.0 * args.n * args.p / d2 / 1e9 / HBM_PEAK_GBPS
            line["streaming_score_pass"] = {
                "value": len(seq) / d2, "unit": "candidates/s", "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * d2, "roofline": roof2,
                "whole_step_frac_of_hbm_peak": whole2,
                "passes_over_X_per_candidate": st2["launches"] / float(len(seq) * args.steps),
                "same_selection": bool(int(o2["best_T0"]) == int(out["best_T0"]) and
                                       np.array_equal(np.nonzero(o2["beta"])[0], np.nonzero(out["beta"])[0])),
                "same_candidates": bool(np.array_equal(o2["cand_support"], out["cand_support"]) and
                                        np.array_equal(o2["cand_iters"], out["cand_iters"]))}
            # flat keys inside the dictionaries every consumer of the line keeps (the nested leg above is easy to drop)
            roof["streaming_candidates_per_s"] = len(seq) / d2
            roof["streaming_ms_per_step"] = 1e3 * d2
            roof["streaming_steps"] = args.steps
            roof["streaming_kernel"] = ("k_xtv_mc<8,false> (one pass over X serves every chunk chain at its score pass)" if shared
                                        else "k_xtv<8,16,false> (X^T r, one pass over X per PDAS iteration)")
            if shared:
                roof["streaming_chains_served_per_pass"] = roof2["chains_served_per_launch"]
                roof["streaming_shared_kernel_frac"] = roof2["frac"]
                roof["streaming_shared_kernel_avg_launch_ms"] = roof2["avg_launch_ms"]
            # the kernel against the HBM roof: with the device to itself (single chain) where the leg ran as chunk chains
            ksrc = single2["roofline"] if single2 else roof2
            roof["streaming_kernel_frac"] = ksrc["frac"]
            roof["streaming_kernel_avg_launch_ms"] = ksrc["avg_launch_ms"]
            if single2:
                line["streaming_score_pass"]["chunk_chains"] = {"chains": cnt2.get("kpath_chains_last_path"),
                                                                 "single_chain": single2}
                roof["streaming_chains"] = cnt2.get("kpath_chains_last_path")
                roof["streaming_kernel_frac_beside_other_chains"] = roof2["frac"]
                roof["streaming_single_chain_candidates_per_s"] = single2["value"]
                roof["streaming_single_chain_whole_step_frac"] = single2["whole_step_frac_of_hbm_peak"]
            roof["streaming_whole_step_frac"] = whole2
            roof["streaming_passes_over_X_per_candidate"] = st2["launches"] / float(len(seq) * args.steps)
            line["config"]["streaming_mode"] = ("roofline.streaming_*: the same path with score_mode = streaming (every "
                                                "PDAS iteration reads X once), same --steps / --warmup")
        if ic_curves is not None and not kpath:
            line["ic_curves_gathered"] = int(ic_curves.shape[0])
            line["best_k_per_problem"] = [int(bdist.select_best(c)) + 1 for c in ic_curves]
        if not args.no_cpu_baseline and world == 1 and not cox:
            try:
                line["cpu_baseline"] = cpu_baseline(X, y, out, norm, args.kmax, args.cpu_budget)
            except Exception as e:  # the checker is optional for the measurement itself
                line["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": 1, "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        if world == 1 and not cox and not args.no_other_configs and (args.n, args.p, args.kmax) == (50000, 10000, 200):
            sess.close()
            try:
                line["other_configs"] = measure_other_configs(local_rank, X, y,
                                                              None if args.no_cpu_baseline else args.cpu_budget)
            except Exception as e:  # never lose the headline line to a secondary measurement
                line["other_configs"] = {"error": repr(e)}
        emit(line)
    sess.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def bench_cv(args, world, rank, local_rank, comm_dev, barrier, max_over_ranks):
    """BASELINE configs[3]: gs_path on [1, kmax] with 5-fold CV, fold chains (and, in the final sweep, fold x s
    pairs) dealt to the ranks; a step = one whole golden-section path.  candidates = distinct (s) evaluated by the
    path; fits = Algorithm::fit calls (full data + folds)."""
    import torch
    import torch.distributed as dist
    from bess_amd import capi, synth
    from bess_amd import dist as bdist
    X, y, _, _ = synth.make_lm(args.n, args.p, args.k_true)
    t0 = time.time()
    sess = capi.Session(X, y, data_type=1, model_type=1, device=local_rank)
    sess.set_cv(5, synth.make_cv_folds(args.n, 5))
    del X
    torch.cuda.synchronize()
    upload_s = time.time() - t0
    cdev = "cuda" if comm_dev == "cuda" else None
    out = None
    for _ in range(args.warmup):
        out = bdist.FoldShardedCV(sess, 5, world, rank, device=cdev).gs_path(1, args.kmax)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        out = bdist.FoldShardedCV(sess, 5, world, rank, device=cdev).gs_path(1, args.kmax)
    barrier()
    dt = max_over_ranks(time.time() - t0)
    if rank == 0:
        emit({
            "metric": "candidate subsets solved/sec (n=50k,p=10k, LM gs_path + 5-fold CV)",
            "value": out["n_candidates"] * args.steps / dt, "unit": "candidates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3]: LM golden-section path on [1,%d] + 5-fold CV, Gaussian X n=%d p=%d, "
                                   "fixed folds (synth.make_cv_folds)" % (args.kmax, args.n, args.p),
                       "units_sharded": "5 fold chains + the full-data chain (unit u on rank u %% N); final sweep: "
                                        "(fold x s) pairs", "collective": "all_gather of the fit records"},
            "fits_per_s": out["n_fits"] * args.steps / dt, "fits_per_step": int(out["n_fits"]),
            "pdas_iterations_per_step": int(out["n_pdas_iters"]), "evaluation_rounds": int(out["evaluations"]),
            "selected_k": int(out["best_T0"]), "cv_loss": float(out["ic"]), "upload_and_normalise_seconds": upload_s})
    sess.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _report_rank_failure(exc):
    """A rank that dies says so itself: `[rank r] <exception> | bessx_last_error` on stdout, stderr and in a per-rank
    file (BESSX_BENCH_ERRDIR, default gpurun_out/bench_errors under the repo) -- the launcher's epilogue and the other
    ranks' secondary "connection closed by peer" tracebacks bury the one message that matters."""
    import traceback
    rank = os.environ.get("RANK", "0")
    last = ""
    try:
        from bess_amd import capi
        last = capi.last_error()
    except Exception as e2:  # (the library itself may be what failed)
        last = "<bessx_last_error unavailable: %r>" % (e2,)
    head = "[rank %s] %s: %s | bessx_last_error: %s" % (rank, type(exc).__name__, exc, last)
    text = head + "\n" + "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))
    print(head, flush=True)
    print(text, file=sys.stderr, flush=True)
    try:
        d = os.environ.get("BESSX_BENCH_ERRDIR") or os.path.join(ROOT, "gpurun_out", "bench_errors")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "rank%s_pid%d.err" % (rank, os.getpid())), "w") as f:
            f.write("argv: %s\n%s" % (json.dumps(sys.argv), text))
    except OSError:
        pass


def _arm_watchdog():
    """BESSX_BENCH_WATCHDOG_S=<seconds>: a rank that is still running after that long dumps the Python stack of every
    thread into its per-rank file (and stderr) and exits -- a hang then says where it sits (tools/soak_bench_ranks.py)."""
    secs = float(os.environ.get("BESSX_BENCH_WATCHDOG_S", "0") or 0)
    if secs <= 0:
        return
    if os.environ.get("WORLD_SIZE") is None and parse_args(None).gpus > 1:
        return  # (the launching parent of an N-rank run only waits for its children: they carry the watchdog)
    import faulthandler
    d = os.environ.get("BESSX_BENCH_ERRDIR") or os.path.join(ROOT, "gpurun_out", "bench_errors")
    os.makedirs(d, exist_ok=True)
    f = open(os.path.join(d, "rank%s_pid%d.watchdog" % (os.environ.get("RANK", "0"), os.getpid())), "w")
    f.write("watchdog armed: %s s, argv %s\n" % (secs, json.dumps(sys.argv)))
    f.flush()
    faulthandler.dump_traceback_later(secs, exit=True, file=f)
    globals()["_watchdog_file"] = f  # (kept open for the dump)


if __name__ == "__main__":
    _arm_watchdog()
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001 -- reported, then the rank exits non-zero
        _report_rank_failure(exc)
        sys.exit(1)
