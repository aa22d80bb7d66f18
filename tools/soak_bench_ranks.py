"""Soak of the multi-process rehearsals of bench.py on ONE device (the commands tests/test_zz_bench_ranks_gpu.py runs):
the same command over and over, each time in fresh processes; the first non-zero exit stops the soak and everything
the ranks said (stdout, stderr, the per-rank files of bench._report_rank_failure) is kept.
  python tools/soak_bench_ranks.py [repetitions] [outdir] [-- bench args]
SOAK_HOLD_GPU=1: this process holds a HIP context of its own on the device while the ranks run (what the pytest
process of the driver's run does); SOAK_STRESS=N: N busy host processes beside the ranks (a loaded box).
Default command: the one that failed on the driver's box in round 4 (GPUTEST_r04.json):
  BESSX_BENCH_ONE_DEVICE=1 python bench.py --n 3000 --p 800 --kmax 30 --k-true 10 --steps 2 --warmup 1 --gpus 2
      --workload lm-cv-gs --no-cpu-baseline"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
extra = None
if "--" in argv:
    i = argv.index("--")
    argv, extra = argv[:i], argv[i + 1:]
reps = int(argv[0]) if argv else 50
outdir = os.path.abspath(argv[1]) if len(argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_bench_ranks")
bench_args = extra or ["--n", "3000", "--p", "800", "--kmax", "30", "--k-true", "10", "--steps", "2", "--warmup", "1",
                       "--gpus", "2", "--workload", "lm-cv-gs", "--no-cpu-baseline"]
os.makedirs(outdir, exist_ok=True)
env = dict(os.environ)
env["BESSX_BENCH_ONE_DEVICE"] = "1"
env["BESSX_BENCH_ERRDIR"] = os.path.join(outdir, "errors")
env.setdefault("BESSX_BENCH_WATCHDOG_S", "60")  # a rank still alive after a minute dumps its Python stacks and exits
shutil.rmtree(env["BESSX_BENCH_ERRDIR"], ignore_errors=True)
stress = []
if int(os.environ.get("SOAK_STRESS", "0")) > 0:
    for _ in range(int(os.environ["SOAK_STRESS"])):
        stress.append(subprocess.Popen([sys.executable, "-c", "while True:\n    sum(i * i for i in range(10000))"]))
if os.environ.get("SOAK_HOLD_GPU") == "1":
    sys.path.insert(0, ROOT)
    import torch
    from bess_amd import capi
    assert torch.cuda.is_available()
    print("holding a context:", capi.device_info(), flush=True)
lines = []
t0 = time.time()
import atexit
atexit.register(lambda: [q.kill() for q in stress])
for r in range(reps):
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bench_args, cwd=ROOT, env=env,
                             capture_output=True, text=True, timeout=float(os.environ.get("SOAK_REP_TIMEOUT_S", "120")))
    except subprocess.TimeoutExpired as e:
        with open(os.path.join(outdir, "hang_rep%d.txt" % r), "w") as f:
            f.write("==== stdout\n%s\n==== stderr\n%s\n" % (e.stdout, e.stderr))
        print("HANG at repetition %d after %.0f s; per-rank watchdog dumps in %s" % (r, time.time() - t0, env["BESSX_BENCH_ERRDIR"]), flush=True)
        for fn in sorted(os.listdir(env["BESSX_BENCH_ERRDIR"])) if os.path.isdir(env["BESSX_BENCH_ERRDIR"]) else []:
            txt = open(os.path.join(env["BESSX_BENCH_ERRDIR"], fn)).read()
            if "Thread" in txt or "Traceback" in txt:
                print("----", fn)
                print(txt[-3000:], flush=True)
        sys.exit(1)
    js = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or len(js) != 1:
        with open(os.path.join(outdir, "failure_rep%d.txt" % r), "w") as f:
            f.write("rc %d\n==== stdout\n%s\n==== stderr\n%s\n" % (out.returncode, out.stdout, out.stderr))
        print("FAILED at repetition %d (rc %d) after %.0f s; see %s" % (r, out.returncode, time.time() - t0, outdir), flush=True)
        print(out.stdout[-1500:])
        sys.exit(1)
    d = json.loads(js[0])
    lines.append({k: d.get(k) for k in ("selected_k", "cv_loss", "fits_per_step", "pdas_iterations_per_step", "value")})
    first = lines[0]
    for k in ("selected_k", "cv_loss", "fits_per_step", "pdas_iterations_per_step"):
        if k in first and first[k] is not None and lines[-1][k] != first[k]:
            print("repetition %d: %s = %r, first repetition had %r" % (r, k, lines[-1][k], first[k]), flush=True)
            sys.exit(1)
    if r % 5 == 4 or r == reps - 1:
        print("%d repetitions ok, %.0f s" % (r + 1, time.time() - t0), flush=True)
print("soak OK:", json.dumps(lines[0]))
