#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command on the GPU box; only the small CSV summaries come back
# (the trace database stays in /tmp: gpurun merges at most 64 MiB).   tools/prof_stats.sh NAME script.py [args...]
# Besides rocprofv3's own <NAME>_kernel_stats.csv: <NAME>_kernel_real_pass.csv -- per kernel the launches that did
# real work (longer than 100 us: a device-gated launch that falls through takes 2-5 us and drags the plain average down),
# so that an average launch duration can be recomputed from profiles/ alone.
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/prof_$name
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 "$@" > $root/gpurun_out/prof_${name}.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${name}_kernel_stats.csv
python3 - "$out" > $root/gpurun_out/${name}_kernel_real_pass.csv <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "RealPassCalls(>100us)", "RealPassTotalNs", "RealPassAverageNs",
            "Calls_0.1-1.05ms", "AverageNs_0.1-1.05ms", "Calls_>1.05ms", "AverageNs_>1.05ms"])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    real = [x for x in v if x > 100000]
    lo = [x for x in real if x <= 1050000]  # (k_cov_panel_dp on configs[1]: one 32-column group per launch ...
    hi = [x for x in real if x > 1050000]   #  ... two groups)
    w.writerow([k[:160], len(v), sum(v), "%.1f" % (sum(v) / len(v)), len(real), sum(real), "%.1f" % (sum(real) / len(real)) if real else "",
                len(lo), "%.1f" % (sum(lo) / len(lo)) if lo else "", len(hi), "%.1f" % (sum(hi) / len(hi)) if hi else ""])
PY
head -30 $root/gpurun_out/${name}_kernel_stats.csv | cut -c1-160
