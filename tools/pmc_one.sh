#!/bin/bash
# One rocprofv3 counter pass (--pmc with --kernel-trace only) of a python command; prints per kernel the median counter
# values, the dispatch count and the median duration OF THE SAME RUN (a profiled pass is slower than a plain one: never
# divide a counter of this run by a duration of another); keeps nothing but that summary.
#   tools/pmc_one.sh NAME "COUNTERS" script.py [args...]        PMC_FILTER=substring keeps only matching kernels
name=$1; shift
ctr=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/pmc_$name
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $name -- python3 "$@" > $root/gpurun_out/pmc_${name}.log 2>&1
python3 - "$out" "$name" "${PMC_FILTER:-}" > $root/gpurun_out/pmc_${name}_summary.txt <<'PY'
import csv, glob, sys, statistics, collections
d, name, flt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60]
        if flt and flt not in k:
            continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            dur[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
if not dur:
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"][:60]
            if flt and flt not in k:
                continue
            dur[(k, r.get("Dispatch_Id", len(dur)))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
bykern = collections.defaultdict(list)
for (k, _), v in dur.items():
    bykern[k].append(v)
tot = collections.Counter()
for (k, c), v in acc.items():
    tot[k] += sum(v)
for k in sorted(bykern, key=lambda q: -sum(bykern[q]))[:8]:
    v = bykern[k]
    real = [x for x in v if x > 100000] or v
    print("%-60s dispatches=%d median_duration_ns=%.0f real_pass_median_ns=%.0f" % (k, len(v), statistics.median(v), statistics.median(real)))
    for (kk, c), vals in sorted(acc.items()):
        if kk == k:
            big = [x for x in vals if x > 0.25 * max(vals)] or vals  # (launches that fell through their gate count next to nothing)
            print("    %-30s n=%4d median=%.6g median_of_real_passes=%.6g" % (c, len(vals), statistics.median(vals), statistics.median(big)))
PY
cat $root/gpurun_out/pmc_${name}_summary.txt
