#!/bin/bash
# One rocprofv3 counter pass (--pmc with --kernel-trace only) of a python command; prints per kernel the median counter
# value and dispatch count, keeps nothing but that summary.   tools/pmc_one.sh NAME "COUNTERS" script.py [args...]
name=$1; shift
ctr=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/pmc_$name
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $name -- python3 "$@" > $root/gpurun_out/pmc_${name}.log 2>&1
python3 - "$out" "$name" > $root/gpurun_out/pmc_${name}_summary.txt <<'PY'
import csv, glob, sys, statistics, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-70s %-28s n=%5d median=%.6g sum=%.6g" % (k, c, len(v), statistics.median(v), sum(v)))
PY
cat $root/gpurun_out/pmc_${name}_summary.txt
