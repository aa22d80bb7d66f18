#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command on the GPU box; only the small CSV summaries come back
# (the trace database stays in /tmp: gpurun merges at most 64 MiB).   tools/prof_stats.sh NAME script.py [args...]
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/prof_$name
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 "$@" > $root/gpurun_out/prof_${name}.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${name}_kernel_stats.csv
head -30 $root/gpurun_out/${name}_kernel_stats.csv | cut -c1-160
