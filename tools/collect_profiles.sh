#!/bin/bash
# Collects the rocprofv3 runs behind profiles/r02_* on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'
# then, back in the build container:  python tools/summarize_profiles.py ... (see profiles/README.md).
# Every profiler pass is its own process; --pmc passes carry --kernel-trace only (never the sys / hip / hsa traces).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
F="python3 $R/tools/bench_family.py"
run() { name=$1; shift; echo "== $name"; "$@" > $O/$name.log 2>&1 || echo "   (rc $?)"; }
run stats_cov   rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cov   -- $B --steps 2 --warmup 1 --no-streaming-leg
run stats_strm  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_strm  -- $B --steps 2 --warmup 1 --score-mode streaming
run stats_logit rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_logit -- $F logistic
run stats_cox   rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cox   -- $F cox 100000 10000 60
run stats_lmcv  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lmcv  -- $F lmcv
S="--steps 1 --warmup 0 --no-streaming-leg --kmax 60"
run pmc_fetch_v3 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_v3 -- $B $S
run pmc_write_v3 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_v3 -- $B $S
run pmc_mfma_v3  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_v3 -- $B $S
run pmc_fetch_xtv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_xtv -- $B --steps 1 --warmup 0 --score-mode streaming --kmax 20
run pmc_write_xtv rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_xtv -- $B --steps 1 --warmup 0 --score-mode streaming --kmax 20
export BESSX_PANEL_VARIANT=4
run stats_cov_v4 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cov_v4 -- $B --steps 2 --warmup 1 --no-streaming-leg
run pmc_fetch_v4 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_v4 -- $B $S
run pmc_write_v4 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_v4 -- $B $S
run pmc_mfma_v4  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_v4 -- $B $S
unset BESSX_PANEL_VARIANT
# keep what the summaries need, drop the bulky traces
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
