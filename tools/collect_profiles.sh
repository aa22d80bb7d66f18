#!/bin/bash
# Collects the rocprofv3 runs behind profiles/r03_* on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'
# then, back in the build container:  python tools/summarize_profiles.py ... (see profiles/README.md).
# Every profiler pass is its own process; --pmc passes carry --kernel-trace only (never the sys / hip / hsa traces).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs"
F="python3 $R/tools/bench_family.py"
run() { name=$1; shift; echo "== $name"; "$@" > $O/$name.log 2>&1 || echo "   (rc $?)"; }
run stats_cov   rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cov   -- $B --steps 2 --warmup 1 --no-streaming-leg
run stats_strm  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_strm  -- $B --steps 2 --warmup 1 --score-mode streaming
run stats_logit rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_logit -- $F logistic
run stats_cox   rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cox   -- $F cox
run stats_lmcv  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lmcv  -- $F lmcv
S="--steps 1 --warmup 0 --no-streaming-leg --kmax 60"
run pmc_fetch_cov rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_cov -- $B $S
run pmc_write_cov rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_cov -- $B $S
run pmc_mfma_cov  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_cov -- $B $S
run pmc_fetch_xtv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_xtv -- $B --steps 1 --warmup 0 --score-mode streaming --kmax 20
run pmc_write_xtv rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_xtv -- $B --steps 1 --warmup 0 --score-mode streaming --kmax 20
# the families: the IRLS Gram kernel at k <= 100 and the Cox kernels at full size, k <= 30 (a shorter path: the counters
# are per launch)
run pmc_fetch_logit rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_logit -- $F logistic
run pmc_mfma_logit  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_logit -- $F logistic
run pmc_fetch_cox rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_cox -- $F cox 200000 20000 30
run pmc_write_cox rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_cox -- $F cox 200000 20000 30
run pmc_mfma_cox  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_cox -- $F cox 200000 20000 150
# gpurun copies back at most 64 MiB: summarise on the box, drop the raw traces
python3 $R/tools/summarize_r03.py $O
du -sh $O
