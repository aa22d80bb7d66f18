#!/bin/bash
# The rocprofv3 runs behind profiles/r04_* (one gpurun call; every profiler pass its own python process, --pmc passes with
# --kernel-trace only).  Only the small CSV / text summaries come back.
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles_r04.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
export GPU_MAX_HW_QUEUES=8   # (the profiler may start the HIP runtime before bench.py can ask for it)
B="$R/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1"
F=$R/tools/bench_family.py
echo "== configs[1] covariance (chunk chains)"; tools/prof_stats.sh r04_lm_config2_covariance $B --no-streaming-leg > /dev/null
echo "== configs[1] covariance, one chain"; BESSX_KPATH_CHAINS=1 tools/prof_stats.sh r04_lm_config2_covariance_single_chain $B --no-streaming-leg > /dev/null
echo "== configs[1] streaming";  tools/prof_stats.sh r04_lm_config2_streaming $B --score-mode streaming > /dev/null
echo "== configs[2] logistic";   tools/prof_stats.sh r04_logistic_config3 $F logistic > /dev/null
echo "== configs[3] lmcv";       BENCH_FAMILY_WARMUP=1 BENCH_FAMILY_TIMING=0 tools/prof_stats.sh r04_lmcv_config4 $F lmcv > /dev/null
echo "== poisson";               tools/prof_stats.sh r04_poisson_n100k_p5k $F poisson > /dev/null
echo "== grouped";               tools/prof_stats.sh r04_grouped_lm $F grouped > /dev/null
echo "== default sequence";      tools/prof_stats.sh r04_default_sequence $R/tools/default_sequence.py > /dev/null
echo "== configs[4] cox";        tools/prof_stats.sh r04_cox_config5_full_size $F cox > /dev/null
echo "== pmc cox score";         tools/pmc_one.sh r04_cox_score_fetch FETCH_SIZE $R/tools/cox_score_bench.py 200000 20000 3 1 0 > /dev/null
tools/pmc_one.sh r04_cox_score_write WRITE_SIZE $R/tools/cox_score_bench.py 200000 20000 3 1 0 > /dev/null
ls -la $R/gpurun_out | grep r04_ | awk '{print $5, $9}'
