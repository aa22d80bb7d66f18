#!/bin/bash
# The rocprofv3 --kernel-trace --stats runs behind profiles/r05_*_kernel_stats.csv / *_kernel_real_pass.csv (one gpurun
# call; every profiler pass its own python process).   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles_r05b.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
B="$R/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1"
F=$R/tools/bench_family.py
echo "== configs[1] covariance (chunk chains)"; tools/prof_stats.sh r05_lm_config2_covariance $B --no-streaming-leg > /dev/null
echo "== configs[1] covariance, one chain"; BESSX_KPATH_CHAINS=1 tools/prof_stats.sh r05_lm_config2_covariance_single_chain $B --no-streaming-leg > /dev/null
echo "== configs[1] streaming (chunk chains)"; tools/prof_stats.sh r05_lm_config2_streaming $B --score-mode streaming > /dev/null
echo "== configs[1] streaming, one chain"; BESSX_KPATH_CHAINS=1 tools/prof_stats.sh r05_lm_config2_streaming_single_chain $B --score-mode streaming > /dev/null
echo "== configs[2] logistic";   tools/prof_stats.sh r05_logistic_config3 $F logistic > /dev/null
echo "== configs[3] lmcv";       BENCH_FAMILY_WARMUP=1 BENCH_FAMILY_TIMING=0 tools/prof_stats.sh r05_lmcv_config4 $F lmcv > /dev/null
echo "== poisson";               tools/prof_stats.sh r05_poisson_n100k_p5k $F poisson > /dev/null
echo "== configs[4] cox";        tools/prof_stats.sh r05_cox_config5_full_size $F cox > /dev/null
ls -la $R/gpurun_out | grep "r05_.*kernel_" | awk '{print $5, $9}'
