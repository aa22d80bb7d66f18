#!/bin/bash
# Round 6, final library: rocprofv3 kernel-trace summaries of the other configs with shared passes (logistic, Poisson,
# Cox at full size, the streaming LM path): which kernels the paths consist of now.
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles_r06b.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
tools/prof_stats.sh r06_logistic_config3 $R/tools/bench_family.py logistic > /dev/null
tools/prof_stats.sh r06_poisson_n100k_p5k $R/tools/bench_family.py poisson > /dev/null
tools/prof_stats.sh r06_lm_config2_streaming_shared $R/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --score-mode streaming > /dev/null
tools/prof_stats.sh r06_cox_config5_full_size $R/tools/bench_family.py cox > /dev/null
ls -la $R/gpurun_out | grep "r06_.*kernel" | awk '{print $5, $9}'
