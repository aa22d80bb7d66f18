#!/bin/bash
# Round 6, the FINAL library: kernel-trace summaries of the bench's headline command (what roofline.avg_launch_ms must
# agree with), of the streaming leg as shared passes, and the separate counter passes (FETCH_SIZE / WRITE_SIZE, one
# counter per run, --kernel-trace only) of the kernels that stream X.  One gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles_r06.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
B="$R/bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline"
# 1. the headline command under the profiler (kernel trace + stats): k_cov_panel_dp by launch width, k_xtv_mc of the streaming leg
tools/prof_stats.sh r06_lm_config2_bench $B > /dev/null
# 2. counter passes of the same command: the panel kernel (headline) and the shared streaming pass
PMC_FILTER=k_cov_panel tools/pmc_one.sh r06_panel_fetch "FETCH_SIZE" $B --no-streaming-leg > /dev/null
PMC_FILTER=k_cov_panel tools/pmc_one.sh r06_panel_write "WRITE_SIZE" $B --no-streaming-leg > /dev/null
PMC_FILTER=k_xtv tools/pmc_one.sh r06_xtvmc_fetch "FETCH_SIZE" $R/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --score-mode streaming > /dev/null
PMC_FILTER=k_xtv tools/pmc_one.sh r06_xtvmc_write "WRITE_SIZE" $R/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --score-mode streaming > /dev/null
# 3. the multi-chain Cox score kernel alone at full size (4 chains per launch), and the MFMA counters of the panel kernel
PMC_FILTER=k_cox_score1p tools/pmc_one.sh r06_coxmc_fetch "FETCH_SIZE" $R/tools/cox_score_bench.py 200000 20000 5 1 14 > /dev/null
PMC_FILTER=k_cov_panel tools/pmc_one.sh r06_panel_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" $B --no-streaming-leg > /dev/null
ls -la $R/gpurun_out | grep "r06_" | awk '{print $5, $9}'
