#!/bin/bash
# The final library of round 5 (k_cov_panel_dp as the fills' kernel): counter passes of the panel kernel alone, then the
# kernel-trace summaries of every bench workload (tools/collect_profiles_r05b.sh).  One gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles_r05c.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
P="$R/tools/panel_bench.py 50000 10000 12"
mkdir -p $R/gpurun_out
python3 $P > $R/gpurun_out/r05_paneldp_bench.jsonl 2>&1 || { tail -5 $R/gpurun_out/r05_paneldp_bench.jsonl; exit 1; }
cat $R/gpurun_out/r05_paneldp_bench.jsonl
tools/pmc_one.sh r05_paneldp_clock "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" $P > /dev/null
tools/pmc_one.sh r05_paneldp_waits "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" $P > /dev/null
tools/pmc_one.sh r05_paneldp_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_LEVEL_LDS" $P > /dev/null
tools/pmc_one.sh r05_paneldp_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU" $P > /dev/null
tools/pmc_one.sh r05_paneldp_fetch "FETCH_SIZE" $P > /dev/null
tools/pmc_one.sh r05_paneldp_write "WRITE_SIZE" $P > /dev/null
ls -la $R/gpurun_out | grep r05_paneldp | awk '{print $5, $9}'
bash $R/tools/collect_profiles_r05b.sh
