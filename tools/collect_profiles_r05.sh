#!/bin/bash
# Counter passes of the panel kernels alone (tools/panel_bench.py) for profiles/r05_panel_*: every --pmc pass is its own
# python process with --kernel-trace only (the pool refuses --pmc with other trace domains).  Only summaries come back.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles_r05.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
P="$R/tools/panel_bench.py 50000 10000 12"
mkdir -p $R/gpurun_out
python3 $P --check > $R/gpurun_out/r05_panel_bench.jsonl 2>&1 || { tail -5 $R/gpurun_out/r05_panel_bench.jsonl; exit 1; }
cat $R/gpurun_out/r05_panel_bench.jsonl
# (1) clocks and occupancy  (2) where the waves' cycles go  (3) LDS  (4) matrix cores  (5) / (6) HBM traffic
tools/pmc_one.sh r05_panel_clock "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" $P > /dev/null
tools/pmc_one.sh r05_panel_waits "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" $P > /dev/null
tools/pmc_one.sh r05_panel_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" $P > /dev/null
tools/pmc_one.sh r05_panel_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU" $P > /dev/null
tools/pmc_one.sh r05_panel_fetch "FETCH_SIZE" $P > /dev/null
tools/pmc_one.sh r05_panel_write "WRITE_SIZE" $P > /dev/null
tools/pmc_one.sh r05_panel_l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" $P > /dev/null
ls -la $R/gpurun_out | grep r05_panel | awk '{print $5, $9}'
