#!/bin/bash
# The measurements behind DESIGN.md 3b's round-5 paragraphs on staged fills, the coarse chain beside the chunks and the
# number of chains (one gpurun call):   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_chain_timelines.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
run_bench() {  # name, hooks, chain counts...
  local name=$1 hooks=$2; shift 2
  BESSX_TEST_HOOKS=$hooks timeout -k 10 300 python3 $R/tools/kchunks_bench.py "$@" > $O/r05_kchunks_$name.jsonl 2>&1 || exit 1
  tail -1 $O/r05_kchunks_$name.jsonl | cut -c1-120
}
run_bench staged_fills "" 4 &&
run_bench rendezvous kchunks_staged=0 4 &&
run_bench staged_own_stream kchunks_reserve=0 4 &&
run_bench pipeline kchunks_pipeline=1 3 4 5 &&
run_bench five_to_eight_chains "" 5 6 8 || exit 1
for c in 4 5 8; do
  BESSX_TEST_HOOKS=kchunks_log=1 timeout -k 10 200 python3 $R/tools/pipe_log_run.py $c > $O/log.tmp 2>&1 || exit 1
  awk '/==== path 3/{f=1} f' $O/log.tmp > $O/r05_chain_log_${c}_chains.txt
done
BESSX_TEST_HOOKS=kchunks_log=1,kchunks_pipeline=1 timeout -k 10 200 python3 $R/tools/pipe_log_run.py 4 > $O/log.tmp 2>&1 || exit 1
awk '/==== path 3/{f=1} f' $O/log.tmp > $O/r05_chain_log_pipeline_4_chunks.txt
rm -f $O/log.tmp
timeout -k 10 300 $R/tools/pipe_trace.sh 4 > /dev/null 2>&1 && cp $O/pipe_trace_summary.txt $O/r05_chain_timeline_staged_fills.txt &&
BESSX_TEST_HOOKS=kchunks_pipeline=1 timeout -k 10 300 $R/tools/pipe_trace.sh 4 > /dev/null 2>&1 && cp $O/pipe_trace_summary.txt $O/r05_chain_timeline_pipeline.txt
ls -la $O | grep "r05_chain\|r05_kchunks_" | awk '{print $5, $9}'
