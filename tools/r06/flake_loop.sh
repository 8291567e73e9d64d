#!/bin/bash
# Round 6, review item 2: the FULL GPU suite, exactly as the driver runs it (pytest tests/ -x -q -m gpu), over and over,
# with the retry of tests/test_gpu_rollout.py deleted.  LANES suites run side by side on the one GPU (more timing stress
# on the forked rollout workers than a suite alone), each lane repeats until BUDGET_S seconds have passed.
# Usage (GPU box): bash tools/r06/flake_loop.sh [LANES=2] [BUDGET_S=3600]
LANES=${1:-2}; BUDGET_S=${2:-3600}
OUT=gpurun_out/flake; mkdir -p $OUT
T0=$(date +%s)
lane() {
  local L=$1 i=0
  while [ $(( $(date +%s) - T0 )) -lt $BUDGET_S ]; do
    i=$((i+1))
    local log=$OUT/lane${L}_run${i}.log
    local t1=$(date +%s)
    UPAMD_TEST_STATS=$OUT/serving_stats.jsonl UPAMD_ABORT_PROBE=$OUT/abort_lane${L}_run${i}.txt timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $log 2>&1
    local rc=$?
    echo "lane $L run $i rc=$rc $(( $(date +%s) - t1 ))s $(tail -1 $log)" >> $OUT/summary.txt
    if [ $rc -eq 0 ]; then tail -3 $log > $log.tail; rm -f $log; fi      # keep full logs of failures only
  done
}
for L in $(seq 1 $LANES); do lane $L & done
wait
cat $OUT/summary.txt
