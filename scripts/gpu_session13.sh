#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s13}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in "" and_w4.so and_w5.so and_w8.so; do
  for abpi in 0 4; do
  echo "== and3 lib=${w:-default(w6)} ABPI=$abpi" | tee -a $OUT/session.log
  ( [ -n "$w" ] && export RUCENE_GPU_LIB=$R/build_variants/$w; ABPI=$abpi timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -1 ) | tee -a $OUT/session.log
  done
done
echo "== or10 default" | tee -a $OUT/session.log
timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -1 | tee -a $OUT/session.log
