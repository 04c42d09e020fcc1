#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s7}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in or_abl1.so or_abl2.so or_abl3.so; do
for cfg in "ORW=1024 ORD=-1" "ORW=1024 ORD=0"; do
  echo "== or10 $v $cfg" | tee -a $OUT/session.log
  env $cfg RUCENE_GPU_LIB=$R/build_variants/$v timeout 300 python scripts/run_workload.py or10 2 2>&1 | tail -1 | tee -a $OUT/session.log
done
done
