#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s5}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -15 $OUT/pytest.log | tee -a $OUT/session.log
for cfg in "ORW=1024 ORD=0" "ORW=768 ORD=0" "ORW=1536 ORD=0" "ORW=2048 ORD=0" "ORW=1024 ORD=-1" "ORW=1024 ORD=1"; do
  echo "== or10 $cfg" | tee -a $OUT/session.log
  env $cfg timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -1 | tee -a $OUT/session.log
done
