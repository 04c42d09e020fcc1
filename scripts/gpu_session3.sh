#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s3}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -3 $OUT/pytest.log | tee -a $OUT/session.log
for w in "" and_w8.so; do
  echo "== and3 lib=${w:-default}" | tee -a $OUT/session.log
  ( [ -n "$w" ] && export RUCENE_GPU_LIB=$R/build_variants/$w; timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -1 ) | tee -a $OUT/session.log
done
for bpi in 0 64 256 512 1024; do
  echo "== term BPI=$bpi" | tee -a $OUT/session.log
  BPI=$bpi timeout 300 python scripts/run_workload.py term 10 2>&1 | tail -1 | tee -a $OUT/session.log
done
