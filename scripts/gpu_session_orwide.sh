#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_session_orwide.sh <tag> [variant libs...]'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-orw}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== pytest" | tee $OUT/session.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log
tail -15 $OUT/pytest.log | tee -a $OUT/session.log
for ww in ${ORWWS:-0}; do
  echo "== or10 wide WS=$ww" | tee -a $OUT/session.log
  ORWW=$ww timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -1 | tee -a $OUT/session.log
done
for lib in "$@"; do
  echo "== $lib" | tee -a $OUT/session.log
  RUCENE_GPU_LIB=$R/build_variants/$lib timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -1 | tee -a $OUT/session.log
done
