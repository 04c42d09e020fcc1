#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_session_or.sh <tag> [lib under build_variants ...]'
# OR parity tests on the default library, then the 10-term OR batch on it and on each variant library.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-or}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "disjunction or wide or should or fullsize or min_should" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -3 $OUT/pytest.log | tee -a $OUT/session.log
echo "== default" | tee -a $OUT/session.log
timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -2 | tee -a $OUT/session.log
for lib in "$@"; do
  echo "== $lib" | tee -a $OUT/session.log
  RUCENE_GPU_LIB=$R/build_variants/$lib timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -2 | tee -a $OUT/session.log
done
