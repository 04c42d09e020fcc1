#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_session_and.sh <tag> [lib under build_variants ...]'
# conjunction / phrase / rescore parity tests on the default library, then the 3-term AND batch on it and on each variant
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-and}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "conjunction or must or filter or phrase or rescor or smoke or docs_only or live or fullsize" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -3 $OUT/pytest.log | tee -a $OUT/session.log
for lib in "" "$@"; do
  echo "== ${lib:-default}" | tee -a $OUT/session.log
  ( [ -n "$lib" ] && export RUCENE_GPU_LIB=$R/build_variants/$lib; timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -2 ) | tee -a $OUT/session.log
done
