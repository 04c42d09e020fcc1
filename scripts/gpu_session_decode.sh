#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_session_decode.sh <tag> [lib under build_variants ...]'
# block-decode kernel at 10 M and 100 M docs on the default library and on each variant; decode parity tests on each variant
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-dec}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for lib in "" "$@"; do
  echo "== ${lib:-default}" | tee -a $OUT/session.log
  ( [ -n "$lib" ] && export RUCENE_GPU_LIB=$R/build_variants/$lib
    [ -n "$lib" ] && timeout 300 python -m pytest tests -m gpu -x -q -k "decode" 2>&1 | tail -1
    timeout 300 python scripts/run_workload.py decode 5 2>&1 | tail -1
    DOCS=100000000 timeout 300 python scripts/run_workload.py decode 5 2>&1 | tail -1 ) | tee -a $OUT/session.log
done
