#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_session_abl.sh <tag> <workload> <lib...>'   (libs under build_variants/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-abl}; W=${2:-or10}; shift 2
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== default" | tee $OUT/session.log
timeout 300 python scripts/run_workload.py $W 3 2>&1 | tail -2 | tee -a $OUT/session.log
for lib in "$@"; do
  echo "== $lib" | tee -a $OUT/session.log
  RUCENE_GPU_LIB=$R/build_variants/$lib timeout 300 python scripts/run_workload.py $W 3 2>&1 | tail -2 | tee -a $OUT/session.log
done
