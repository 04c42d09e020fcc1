#!/bin/bash
# A/B of build variants on ONE box (kernel times differ by 5-10 % from box to box): every workload through every library.
# usage (GPU box): bash scripts/ab.sh <tag> "<lib> <lib> ..." "<workload[@docs]> ..."    lib = "default" or build_variants/<name>.so
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ab}; LIBS=${2:-default}; WORK=${3:-decode}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in $WORK; do
  kind=${w%@*}; docs=10000000
  [ "$w" != "$kind" ] && docs=${w#*@}
  for rep in 1 2; do
    for lib in $LIBS; do
      if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/$lib; fi
      echo "== $kind docs=$docs lib=$lib rep=$rep" | tee -a $OUT/ab.log
      DOCS=$docs timeout 600 python scripts/run_workload.py $kind ${REPS:-8} 2>&1 | tail -3 | cut -c1-1500 | tee -a $OUT/ab.log
    done
  done
done
