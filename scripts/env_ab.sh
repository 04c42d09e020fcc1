#!/bin/bash
# A/B of an environment knob on ONE box: every workload under every value.
# usage (GPU box): bash scripts/env_ab.sh <tag> <VAR> "<value> ..." "<workload[@docs]> ..."
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-envab}; VAR=${2:?variable}; VALS=${3:?values}; WORK=${4:-term}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in $WORK; do
  kind=${w%@*}; docs=10000000
  [ "$w" != "$kind" ] && docs=${w#*@}
  for rep in 1 2; do
    for v in $VALS; do
      echo "== $kind docs=$docs $VAR=$v rep=$rep" | tee -a $OUT/ab.log
      env $VAR=$v DOCS=$docs timeout 600 python scripts/run_workload.py $kind ${REPS:-10} 2>&1 | tail -3 | cut -c1-1500 | tee -a $OUT/ab.log
    done
  done
done
