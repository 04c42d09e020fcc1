#!/bin/bash
# k_search_term against the smallest item of a split query (RGPU_TERM_MIN_ITEM_BLOCKS) and the sketch floor (build variants), on ONE box.
# usage (GPU box): bash scripts/minitem_sweep.sh <tag> "<docs> ..." "<lib> ..." "<min item blocks> ..."
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-minitem}; DOCS_LIST=${2:-10000000}; LIBS=${3:-default}; MINS=${4:-"64 32 16"}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for docs in $DOCS_LIST; do
  for rep in 1 2; do
  for lib in $LIBS; do
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/$lib; fi
  for mi in $MINS; do
    echo "== term docs=$docs lib=$lib min_item=$mi rep=$rep" | tee -a $OUT/sweep.log
    RGPU_TERM_MIN_ITEM_BLOCKS=$mi DOCS=$docs timeout 600 python scripts/run_workload.py term ${REPS:-10} 2>&1 | tail -2 | grep -o "blocks_decoded.: [0-9]*\|'k_search_term': ([0-9]*, [0-9.]*)\|'k_merge_items': ([0-9]*, [0-9.]*)\|'k_term_sketch': ([0-9]*, [0-9.]*)" | tee -a $OUT/sweep.log
  done
  done
  done
done
