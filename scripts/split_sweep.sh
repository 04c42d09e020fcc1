#!/bin/bash
# k_search_term against RGPU_TERM_SPLIT (items per query of a few chunks) and RGPU_TERM_TARGET_ITEMS, on ONE box.
# usage (GPU box): bash scripts/split_sweep.sh <tag> "<docs> ..." "<split> ..." "<target items> ..."
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-split}; DOCS_LIST=${2:-10000000}; SPLITS=${3:-"1 2 4 8"}; TARGETS=${4:-12000}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for docs in $DOCS_LIST; do
  for rep in 1 2; do
  for tg in $TARGETS; do
  for sp in $SPLITS; do
    echo "== term docs=$docs split=$sp target=$tg rep=$rep" | tee -a $OUT/split.log
    RGPU_TERM_TARGET_ITEMS=$tg RGPU_TERM_SPLIT=$sp DOCS=$docs timeout 600 python scripts/run_workload.py term ${REPS:-10} 2>&1 | tail -1 | grep -o "'k_search_term': ([0-9]*, [0-9.]*)\|'k_merge_items': ([0-9]*, [0-9.]*)" | tee -a $OUT/split.log
  done
  done
  done
done
