#!/bin/bash
# k_search_term against blocks per item (rgpu_config.blocks_per_item; 0 = the library's choice), on ONE box.
# usage (GPU box): bash scripts/bpi_sweep.sh <tag> "<docs> ..." "<bpi> ..."
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-bpi}; DOCS_LIST=${2:-10000000}; BPIS=${3:-"0 128 256 512 1024 2048 4096"}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for docs in $DOCS_LIST; do
  for bpi in $BPIS; do
    echo "== term docs=$docs bpi=$bpi" | tee -a $OUT/bpi.log
    BPI=$bpi DOCS=$docs timeout 600 python scripts/run_workload.py term ${REPS:-10} 2>&1 | tail -1 | grep -o "'k_search_term': ([0-9]*, [0-9.]*)\|'k_merge_items': ([0-9]*, [0-9.]*)" | tee -a $OUT/bpi.log
  done
done
