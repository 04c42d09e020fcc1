#!/bin/bash
# round 5, GPU call 24: what k_search_term / k_search_and cost when every item starts from its query's FINAL threshold (variant
# build keep_tau: the shared thresholds are not zeroed between launches of the same batch) — the ceiling of a threshold pre-pass
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c24; mkdir -p $OUT
cd $R
run() {  # lib workload docs
  local lib=$1 w=$2 docs=$3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib" | tee -a $OUT/ab.log
  DOCS=$docs timeout 600 python scripts/run_workload.py $w 20 2>&1 | tail -1 | grep -o "'k_merge_items[^)]*)\|'k_search_and'[^)]*)\|'k_search_term[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for lib in default keep_tau; do run $lib term 10000000; run $lib and3 10000000; done
for lib in default keep_tau; do run $lib term 100000000; run $lib and3 100000000; done
