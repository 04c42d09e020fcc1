#!/bin/bash
# round 5, GPU call 27: blocks k_search_term unpacks with / without the head-first wait and with final thresholds (keep_tau)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c27; mkdir -p $OUT
cd $R
for lib in term_wait0 default term_wait_long keep_tau; do
  for bpi in 0 128; do
    if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
    echo "== term lib=$lib BPI=$bpi" | tee -a $OUT/ab.log
    BPI=$bpi DOCS=10000000 timeout 600 python scripts/run_workload.py term 10 2>&1 | grep "last launch\|k_search_term" | cut -c1-500 | tee -a $OUT/ab.log
  done
done
