#!/bin/bash
# round 5, GPU call 20: k_search_term's item size (blocks per item) after the best-bound-first change: auto (512 at both sizes) vs fixed
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c20; mkdir -p $OUT
cd $R
for rep in 1 2; do
  for bpi in 0 64 128 256 1024; do
    echo "== term docs=10000000 BPI=$bpi" | tee -a $OUT/ab.log
    BPI=$bpi DOCS=10000000 timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_search_term[^)]*)\|'k_merge_items[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
  done
done
for bpi in 0 256 1024 2048; do
  echo "== term docs=100000000 BPI=$bpi" | tee -a $OUT/ab.log
  BPI=$bpi DOCS=100000000 timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_search_term[^)]*)\|'k_merge_items[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
done
