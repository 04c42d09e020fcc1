#!/bin/bash
# round 5, GPU call 26: with the head-first wait, shorter TERM items may pay (an item no longer warms up on its own): blocks per
# item x wait length (none / 48 polls ~ 20 us / 160 polls ~ 70 us)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c26; mkdir -p $OUT
cd $R
run() {  # lib docs bpi
  local lib=$1 docs=$2 bpi=$3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== term docs=$docs lib=$lib BPI=$bpi" | tee -a $OUT/ab.log
  BPI=$bpi DOCS=$docs timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_merge_items[^)]*)\|'k_search_term[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for bpi in 0 256 128 64; do
  for lib in term_wait0 default term_wait_long; do run $lib 10000000 $bpi; done
done
for bpi in 0 512 256; do
  for lib in term_wait0 term_wait_long; do run $lib 100000000 $bpi; done
done
