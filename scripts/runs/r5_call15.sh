#!/bin/bash
# round 5, GPU call 15: k_search_term with the published key's score half only (default) against the whole key (term_seen64), same box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c15; mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  for lib in default term_seen64; do
    for docs in 10000000 100000000; do
      [ $docs = 100000000 ] && [ $rep = 3 ] && continue
      echo "== term docs=$docs lib=$lib" | tee -a $OUT/ab.log
      if [ $lib = default ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
      DOCS=$docs timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_search_term[^)]*)" | tee -a $OUT/ab.log
    done
  done
done
unset RUCENE_GPU_LIB
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "single_term" 2>&1 | tail -1 | tee -a $OUT/ab.log
RUCENE_GPU_LIB=$R/build_variants/term_seen64.so timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "single_term" 2>&1 | tail -1 | tee -a $OUT/ab.log
