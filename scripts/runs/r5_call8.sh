#!/bin/bash
# round 5, GPU call 8: how often k_search_term's wavefronts should exchange thresholds (never / every chunk / chunks 1, 2, 4, 8),
# cold-path host timing with the bulk-adopted term table, the prepared-term budget test
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c8; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)\|'k_phrase_match_lanes[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in term_x0 term_x1 term_x2 and_old; do run $lib term 10000000 X=1; done
done
for lib in term_x0 term_x1 term_x2 and_old; do run $lib term 100000000 X=1; done
unset RUCENE_GPU_LIB
echo "== cold 100M host timing" | tee -a $OUT/ab.log
RGPU_HOST_TIMING=1 DOCS=100000000 timeout 600 python scripts/run_workload.py cold 3 2>&1 | grep -i "prepare host" | cut -c1-600 | tee -a $OUT/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "budget or corrupt or error_codes or decode or docs_only or multi_leaf" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
