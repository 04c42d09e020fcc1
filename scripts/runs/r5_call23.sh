#!/bin/bash
# round 5, GPU call 23: k_merge_items with the lists of four (default) / one / eight entering items in flight
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c23; mkdir -p $OUT
cd $R
run() {  # lib workload docs
  local lib=$1 w=$2 docs=$3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib" | tee -a $OUT/ab.log
  DOCS=$docs timeout 600 python scripts/run_workload.py $w 10 2>&1 | tail -1 | grep -o "'k_merge_items[^)]*)\|'k_search_and'[^)]*)\|'k_search_term[^)]*)\|'k_or_lazy[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default merge1 merge8; do run $lib and3 10000000; run $lib term 10000000; done
done
for lib in default merge1 merge8; do run $lib or10 10000000; done
for lib in default merge1; do run $lib and3 100000000; done
unset RUCENE_GPU_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
