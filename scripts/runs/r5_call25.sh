#!/bin/bash
# round 5, GPU call 25: k_search_term with the query's head item publishing after its first 64 blocks and the other items waiting
# (bounded) for that key — against the build without the wait, two sleep lengths, parity tests on the default
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c25; mkdir -p $OUT
cd $R
run() {  # lib workload docs
  local lib=$1 w=$2 docs=$3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib" | tee -a $OUT/ab.log
  DOCS=$docs timeout 600 python scripts/run_workload.py $w 20 2>&1 | tail -1 | grep -o "'k_merge_items[^)]*)\|'k_search_term[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default term_wait0 term_wait_s4 term_wait_s64; do run $lib term 10000000; done
done
for lib in default term_wait0 term_wait_s4 term_wait_s64; do run $lib term 100000000; done
unset RUCENE_GPU_LIB
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "single_term or term or k_above or mixed or knobs or tie_heavy or enqueue or sharded or native_planner or counters or multi_leaf or live_docs" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
