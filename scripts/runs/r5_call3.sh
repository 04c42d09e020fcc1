#!/bin/bash
# round 5, GPU call 3: auto-sized AND items, next-group row prefetch, what the two kinds of candidate vectors cost (ablations)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c3; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_and" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_merge_items[^)]*)\|'k_phrase_match_lanes[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default and_pf and_t20k and_t40k and_pf_g2; do run $lib and3 10000000 X=1; done
done
for lib in and_abl4 and_abl5 and_abl6; do run $lib and3 10000000 X=1; done
run and_old and3 10000000 X=1
for lib in default and_pf and_t40k and_abl4 and_abl5 and_abl6; do run $lib and3 100000000 X=1; done
run default phrase2 10000000 X=1
run and_pf phrase2 10000000 X=1
unset RUCENE_GPU_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conjunction or lazy or must or phrase or filter or shard or record or sloppy" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
