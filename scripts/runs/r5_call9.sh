#!/bin/bash
# round 5, GPU call 9: the whole GPU suite on the current tree, TERM with the few-items exchange rule, k_or_lazy's phase breakdown, bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c9; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks\|k_or_lazy" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)\|'k_or_lazy[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default term_x0 and_old; do run $lib term 10000000 X=1; done
done
for lib in default term_x0 and_old; do run $lib term 100000000 X=1; done
unset RUCENE_GPU_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
echo "== or10 phase breakdown (RGPU_LZ_TIME)" | tee -a $OUT/ab.log
RUCENE_GPU_LIB=$R/build_variants/lz_time.so timeout 600 python scripts/run_workload.py or10 4 2>&1 | grep -i "lz\|dbg\|k_or_lazy" | cut -c1-900 | tail -8 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1500 > $OUT/show.log; head -4 $OUT/show.log
