#!/bin/bash
# round 5, GPU call 6: the whole GPU suite on the new default (membership-bit probe, XCD chunks), probe A/B, first full bench.py of the round
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c6; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)\|'k_phrase_match_lanes[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default and_probe1 and_probe2 and_noxcd; do run $lib and3 10000000 X=1; done
done
for lib in default and_probe1 and_probe2; do run $lib and3 100000000 X=1; done
run default phrase2 10000000 X=1
unset RUCENE_GPU_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -5 $OUT/bench.err | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1500 | tee -a $OUT/show.log | head -60
