#!/bin/bash
# round 5, GPU call 7: k_search_term visiting blocks best bound first + per-chunk threshold exchange (A/B against the round-4 kernel),
# the whole GPU suite, cold-path host timing, what the box gives the CPU leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c7; mkdir -p $OUT
cd $R
echo "cpu: nproc=$(nproc) affinity=$(python -c 'import os; print(len(os.sched_getaffinity(0)))') cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) cfs=$(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)" | tee -a $OUT/ab.log
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)\|'k_phrase_match_lanes[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default and_old; do run $lib term 10000000 X=1; done
done
for lib in default and_old; do run $lib term 100000000 X=1; done
unset RUCENE_GPU_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
echo "== cold 100M host timing" | tee -a $OUT/ab.log
RGPU_HOST_TIMING=1 DOCS=100000000 timeout 600 python scripts/run_workload.py cold 3 2>&1 | grep -i "prepare host" | cut -c1-600 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py --configs none > $OUT/bench_head.json 2> $OUT/bench_head.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench_head.json 2>&1 | cut -c1-1800 | tee -a $OUT/show.log | head -12
