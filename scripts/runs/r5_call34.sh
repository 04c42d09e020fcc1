#!/bin/bash
# round 5, GPU call 34: k_term_floor — a starting threshold per query from the frontier words (the k-th largest "largest-freq posting"
# score over the term's blocks) in front of k_search_term: time, blocks unpacked, parity
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c34; mkdir -p $OUT
cd $R
for docs in 10000000 100000000; do
  for lib in default term_floor; do
    if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
    echo "== term docs=$docs lib=$lib" | tee -a $OUT/ab.log
    DOCS=$docs timeout 600 python scripts/run_workload.py term 20 2>&1 | grep "last launch\|k_search_term" | sed "s/'k_skip_dir.*'k_prepare_norms': ([0-9]*, [0-9.]*), //" | cut -c1-400 | tee -a $OUT/ab.log
  done
done
export RUCENE_GPU_LIB=$R/build_variants/term_floor.so
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "single_term or term or k_above or mixed or knobs or tie_heavy or enqueue or sharded or native_planner or counters or multi_leaf or live_docs or negative_boost or docs_only" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
