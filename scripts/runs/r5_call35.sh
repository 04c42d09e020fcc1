#!/bin/bash
# round 5, GPU call 35: block-max sketches (k_term_sketch + a starting threshold in k_search_term) against RGPU_TERM_SKETCH=0: time,
# blocks unpacked, the whole GPU suite with sketches on
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c35; mkdir -p $OUT
cd $R
for docs in 10000000 100000000; do
  for sk in 0 1; do
    echo "== term docs=$docs RGPU_TERM_SKETCH=$sk" | tee -a $OUT/ab.log
    RGPU_TERM_SKETCH=$sk DOCS=$docs timeout 600 python scripts/run_workload.py term 20 2>&1 | grep "last launch\|k_search_term" | sed "s/'k_skip_dir.*'k_prepare_norms': ([0-9]*, [0-9.]*), //" | cut -c1-400 | tee -a $OUT/ab.log
  done
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
