#!/bin/bash
# round 5, GPU call 18: the whole GPU suite on the tree as it stands (bulk plan by host threads with the compact term array,
# k_search_term's non-headline instantiations at seven wavefronts per SIMD), cold-path host timing, TERM k = 100 timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c18; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
for th in 1 default; do
  echo "== cold 100M host timing, RGPU_HOST_THREADS=$th" | tee -a $OUT/ab.log
  if [ $th = default ]; then unset RGPU_HOST_THREADS; else export RGPU_HOST_THREADS=$th; fi
  RGPU_HOST_TIMING=1 DOCS=100000000 timeout 600 python scripts/run_workload.py cold 5 2>&1 | grep -i "prepare host\|cold wall" | cut -c1-400 | tee -a $OUT/ab.log
done
unset RGPU_HOST_THREADS
echo "== term 10M / 100M" | tee -a $OUT/ab.log
DOCS=10000000 timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_search_term[^)]*)" | tee -a $OUT/ab.log
DOCS=100000000 timeout 600 python scripts/run_workload.py term 20 2>&1 | tail -1 | grep -o "'k_search_term[^)]*)" | tee -a $OUT/ab.log
