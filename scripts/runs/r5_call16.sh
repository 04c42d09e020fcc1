#!/bin/bash
# round 5, GPU call 16: the bulk first touch planned by host threads — parity test, host timing at 100 M docs with 1 / 4 / 8 threads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c16; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bulk_first_touch or budget or corrupt or decode or docs_only" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
for th in 1 8 default; do
  echo "== cold 100M host timing, RGPU_HOST_THREADS=$th" | tee -a $OUT/ab.log
  if [ $th = default ]; then unset RGPU_HOST_THREADS; else export RGPU_HOST_THREADS=$th; fi
  RGPU_HOST_TIMING=1 DOCS=100000000 timeout 600 python scripts/run_workload.py cold 4 2>&1 | grep -i "prepare host\|cold wall" | cut -c1-400 | tee -a $OUT/ab.log
done
unset RGPU_HOST_THREADS
echo "== cold 10M" | tee -a $OUT/ab.log
RGPU_HOST_TIMING=1 DOCS=10000000 timeout 600 python scripts/run_workload.py cold 3 2>&1 | grep -i "prepare host\|cold wall" | cut -c1-400 | tee -a $OUT/ab.log
