#!/bin/bash
# round 5, GPU call 12: the two-phase cut-off and the groups' first live candidate reduced chunk by chunk — parity and time
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c12; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "phrase or sloppy or payload" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
echo "== sloppy2" | tee -a $OUT/ab.log
DOCS=10000000 timeout 600 python scripts/run_workload.py sloppy2 3 2>&1 | tail -1 | cut -c1-900 | tee -a $OUT/ab.log
echo "== phrase2" | tee -a $OUT/ab.log
DOCS=10000000 timeout 600 python scripts/run_workload.py phrase2 3 2>&1 | tail -1 | cut -c1-900 | tee -a $OUT/ab.log
