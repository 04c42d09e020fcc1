#!/bin/bash
# round 5, GPU call 39: the bench line with a pre-warmed GPU and the without-sketches leg with a resident plan
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c39; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketches" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1800 > $OUT/show.log; head -3 $OUT/show.log
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('without', d.get('without_sketches')); print('big term', {k: d['configs']['out_of_cache']['term'][k] for k in ('ms_per_step','queries_per_sec','parity_vs_oracle')}, d['configs']['out_of_cache']['term']['roofline'])" | tee -a $OUT/ab.log
