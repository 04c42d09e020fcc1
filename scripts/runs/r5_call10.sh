#!/bin/bash
# round 5, GPU call 10: the whole GPU suite (deferred disjunction batches, prepared-term budget), full bench with the or10 deferred leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c10; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1500 > $OUT/show.log; head -4 $OUT/show.log
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('or10', d['configs']['or10'].get('deferred')); print('cold big', {k: d['configs']['out_of_cache']['cold'][k] for k in ('kernels_ms_total','wall_ms_incl_host_planning')})" | tee -a $OUT/ab.log
