#!/bin/bash
# round 5, GPU call 40: the final tree — whole GPU suite, smoke, the bench line, the --force-dist leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c40; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
grep -n "^E " $OUT/pytest.log | head -10 | tee -a $OUT/ab.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $OUT/ab.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1800 > $OUT/show.log; head -3 $OUT/show.log
( time timeout 600 python bench.py --force-dist --configs and3 --no-cpu-baseline > $OUT/bench_dist.json 2> $OUT/bench_dist.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -3 $OUT/bench_dist.err | cut -c1-300 | tee -a $OUT/ab.log
python -c "
import json; d=json.loads(open('$OUT/bench_dist.json').read().strip().splitlines()[-1]); print('force-dist', {k: d.get(k) for k in ('value','ms_per_step','whole_index_queries_per_sec','and3_queries_per_sec')})" | tee -a $OUT/ab.log
