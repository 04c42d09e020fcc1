#!/bin/bash
# round 5, GPU call 41: the bench line twice with the driver's arguments (the interpreter's collector off inside timed regions)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c41; mkdir -p $OUT
cd $R
for i in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > $OUT/bench_head$i.json 2> $OUT/bench_head$i.err
  python -c "
import json; d=json.loads(open('$OUT/bench_head$i.json').read().strip().splitlines()[-1]); print('head$i', round(d['value']/1e6,2), {k: round(v,4) for k,v in d['streams'].items() if k!='note'}, round(d['without_sketches']['queries_per_sec']/1e6,2))" | tee -a $OUT/ab.log
done
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1800 > $OUT/show.log; head -2 $OUT/show.log
