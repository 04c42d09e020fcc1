#!/bin/bash
# round 5, GPU call 29: the bench line after its two changes (deferred leg warmed over all four scratch slots; cold wall on a released store)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c29; mkdir -p $OUT
cd $R
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/ab.log
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/ab.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | cut -c1-1800 > $OUT/show.log; head -3 $OUT/show.log
