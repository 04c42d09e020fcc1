#!/bin/bash
# One call: the whole GPU test suite, smoke(), bench.py. usage: gpurun --timeout 1500 -- 'bash scripts/gpu_full.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-full}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -5 $OUT/pytest.log | tee -a $OUT/session.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/session.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/session.log
tail -5 $OUT/bench.err | tee -a $OUT/session.log
python scripts/show_bench.py $OUT/bench.json 2>&1 | tee -a $OUT/session.log
