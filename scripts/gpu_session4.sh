#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s4}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -3 $OUT/pytest.log | tee -a $OUT/session.log
for bpi in 0 512; do
  echo "== term BPI=$bpi" | tee -a $OUT/session.log
  BPI=$bpi timeout 300 python scripts/run_workload.py term 10 2>&1 | tail -1 | tee -a $OUT/session.log
done
echo "== bench.py" | tee -a $OUT/session.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3 | tee -a $OUT/session.log
tail -5 $OUT/bench.err | tee -a $OUT/session.log
python - <<PY 2>&1 | tee -a $OUT/session.log
import json
d=json.load(open('$OUT/bench.json'))
def short(x, depth=0):
    if isinstance(x, dict): return {k: short(v, depth+1) for k, v in x.items() if k not in ('note','sample','unit')}
    if isinstance(x, float): return round(x, 4)
    return x
print(json.dumps(short(d), indent=1))
PY
