#!/bin/bash
# round 6, first call: the new/changed GPU tests, then bench.py exactly as the driver runs it; checks the final line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6a}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "world_of_one or deferred or kernel_stats or abi" > $OUT/pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee $OUT/session.log
tail -5 $OUT/pytest_new.log | tee -a $OUT/session.log
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/session.log
echo "bench rc=$?" | tee -a $OUT/session.log
tail -c 1500 $OUT/bench.err | grep -v "^bench detail" | tail -8 | tee -a $OUT/session.log
cp $R/bench_detail.json $OUT/bench_detail.json 2>/dev/null
python - <<'P' 2>&1 | tee -a $OUT/session.log
import json,sys,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out",sys.argv[1] if len(sys.argv)>1 else "r6a","bench.json")
P
python -c "
import json
lines=open('$OUT/bench.json').read().splitlines()
print('stdout lines:',len(lines),'bytes of last:',len(lines[-1]) if lines else 0)
d=json.loads(lines[-1])
for k,v in d.items(): print(' ',k,'=',v)
" 2>&1 | tee -a $OUT/session.log
