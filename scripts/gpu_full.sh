#!/bin/bash
# One call: the whole GPU test suite, smoke(), bench.py. usage: gpurun --timeout 1500 -- 'bash scripts/gpu_full.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-full}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -5 $OUT/pytest.log | tee -a $OUT/session.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/session.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee -a $OUT/session.log
python - <<PY 2>&1 | tee -a $OUT/session.log
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','parity_vs_oracle_full_batch','gpu_over_cpu')})
print('roofline', d['roofline'])
for k,c in d.get('configs',{}).items():
    if k=='out_of_cache':
        print(k, c['block_decode']['roofline']['frac'], c['block_decode']['kernel_ms'], c['term']['kernels_ms_isolated'], c['term'].get('parity_vs_oracle_full_batch'))
    else:
        print(k, {x: c.get(x) for x in ('ms_per_step','kernels_ms_isolated','kernels_ms_per_step','parity_vs_oracle_full_batch','gpu_over_cpu','kernel_ms')}, c.get('roofline',{}).get('frac'), c.get('roofline',{}).get('traffic'))
PY
