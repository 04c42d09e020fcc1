#!/bin/bash
# round 6: the full GPU suite, smoke, then bench.py as the driver runs it; prints the contract line's keys
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6b}
shift || true
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -4 $OUT/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a $OUT/session.log
if [ "${SKIP_SMOKE:-0}" != "1" ]; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/session.log; fi
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/session.log
cp $R/bench_detail.json $OUT/bench_detail.json 2>/dev/null
python - "$OUT/bench.json" <<'P' 2>&1 | tee -a $OUT/session.log
import json, sys
lines = open(sys.argv[1]).read().splitlines()
print("stdout lines:", len(lines), "bytes of last:", len(lines[-1]) if lines else 0)
d = json.loads(lines[-1])
for k, v in d.items():
    if k not in ("config", "cpu_baseline", "parity", "unit", "metric", "data", "dtype"):
        print(" ", k, "=", v)
P
grep -v "^bench detail\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/bench.err | tail -5 | tee -a $OUT/session.log
