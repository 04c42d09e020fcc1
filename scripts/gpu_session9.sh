#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s9}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -15 $OUT/pytest.log | tee -a $OUT/session.log
echo "== bench --force-dist" | tee -a $OUT/session.log
timeout 600 python bench.py --force-dist --configs none --steps 10 --warmup 2 > $OUT/bench_fd.json 2> $OUT/bench_fd.err; echo "rc=$?" | tee -a $OUT/session.log
tail -4 $OUT/bench_fd.err | tee -a $OUT/session.log
python -c "
import json; d=json.load(open('$OUT/bench_fd.json')); print(d['value'], d['ms_per_step'], d.get('parity_vs_oracle_full_batch'), d['streams'])" 2>&1 | tee -a $OUT/session.log
echo "== or10 one launch" | tee -a $OUT/session.log
timeout 300 python scripts/run_workload.py or10 3 2>&1 | tail -1 | tee -a $OUT/session.log
