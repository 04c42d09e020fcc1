#!/bin/bash
# One gpurun call = one session: parity first, then A/B of the kernel variants under build_variants/ (built locally,
# shipped with the snapshot). usage: gpurun --timeout 900 -- 'bash scripts/gpu_session.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s1}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/session.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log
tail -5 $OUT/pytest.log | tee -a $OUT/session.log
run() {  # label, lib, env..., workload
  local label=$1 lib=$2; shift 2
  echo "== $label" | tee -a $OUT/session.log
  ( [ -n "$lib" ] && export RUCENE_GPU_LIB=$R/build_variants/$lib; env "$@" timeout 300 python scripts/run_workload.py 2>&1 | tail -3 ) | tee -a $OUT/session.log
}
for w in "" and_w5.so and_w8.so; do
  echo "== and3 lib=${w:-default(w6)}" | tee -a $OUT/session.log
  ( [ -n "$w" ] && export RUCENE_GPU_LIB=$R/build_variants/$w; timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -2 ) | tee -a $OUT/session.log
  ( [ -n "$w" ] && export RUCENE_GPU_LIB=$R/build_variants/$w; ABPI=4 timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -1 ) | tee -a $OUT/session.log
done
for bpi in 0 32 128 256 512; do
  echo "== term prune BPI=$bpi" | tee -a $OUT/session.log
  BPI=$bpi timeout 300 python scripts/run_workload.py term 10 2>&1 | tail -1 | tee -a $OUT/session.log
done
echo "== term noprune" | tee -a $OUT/session.log
RUCENE_GPU_LIB=$R/build_variants/term_noprune.so timeout 300 python scripts/run_workload.py term 10 2>&1 | tail -1 | tee -a $OUT/session.log
echo "== counters (expcount build)" | tee -a $OUT/session.log
RUCENE_GPU_LIB=$R/build_variants/expcount.so timeout 300 python scripts/run_workload.py term 3 2>&1 | tail -2 | tee -a $OUT/session.log
RUCENE_GPU_LIB=$R/build_variants/expcount.so timeout 300 python scripts/run_workload.py and3 3 2>&1 | tail -2 | tee -a $OUT/session.log
echo "== bench.py" | tee -a $OUT/session.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/session.log
python -c "
import json
d=json.load(open('$OUT/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','parity_vs_oracle_full_batch','gpu_over_cpu') if k in d}); print(d['roofline']); print(d['kernels_ms_per_step'])
" 2>&1 | tee -a $OUT/session.log
