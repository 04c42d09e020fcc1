set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g4/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g4/pytest.log
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/g4/bench.json 2> gpurun_out/g4/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/g4/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/g4/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step','kernels_ms_per_step','gpu_over_cpu','parity_vs_oracle_full_batch') if k in d}); print(d['roofline']); print(d['cpu_baseline'])
PY
