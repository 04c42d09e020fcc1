cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_disjunctions or lazy_disjunctions or search_counters or native_planner" > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error\|assert" gpurun_out/$1/pytest.log | tail -8 | cut -c1-400
timeout 900 python bench.py --configs or10 --steps 20 --no-cpu-baseline > gpurun_out/$1/bench_or10.json 2> gpurun_out/$1/bench_or10.err; echo "bench rc=$?"; tail -3 gpurun_out/$1/bench_or10.err | cut -c1-300
python scripts/show_bench.py gpurun_out/$1/bench_or10.json | grep -A4 "^or10" | cut -c1-900
