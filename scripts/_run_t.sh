cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/$1/pytest.log | tail -12 | cut -c1-300
timeout 900 python bench.py --configs or10 --steps 20 > gpurun_out/$1/bench_or10.json 2> gpurun_out/$1/bench_or10.err; echo "bench rc=$?"; tail -3 gpurun_out/$1/bench_or10.err | cut -c1-300
python scripts/show_bench.py gpurun_out/$1/bench_or10.json | cut -c1-900
