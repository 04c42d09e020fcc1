cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error" gpurun_out/$1/pytest.log | tail -12 | cut -c1-300
timeout 600 python bench.py --force-dist --configs and3 --no-cpu-baseline --steps 5 > gpurun_out/$1/bench_dist.json 2> gpurun_out/$1/bench_dist.err; echo "force-dist rc=$?"; tail -2 gpurun_out/$1/bench_dist.err | cut -c1-300
python scripts/show_bench.py gpurun_out/$1/bench_dist.json | cut -c1-600
