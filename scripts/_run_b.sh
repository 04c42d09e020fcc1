cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r3b/pytest.log
timeout 600 python bench.py --configs and3 --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; tail -3 gpurun_out/r3b/bench.err
python scripts/show_bench.py gpurun_out/r3b/bench.json
