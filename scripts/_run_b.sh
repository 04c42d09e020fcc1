cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/$1/pytest.log | cut -c1-300
timeout 900 python bench.py --configs ${2:-and3,cold,out_of_cache} > gpurun_out/$1/bench.json 2> gpurun_out/$1/bench.err; tail -3 gpurun_out/$1/bench.err
python scripts/show_bench.py gpurun_out/$1/bench.json
