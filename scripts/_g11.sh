cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g11
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g11/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/g11/pytest.log | cut -c1-300
for i in 1 2; do python scripts/run_workload.py or10 5 | tail -1; done
