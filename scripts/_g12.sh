cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g12/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/g12/pytest.log | cut -c1-300
python scripts/run_workload.py or10 5 | tail -1
