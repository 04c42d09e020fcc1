cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g9
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g9/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/g9/pytest.log | cut -c1-400
for w in and3 or10; do python scripts/run_workload.py $w 5 | tail -1; done
