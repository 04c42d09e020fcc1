set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g5/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g5/pytest.log
for w in or10 and3; do timeout 300 python scripts/run_workload.py $w 5 2>&1 | tail -1; done
for o in 512 2048; do echo ORW=$o; ORW=$o timeout 300 python scripts/run_workload.py or10 5 2>&1 | tail -1; done
