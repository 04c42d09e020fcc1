cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lazy_disjunctions or conjunction or must or phrase or long_clause or counters" > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error\|^E " gpurun_out/$1/pytest.log | tail -8 | cut -c1-400
timeout 600 python scripts/run_workload.py and3 5 > gpurun_out/$1/and3.log 2>&1; echo "and3 rc=$?"; tail -1 gpurun_out/$1/and3.log | grep -o "'k_search_and': ([0-9]*, [0-9.]*)"
