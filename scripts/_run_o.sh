cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_disjunctions or lazy_disjunctions" > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error\|assert" gpurun_out/$1/pytest.log | tail -12 | cut -c1-400
timeout 600 python scripts/run_workload.py or10 3 > gpurun_out/$1/or10.log 2>&1; echo "or10 rc=$?"; tail -1 gpurun_out/$1/or10.log | cut -c1-1200
RUCENE_GPU_LIB=$PWD/build_variants/lz_time.so timeout 600 python scripts/run_workload.py or10 3 > gpurun_out/$1/or10_time.log 2>&1; echo "or10 rc=$?"; tail -2 gpurun_out/$1/or10_time.log | cut -c1-1500
