cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$2" > gpurun_out/$1/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "FAILED\|passed\|failed\|Error\|assert\|^E " gpurun_out/$1/pytest.log | tail -14 | cut -c1-400
