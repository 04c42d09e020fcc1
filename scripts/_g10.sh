cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g10
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g10/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g10/pytest.log | cut -c1-300
for i in 1 2; do python scripts/run_workload.py and3 5 | tail -1; done
for v in s0 s4 s16; do echo $v; RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so python scripts/run_workload.py and3 5 | tail -1; done
RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/cnt.so python scripts/run_workload.py and3 1 | tail -2 | head -1
