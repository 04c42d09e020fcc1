cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
RUCENE_GPU_LIB=$PWD/build_variants/lz_time.so timeout 600 python scripts/run_workload.py or10 3 > gpurun_out/$1/or10_time.log 2>&1; echo "or10 rc=$?"; tail -3 gpurun_out/$1/or10_time.log | cut -c1-1500
