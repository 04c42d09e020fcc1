cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for v in "${@:2}"; do
  RUCENE_GPU_LIB=$PWD/build_variants/$v.so timeout 600 python scripts/run_workload.py and3 5 > gpurun_out/$1/and3_$v.log 2>&1; echo "$v rc=$?"; tail -1 gpurun_out/$1/and3_$v.log | grep -o "'k_search_and': ([0-9]*, [0-9.]*)"
done
