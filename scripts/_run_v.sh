cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for v in "${@:2}"; do
  RUCENE_GPU_LIB=$PWD/build_variants/$v.so timeout 600 python scripts/run_workload.py or10 3 > gpurun_out/$1/or10_$v.log 2>&1; echo "$v rc=$?"; grep "^\[lz\]" gpurun_out/$1/or10_$v.log | tail -1; grep "host\]\|lz steps" gpurun_out/$1/or10_$v.log | tail -3; grep "dbg counters" gpurun_out/$1/or10_$v.log; tail -1 gpurun_out/$1/or10_$v.log | grep -o "'k_or_lazy': ([0-9]*, [0-9.]*)\|'or_lazy[a-z_]*': ([0-9]*" | tr '\n' ' '; echo
done
