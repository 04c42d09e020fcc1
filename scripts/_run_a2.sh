cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for d in 256 1024 4096; do
ORBM=$d timeout 600 python scripts/run_workload.py and3 5 > gpurun_out/$1/and3_$d.log 2>&1; echo "and3 ORBM=$d rc=$?"; tail -1 gpurun_out/$1/and3_$d.log | grep -o "'k_bitmap_build': ([0-9]*, [0-9.]*)\|'k_search_and': ([0-9]*, [0-9.]*)" | tr '\n' ' '; echo
done
