cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for d in 32 128 256; do
ORBM=$d timeout 600 python scripts/run_workload.py or10 3 > gpurun_out/$1/or10_$d.log 2>&1; echo "or10 ORBM=$d rc=$?"; tail -1 gpurun_out/$1/or10_$d.log | grep -o "'k_bitmap_build': ([0-9]*, [0-9.]*)\|'k_or_lazy': ([0-9]*, [0-9.]*)\|'k_or_wide': ([0-9]*, [0-9.]*)\|'k_score_terms': ([0-9]*, [0-9.]*)\|'or_lazy[a-z_]*': ([0-9]*" | tr '\n' ' '; echo
done
