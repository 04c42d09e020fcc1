cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6bm
for i in 1 2; do
for m in 0 1; do echo "== RGPU_AND_MEMB_ONLY=$m"; RGPU_AND_MEMB_ONLY=$m python scripts/run_workload.py mustand 5 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r6bm/ab.log 2>&1
cat gpurun_out/r6bm/ab.log | cut -c1-400
