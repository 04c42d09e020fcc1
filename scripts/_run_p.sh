cd $GRAFT_REPO_ROOT
for w in and3 or10; do
  bash scripts/prof.sh $w prof_r03_$w > gpurun_out/prof_r03_$w.log 2>&1; tail -1 gpurun_out/prof_r03_$w.log | cut -c1-200
done
for w in and3 or10; do
  PROF_SHORT=1 DOCS=100000000 bash scripts/prof.sh $w prof_r03_${w}_big > gpurun_out/prof_r03_${w}_big.log 2>&1; tail -1 gpurun_out/prof_r03_${w}_big.log | cut -c1-200
done
