set -u
cd $GRAFT_REPO_ROOT
for w in term and3 or10 decode; do bash scripts/prof.sh $w prof_r01_$w > gpurun_out/prof_$w.log 2>&1; tail -3 gpurun_out/prof_$w.log | cut -c1-200; done
timeout 1200 python bench.py --steps 50 --warmup 5 --extra > gpurun_out/bench11.json 2> gpurun_out/bench11.err; echo "bench rc=$?"; tail -2 gpurun_out/bench11.err
