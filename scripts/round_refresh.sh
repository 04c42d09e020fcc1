#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/round_refresh.sh'): the GPU parity tests, the four rocprofv3 profile
# sets (scripts/prof.sh) and the bench line with extras. Outputs land in gpurun_out/; copy the summaries that
# should be judged into profiles/ (r<round>_*).
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_final.log | cut -c1-200
for w in term and3 or10 decode; do bash scripts/prof.sh $w prof_r01_$w > gpurun_out/prof_$w.log 2>&1; done
timeout 1200 python bench.py --steps 50 --warmup 5 --extra > gpurun_out/bench_extra.json 2> gpurun_out/bench_extra.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_extra.err
