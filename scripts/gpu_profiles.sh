#!/bin/bash
# rocprofv3 summaries for the round's profiles/ directory: kernel trace + PMC passes per workload (scripts/prof.sh), on the
# build that is in the tree. usage (on the GPU box): bash scripts/gpu_profiles.sh [round tag, default r06] [small|big|all]
# Afterwards copy gpurun_out/prof_<round>_<workload>/summary.txt to profiles/<round>_<workload>_rocprofv3_summary.txt
# (scripts/collect_profiles.sh does it).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r06}
WHAT=${2:-all}
cd $R
if [ "$WHAT" != "big" ]; then
  for w in term and3 or10 decode cold posdec phrase2 sloppy2; do
    bash scripts/prof.sh $w prof_${ROUND}_$w > gpurun_out/prof_${ROUND}_$w.log 2>&1
    tail -2 gpurun_out/prof_${ROUND}_$w.log | cut -c1-300
  done
fi
if [ "$WHAT" != "small" ]; then
  # out of the Infinity Cache: the 100M-doc shard
  for w in decode cold term and3 or10; do
    PROF_SHORT=1 DOCS=100000000 bash scripts/prof.sh $w prof_${ROUND}_${w}_big > gpurun_out/prof_${ROUND}_${w}_big.log 2>&1
    tail -2 gpurun_out/prof_${ROUND}_${w}_big.log | cut -c1-300
  done
fi
ls gpurun_out | grep prof_${ROUND}
