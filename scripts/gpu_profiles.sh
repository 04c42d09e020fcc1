#!/bin/bash
# rocprofv3 summaries for the round's profiles/ directory: kernel trace + PMC passes per workload (scripts/prof.sh)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in term and3 or10 decode; do
  bash scripts/prof.sh $w prof_r02_$w > gpurun_out/prof_r02_$w.log 2>&1
  tail -3 gpurun_out/prof_r02_$w.log
done
# out of the Infinity Cache: the 100M-doc shard
DOCS=100000000 bash scripts/prof.sh decode prof_r02_decode_big > gpurun_out/prof_r02_decode_big.log 2>&1
DOCS=100000000 bash scripts/prof.sh term prof_r02_term_big > gpurun_out/prof_r02_term_big.log 2>&1
ls gpurun_out | grep prof_r02
