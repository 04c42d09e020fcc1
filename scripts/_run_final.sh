#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/prof.sh cold prof_r03_cold > gpurun_out/prof_r03_cold.log 2>&1; tail -1 gpurun_out/prof_r03_cold.log | cut -c1-200
PROF_SHORT=1 DOCS=100000000 bash scripts/prof.sh cold prof_r03_cold_big > gpurun_out/prof_r03_cold_big.log 2>&1; tail -1 gpurun_out/prof_r03_cold_big.log | cut -c1-200
bash scripts/gpu_full.sh r5b
