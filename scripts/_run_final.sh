#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/prof.sh and3 prof_r03_and3 > gpurun_out/prof_r03_and3.log 2>&1; tail -1 gpurun_out/prof_r03_and3.log | cut -c1-200
bash scripts/gpu_full.sh r5c
