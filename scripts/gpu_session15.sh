#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s15}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "rescorer or elias or phrase" > $OUT/pytest_phrase.log 2>&1; echo "phrase rc=$?" | tee $OUT/session.log
tail -40 $OUT/pytest_phrase.log | tee -a $OUT/session.log
