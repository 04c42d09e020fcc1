#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s11}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests -m gpu -x -q -k "corrupt" > $OUT/pytest_corrupt.log 2>&1; echo "corrupt rc=$?" | tee $OUT/session.log
tail -30 $OUT/pytest_corrupt.log | tee -a $OUT/session.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log
tail -15 $OUT/pytest.log | tee -a $OUT/session.log
