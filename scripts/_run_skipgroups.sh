#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== shipped parity (whole file but the sharded tests)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not shard" 2>&1 | tail -3
echo "== shipped cold 10M"; timeout 300 python scripts/run_workload.py cold 5 2>&1 | tail -1
echo "== shipped cold 100M"; DOCS=100000000 timeout 600 python scripts/run_workload.py cold 3 2>&1 | tail -1
