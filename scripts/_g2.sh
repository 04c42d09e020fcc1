set -u
cd $GRAFT_REPO_ROOT
for b in 0 32 128 256 512; do echo "BPI=$b"; BPI=$b timeout 300 python scripts/run_workload.py term 5 2>&1 | tail -1; done
