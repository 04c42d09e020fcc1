cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 | cut -c1-200; done
