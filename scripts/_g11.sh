cd $GRAFT_REPO_ROOT
for o in 512 768 1024 1280 1536 2048; do echo ORW=$o; ORW=$o python scripts/run_workload.py or10 3 | tail -1; done
