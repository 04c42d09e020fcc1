cd $GRAFT_REPO_ROOT
for v in and_abl1 and_abl2 and_abl3 and_nodense; do echo "== $v"; RUCENE_GPU_LIB=$PWD/build_variants/$v.so python scripts/run_workload.py and3 5 2>&1 | tail -1 | cut -c1-400; done
echo "== and3 shipped"; python scripts/run_workload.py and3 5 2>&1 | tail -1
for abpi in 1 2 8; do echo "== and3 shipped ABPI=$abpi"; ABPI=$abpi python scripts/run_workload.py and3 5 2>&1 | tail -1 | cut -c1-300; done
echo "== cold 100M x3"; DOCS=100000000 python scripts/run_workload.py cold 3 2>&1 | tail -1
echo "== cold 10M x3"; python scripts/run_workload.py cold 3 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "corrupt or phrase or elias or docs_only or decode or advance" 2>&1 | tail -3
