#!/bin/bash
# round 5, GPU call 1: parity of the batched first probe (k_search_and) + A/B of its variants on one box
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c1; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conjunction or lazy or must or phrase or filter or min_should or live_docs or multi_leaf or k_above or rescor or payload" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/session.log
tail -5 $OUT/pytest.log | tee -a $OUT/session.log
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | tail -1 | cut -c1-900 | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in and_old default and_w5g4 and_w4g6 and_w4g2; do run $lib and3 10000000 X=1; done
done
run default and3 10000000 ABPI=16
run default and3 10000000 ABPI=32
run and_w4g6 and3 10000000 ABPI=24
run default and3 10000000 ABPI=4
run and_old and3 100000000 X=1
run default and3 100000000 X=1
run default and3 100000000 ABPI=32
run and_old phrase2 10000000 X=1
run default phrase2 10000000 X=1
