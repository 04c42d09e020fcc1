#!/bin/bash
# round 5, GPU call 2: where k_search_and's time goes (RGPU_AND_TIME), item size sweep, sharded path in a world of one
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c2; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | tail -3 | cut -c1-1200 | tee -a $OUT/ab.log
}
run and_time and3 10000000 ABPI=8
run and_time and3 10000000 ABPI=16
for a in 12 16 20 24; do run default and3 10000000 ABPI=$a; done
for a in 12 16 20; do run and_w5g4 and3 10000000 ABPI=$a; done
unset RUCENE_GPU_LIB
timeout 600 python scripts/host_overhead_sharded.py 2>&1 | tail -12 | tee $OUT/sharded.log
KIND=and3 timeout 600 python scripts/host_overhead_sharded.py 2>&1 | tail -12 | tee $OUT/sharded_and3.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shard or sloppy or two_phase or record" 2>&1 | tail -3 | tee $OUT/pytest.log
run and_time and3 100000000 ABPI=32
for a in 32 48 60; do run default and3 100000000 ABPI=$a; done
