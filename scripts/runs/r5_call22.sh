#!/bin/bash
# round 5, GPU call 22: k_or_lazy at four wavefronts per SIMD — 8192-doc windows (8 KB of LDS per wavefront) with four or two
# prefetched run heads (126 / 114 VGPRs, no scratch) against the shipped 16384-doc windows at three wavefronts (152 VGPRs)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c22; mkdir -p $OUT
cd $R
run() {  # lib docs
  local lib=$1 docs=$2
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== or10 docs=$docs lib=$lib" | tee -a $OUT/ab.log
  DOCS=$docs timeout 600 python scripts/run_workload.py or10 6 2>&1 | tail -1 | grep -o "'k_or_lazy[^)]*)\|'k_or_wide[^)]*)\|'k_score_terms[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default lz_s4w3 lz_s4w4p4 lz_s4w4p2 lz_s8p4; do run $lib 10000000; done
done
for lib in default lz_s4w4p4 lz_s4w4p2; do run $lib 100000000; done
for lib in lz_s4w4p4 lz_s4w4p2; do
  RUCENE_GPU_LIB=$R/build_variants/$lib.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lazy or wide or deferred or disjunctions" 2>&1 | tail -1 | tee -a $OUT/ab.log
done
