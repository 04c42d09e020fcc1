#!/bin/bash
# Variant builds of librucene_gpu.so for A/B runs on the GPU box (RUCENE_GPU_LIB=build_variants/<name>.so selects one at run time):
#   scripts/build_variants.sh <name> <-D flags ...>      e.g.  scripts/build_variants.sh orx_w16_32k -DRGPU_ORX_WAVES=16 -DRGPU_ORX_LOOK=5 -DRGPU_ORX_WS16=32768
# The next steps DESIGN.md §8 names for k_or_wide, ready to build:
#   orx_w16_32k   one 16-wavefront workgroup per CU, 32768-doc windows (128 KB of accumulators)
#   orx_time      -DRGPU_ORX_TIME: wave-cycles per phase of the window loop (rgpu_debug_counters; run_workload.py prints them)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=${1:?variant name}; shift
mkdir -p $R/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function "$@" \
  -o $R/build_variants/$NAME.so $R/rucene_amd/csrc/rgpu_api.hip -L/opt/rocm/lib -lrccl
ls -la $R/build_variants/$NAME.so
