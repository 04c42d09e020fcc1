#!/bin/bash
# developer sweep: TERM / AND kernels vs prefetch depth and workgroup size (variants are built next to the product
# library and selected with RUCENE_GPU_LIB; the product library is never replaced)
cd $GRAFT_REPO_ROOT
mkdir -p build_variants
build_var() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function $2 -o build_variants/$1.so rucene_amd/csrc/rgpu_api.hip 2>/dev/null & }
build_var d2 "-DRGPU_PREFETCH_DEPTH=2"; build_var d3 "-DRGPU_PREFETCH_DEPTH=3"; build_var d6 "-DRGPU_PREFETCH_DEPTH=6"
build_var w4 "-DRGPU_TERM_WAVES=4"; build_var w16 "-DRGPU_TERM_WAVES=16"; wait
echo base; for w in term and3; do python scripts/run_workload.py $w 5 | tail -1; done
for v in d2 d3 d6 w4 w16; do echo $v; for w in term and3; do RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so python scripts/run_workload.py $w 5 | tail -1; done; done
