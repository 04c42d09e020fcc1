#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DRGPU_PREFETCH_DEPTH=$d -o rucene_amd/librucene_gpu.so rucene_amd/csrc/rgpu_api.hip 2>/dev/null
  echo DEPTH=$d; python scripts/run_workload.py term 5 | cut -c1-110; python scripts/run_workload.py or10 2 | cut -c30-90
done
