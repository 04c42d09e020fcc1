#!/bin/bash
# developer sweep: k_decode_terms prefetch depth (builds variants next to the product library, never replaces it)
cd $GRAFT_REPO_ROOT
mkdir -p build_variants
for d in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DRGPU_DECODE_DEPTH=$d -o build_variants/dec$d.so rucene_amd/csrc/rgpu_api.hip 2>/dev/null &
done
wait
for d in 1 2 3 4; do
  echo DEPTH=$d; RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/dec$d.so python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, rucene_amd
from rucene_amd import indexgen
seg = indexgen.build_zipf(10_000_000, 1_000_000)
ctx = rucene_amd.Context(profile_kernels=True)
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
sel = np.tile(seg.terms[seg.terms["doc_freq"] >= 128], 16)
total = int(sel["doc_freq"].sum())
d = torch.empty((total,), dtype=torch.int32, device="cuda"); f = torch.empty((total,), dtype=torch.int32, device="cuda")
for _ in range(2): leaf.segment.decode_terms_device(sel, d.data_ptr(), f.data_ptr())
ctx.kernel_stats_reset()
for _ in range(5): leaf.segment.decode_terms_device(sel, d.data_ptr(), f.data_ptr())
st = ctx.kernel_stats()["k_decode_terms"]; print(st["total_ms"]/st["launches"], len(sel), total)
ctx.close()
PY
done
