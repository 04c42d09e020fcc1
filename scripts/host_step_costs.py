"""Where the host's share of a planned single-term step goes: the planner call, the search call (enqueue only), per 1024-query batch.
usage (GPU box): python scripts/host_step_costs.py   (RUCENE_GPU_LIB=build_variants/host_time.so prints search_pass's own split)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rucene_amd
from rucene_amd import indexgen
seg = indexgen.build_zipf(10_000_000, 1_000_000)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
tids = indexgen.log_uniform_ranks(1024, 1, 10_000, 0x527563656E65 ^ 0x51).reshape(-1, 1) - 1
hits = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
totals = torch.empty((1024,), dtype=torch.int64, device="cuda")
pk = s.pack_uniform(0, tids, leaf)
for _ in range(5):
    leaf.segment.search_batch_device(pk[0], pk[1], 10, hits.data_ptr(), totals.data_ptr())
torch.cuda.synchronize()
N = 2000 if "RUCENE_GPU_LIB" not in os.environ else 3
t0 = time.perf_counter()
for _ in range(N):
    pk = s.pack_uniform(0, tids, leaf)
t_plan = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    leaf.segment.search_batch_device(pk[0], pk[1], 10, hits.data_ptr(), totals.data_ptr())
t_call = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / N
print("plan %.1f us, search call (enqueue) %.1f us per batch; %.1f us per batch incl. the GPU's tail" % (1e6 * t_plan, 1e6 * t_call, 1e6 * t_all))
ctx.close()
