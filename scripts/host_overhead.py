"""Developer probe (run on the GPU box): host-side cost of one enqueue-only rgpu_search_batch_device call vs the
steady-state time per call on one stream, for 1024- and 8192-query single-term batches."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rucene_amd
from rucene_amd import indexgen
seg = indexgen.build_zipf(10_000_000, 1_000_000)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
T = rucene_amd.TermQuery
for nq in (1024, 8192):
    tids = indexgen.log_uniform_ranks(nq, 1, 10_000, 77) - 1
    qp, tp = s.pack([T(int(t)) for t in tids], leaf)
    h = torch.empty((nq, 10), dtype=torch.int64, device="cuda"); t = torch.empty((nq,), dtype=torch.int64, device="cuda")
    for _ in range(3): leaf.segment.search_batch_device(qp, tp, 10, h.data_ptr(), t.data_ptr())
    ctx.synchronize()
    # host-only cost: enqueue 4 calls (the scratch slots), then sync; repeat
    tot = 0.0; n = 0
    for rep in range(20):
        t0 = time.perf_counter()
        for _ in range(4): leaf.segment.search_batch_device(qp, tp, 10, h.data_ptr(), t.data_ptr())
        tot += time.perf_counter() - t0; n += 4
        ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): leaf.segment.search_batch_device(qp, tp, 10, h.data_ptr(), t.data_ptr())
    ctx.synchronize()
    el = time.perf_counter() - t0
    print("nq", nq, "host enqueue ms/call", round(1e3 * tot / n, 4), "steady ms/call", round(1e3 * el / 40, 4))
ctx.close()
