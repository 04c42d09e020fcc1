"""Developer probe (run on the GPU box): what a step of the SHARDED path (rgpu_search_batch_sharded: search into a record ->
RCCL all-gather -> k_merge_lists) costs in a world of one next to the plain local search — host time per enqueue and
steady-state time per step on one and on two alternating streams, 1024 single-term queries, planned natively every step.
With one rank the all-gather is a device copy: what is measured is the path's own overhead, not xGMI latency."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rucene_amd
from rucene_amd import indexgen, _lib
seg = indexgen.build_zipf(int(os.environ.get("DOCS", "10000000")), 1_000_000)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
nq, k = 1024, 10
kind = os.environ.get("KIND", "term")
if kind == "term":
    tids = (indexgen.log_uniform_ranks(nq, 1, 10_000, 0x527563656E65 ^ 0x51) - 1).reshape(-1, 1)
    op = _lib.OP_TERM
else:
    tids = (indexgen.log_uniform_ranks(3 * nq, 1, 1000, 0x527563656E65 ^ 0x51 ^ 0xA3) - 1).reshape(-1, 3)
    op = _lib.OP_AND
packed = s.pack_uniform(op, tids, leaf)
lanes = [(torch.cuda.Stream(), torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq,), dtype=torch.int64, device="cuda")) for _ in range(2)]


def local(pk, lane):
    leaf.segment.search_batch_device(pk[0], pk[1], k, lane[1].data_ptr(), lane[2].data_ptr(), lane[0].cuda_stream)


def sharded(pk, lane):
    comm.search_batch_sharded(leaf.segment, pk[0], pk[1], k, lane[1].data_ptr(), lane[2].data_ptr(), lane[0].cuda_stream)


for name, fn in (("local", local), ("sharded", sharded)):
    for n_lanes in (1, 2):
        for replan in (False, True):
            for i in range(6):
                fn(packed, lanes[i % n_lanes])
            torch.cuda.synchronize()
            tot, n = 0.0, 0
            for rep in range(20):   # host-only cost: four enqueues (the scratch / record slots), then a sync
                t0 = time.perf_counter()
                for i in range(4):
                    fn(s.pack_uniform(op, tids, leaf) if replan else packed, lanes[i % n_lanes])
                tot += time.perf_counter() - t0
                n += 4
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                fn(s.pack_uniform(op, tids, leaf) if replan else packed, lanes[i % n_lanes])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print("%-8s streams %d replan %d: host enqueue %.4f ms/call, steady %.4f ms/step" % (name, n_lanes, int(replan), 1e3 * tot / n, 1e3 * el / 200), flush=True)
t0 = time.perf_counter()
for _ in range(200):
    s.pack_uniform(op, tids, leaf)
print("pack_uniform alone: %.4f ms" % (1e3 * (time.perf_counter() - t0) / 200))
comm.close()
ctx.close()
