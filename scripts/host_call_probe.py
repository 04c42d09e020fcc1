#!/usr/bin/env python3
"""What ONE plan+search call costs the calling thread (enqueue only, no sync inside the timed calls): the headline TERM batch.
usage: host_call_probe.py [docs] — prints the per-call host time (median of 400 calls), with the GPU idle-waited every 8 calls so that
no call ever waits for a scratch slot, and back to back (where a call may wait for the slot two calls back)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, rucene_amd  # noqa: E402
from rucene_amd import indexgen, _lib  # noqa: E402
import torch  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seg = indexgen.build_zipf(docs, 1_000_000)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
tids = bench.build_queries(1024, "term", bench.SEED_QUERIES)
h = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
t = torch.empty((1024,), dtype=torch.int64, device="cuda")
for _ in range(20):
    s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, h.data_ptr(), t.data_ptr())
ctx.synchronize()
def run(sync_every):
    out = []
    for i in range(400):
        t0 = time.perf_counter()
        s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, h.data_ptr(), t.data_ptr())
        out.append(time.perf_counter() - t0)
        if sync_every and (i + 1) % sync_every == 0:
            ctx.synchronize()
    ctx.synchronize()
    a = 1e6 * np.array(out)
    return np.median(a), np.percentile(a, 10), np.percentile(a, 90)
print("host time per call, us (median, p10, p90): GPU drained every 2 calls %.1f %.1f %.1f; back to back %.1f %.1f %.1f" % (*run(2), *run(0)))
t0 = time.perf_counter()
for _ in range(400):
    s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, h.data_ptr(), t.data_ptr())
ctx.synchronize()
print("wall per call, 400 back to back + one sync: %.1f us" % (1e6 * (time.perf_counter() - t0) / 400))

# regions of K calls, synchronized on both sides (the bench's timed region), one stream and two alternating ones
st = [torch.cuda.Stream() for _ in range(4)]
bufs = [(h, t)] + [(torch.empty_like(h), torch.empty_like(t)) for _ in range(3)]
def region(K, n_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        hh, tt = bufs[i % n_streams]
        s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, hh.data_ptr(), tt.data_ptr(), st[i % n_streams].cuda_stream)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K
for K in (20, 100, 400):
    for ns in (1, 2, 3, 4):
        r = sorted(region(K, ns) for _ in range(9))
        print("regions of %3d calls on %d caller stream(s): %.1f us per call (median of 9; min %.1f)" % (K, ns, r[4], r[0]))

# the bench's region: one step of the two-call path (resident plan -> rgpu_search_batch_device) as warm-up, sync, gc, then K fused steps
import gc
pk = s.pack_uniform(_lib.OP_TERM, tids, leaf)
def bench_region(K, n_streams, warm_generic):
    if warm_generic:
        leaf.segment.search_batch_device(pk[0], pk[1], 10, h.data_ptr(), t.data_ptr(), st[0].cuda_stream)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for i in range(K):
        hh, tt = bufs[i % n_streams]
        s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, hh.data_ptr(), tt.data_ptr(), st[i % n_streams].cuda_stream)
    torch.cuda.synchronize()
    el = 1e6 * (time.perf_counter() - t0) / K
    gc.enable()
    return el
for wg in (False, True):
    r = sorted(bench_region(20, 2, wg) for _ in range(9))
    print("bench-style regions of 20 calls, 2 streams, generic warm step %s: %.1f us per call (median of 9; min %.1f)" % (wg, r[4], r[0]))
