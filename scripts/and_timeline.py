#!/usr/bin/env python3
"""Timeline of one k_search_and launch from a -DRGPU_AND_TRACE build (RUCENE_GPU_LIB=build_variants/and_trace.so): how long the
launch's items run, when the last ones start, which queries own the tail. usage: and_timeline.py [docs]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rucene_amd  # noqa: E402
from rucene_amd import indexgen, _lib  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seg = indexgen.build_zipf(docs, 1_000_000)
ctx = rucene_amd.Context(profile_kernels=True)
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
tids = bench.build_queries(1024, "and3", bench.SEED_QUERIES)
import torch
h = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
t = torch.empty((1024,), dtype=torch.int64, device="cuda")
for _ in range(4):
    s.search_uniform_device(_lib.OP_AND, tids, leaf, 10, h.data_ptr(), t.data_ptr())
    ctx.synchronize()
L = C.CDLL(_lib.lib_path())
REC = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("q", "<i4"), ("chunk", "<i4"), ("blocks", "<i4"), ("popped", "<i4")])
buf = np.zeros(1 << 17, dtype=REC)
n = L.rgpu_debug_trace(C.c_void_p(buf.ctypes.data), C.c_int32(buf.size))
st = ctx.kernel_stats()["k_search_and"]
rec = buf[buf["t1"] > 0]
t0 = rec["t0"].min()
start = (rec["t0"] - t0) / 100.0   # us
end = (rec["t1"] - t0) / 100.0
dur = end - start
print("k_search_and median %.1f us (HIP events); %d items traced; launch span %.1f us" % (1e3 * st["median_ms"], rec.size, end.max()))
print("item duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("sum of item durations %.0f us = %.1f x the span (mean concurrency in waves)" % (dur.sum(), dur.sum() / end.max()))
for frac in (0.5, 0.75, 0.9, 0.95, 1.0):
    tcut = frac * end.max()
    running = ((start <= tcut) & (end > tcut)).sum()
    print("  at %3.0f %% of the span (%.0f us): %5d items running, %5d not started yet" % (100 * frac, tcut, running, (start > tcut).sum()))
order = np.argsort(-end)[:12]
df = seg.terms["doc_freq"]
print("the last items to finish:")
for i in order:
    r = rec[i]
    print("   end %.1f us  dur %.1f us  start %.1f us  q %d chunk %d  blocks %d  popped %d" % (end[i], dur[i], start[i], r["q"], r["chunk"], r["blocks"], r["popped"]))
# per (device-order) query: total item time, popped survivors
qs = rec["q"]
tot = np.bincount(qs, weights=dur, minlength=1024)
pop = np.bincount(qs, weights=rec["popped"], minlength=1024)
blk = np.bincount(qs, weights=rec["blocks"], minlength=1024)
top = np.argsort(-tot)[:10]
print("heaviest queries (device order index): share of all item time, lead blocks, survivors")
for q in top:
    print("   q %4d  %.1f %%  blocks %d  popped %d  us per block %.2f" % (q, 100 * tot[q] / tot.sum(), blk[q], pop[q], tot[q] / max(1, blk[q])))
# cost model: duration ~ a * blocks + b * popped
A = np.stack([rec["blocks"].astype(np.float64), rec["popped"].astype(np.float64), np.ones(rec.size)], axis=1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("least squares: item us = %.3f x lead blocks + %.4f x survivors popped + %.2f" % tuple(coef))
ctx.close()
