#!/usr/bin/env python3
"""Timeline of one k_search_term launch from a -DRGPU_TERM_TRACE build (RUCENE_GPU_LIB=build_variants/term_trace.so).
usage: term_timeline.py [docs]"""
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
tids = bench.build_queries(1024, "term", bench.SEED_QUERIES)
import torch
h = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
t = torch.empty((1024,), dtype=torch.int64, device="cuda")
for _ in range(6):
    s.search_uniform_device(_lib.OP_TERM, tids, leaf, 10, h.data_ptr(), t.data_ptr())
    ctx.synchronize()
L = C.CDLL(_lib.lib_path())
REC = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("q", "<i4"), ("chunk", "<i4"), ("blocks", "<i4"), ("unpacked", "<i4"),
                ("d_term", "<u4"), ("d_table", "<u4"), ("d_sketch", "<u4"), ("d_pad", "<u4"), ("ph", "<u4", (8,))])
buf = np.zeros(1 << 17, dtype=REC)
n = L.rgpu_debug_trace(C.c_void_p(buf.ctypes.data), C.c_int32(buf.size))
st = ctx.kernel_stats()["k_search_term"]
rec = buf[buf["t1"] > 0]
t0 = rec["t0"].min()
start = (rec["t0"] - t0) / 100.0
end = (rec["t1"] - t0) / 100.0
dur = end - start
print("k_search_term median %.1f us (HIP events); %d items traced; launch span %.1f us" % (1e3 * st["median_ms"], rec.size, end.max()))
print("item duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("item start us: p50 %.1f p90 %.1f max %.1f" % (*np.percentile(start, [50, 90]), start.max()))
print("sum of item durations %.0f us = %.1f x the span" % (dur.sum(), dur.sum() / end.max()))
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    tcut = frac * end.max()
    print("  at %3.0f %% of the span (%.1f us): %5d items running, %5d not started" % (100 * frac, tcut, ((start <= tcut) & (end > tcut)).sum(), (start > tcut).sum()))
order = np.argsort(-end)[:14]
df = seg.terms["doc_freq"]
print("the last items to finish:")
for i in order:
    r = rec[i]
    print("   end %.1f  dur %.1f  start %.1f  q %d (df %d) chunk %d  blocks looked at %d" % (end[i], dur[i], start[i], r["q"], int(df[tids[r["q"], 0]]), r["chunk"], r["blocks"]))
heads = rec["chunk"] == 0
print("head items (chunk 0): mean %.1f us, max %.1f; others: mean %.1f, max %.1f" % (dur[heads].mean(), dur[heads].max(), dur[~heads].mean() if (~heads).any() else 0, dur[~heads].max() if (~heads).any() else 0))
live = rec["d_term"] > 0
print("phases (us from the item's start; items with a term): term descriptor in registers p50 %.1f / p90 %.1f; score table built %.1f / %.1f; sketch threshold folded %.1f / %.1f; item done %.1f / %.1f" % (
    *np.percentile(rec["d_term"][live] / 100.0, [50, 90]), *np.percentile(rec["d_table"][live] / 100.0, [50, 90]),
    *np.percentile(rec["d_sketch"][live & (rec["d_sketch"] > 0)] / 100.0, [50, 90]), *np.percentile(dur[live], [50, 90])))
A = np.stack([rec["blocks"].astype(np.float64), np.ones(rec.size)], axis=1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("least squares: item us = %.4f x blocks looked at + %.2f" % tuple(coef))
ctx.close()

# shader-clock cycles per phase of term_blocks_fast (TERM_PH_ADD), for the items that unpacked something
busy = rec["blocks"] > 0
ph = rec["ph"][busy].astype(np.float64)
nb = rec["blocks"][busy].astype(np.float64)
names = ["chunk frontiers + first request", "per chunk: wait for its directory + bounds + todo", "per block: rows arrive + staged", "per block: freqs, scores, compare",
         "per entering block: doc ids + keys", "per entering block: the group's list (offer)", "term_blocks_fast in all", "offers"]
tot = ph[:, 6].sum()
cyc_per_us = tot / np.maximum(1e-9, (dur[busy] - rec["d_sketch"][busy] / 100.0)).sum()
print("phases of term_blocks_fast over %d items that unpacked blocks (%.0f blocks; ~%.0f shader cycles per us):" % (busy.sum(), nb.sum(), cyc_per_us))
for i in range(6):
    print("   %-52s %5.1f %% of its cycles, %7.0f cycles per unpacked block" % (names[i], 100.0 * ph[:, i].sum() / tot, ph[:, i].sum() / nb.sum()))
print("   offers per unpacked block: %.2f; unaccounted (take(), ring bookkeeping, loop): %.1f %%" % (ph[:, 7].sum() / nb.sum(), 100.0 * (1 - ph[:, :6].sum() / tot)))
slow = np.argsort(-dur)[:200]
sl = rec[slow]
phs = sl["ph"].astype(np.float64)
print("the 200 longest items: %.1f blocks each;" % sl["blocks"].mean(), " ".join("%s %.0f %%" % (n.split(":")[0][:14], 100 * phs[:, i].sum() / phs[:, 6].sum()) for i, n in enumerate(names[:6])))
