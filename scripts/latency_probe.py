#!/usr/bin/env python3
"""Where a batch of ONE goes: per-kernel median durations (HIP events) and the call's wall time for single-query TERM / AND / OR
calls on the bench corpus. usage: latency_probe.py [--docs N] [--kinds term,and3,or10] [--calls 64]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--kinds", default="term,and3,or10")
    ap.add_argument("--calls", type=int, default=64)
    args = ap.parse_args()
    import rucene_amd
    from rucene_amd import indexgen, _lib
    ctx = rucene_amd.Context()
    seg = indexgen.build_zipf(args.docs, args.vocab)
    leaf = rucene_amd.LeafReader.from_synthetic(seg)
    searcher = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
    OPS = {"term": _lib.OP_TERM, "and3": _lib.OP_AND, "or10": _lib.OP_OR}
    for kind in args.kinds.split(","):
        k = bench.K_OF[kind]
        tids = bench.build_queries(1024, kind, bench.SEED_QUERIES)[:args.calls]
        plans = [searcher.pack_uniform(OPS[kind], tids[i:i + 1], leaf) for i in range(tids.shape[0])]
        for q, t in plans:
            leaf.segment.search_batch(q, t, k)
        ctx.set_profiling(False)
        lat = []
        for q, t in plans:
            t0 = time.perf_counter()
            leaf.segment.search_batch(q, t, k)
            lat.append(time.perf_counter() - t0)
        ctx.set_profiling(True)
        ctx.kernel_stats_reset()
        for q, t in plans:
            leaf.segment.search_batch(q, t, k)
        st = ctx.kernel_stats()
        ctx.set_profiling(False)
        print("%s: wall p50 %.1f us, mean %.1f us over %d single-query calls" % (kind, 1e6 * np.median(lat), 1e6 * np.mean(lat), len(lat)))
        tot = 0.0
        for name, s in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
            if s["timed_launches"] > 0:
                per_call = s["total_ms"] / len(plans)
                tot += per_call
                print("   %-22s launches/call %.2f  median %.1f us  per call %.1f us" % (name, s["launches"] / len(plans), 1e3 * s["median_ms"], 1e3 * per_call))
            elif s["launches"]:
                print("   %-22s count/call %.2f" % (name, s["launches"] / len(plans)))
        print("   kernels per call: %.1f us" % (1e3 * tot))
    ctx.close()


if __name__ == "__main__":
    main()
