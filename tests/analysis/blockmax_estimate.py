"""Developer analysis (not a test; run by hand: `python tests/analysis/blockmax_estimate.py`). How many FullBlocks of the
headline workload could a block-max bound skip? Uses the ORACLE's decode, hence lives under tests/.

Two models of the threshold a block is compared with:
  * sequential: one top-k per query fed in posting order (the best any schedule can do);
  * group-local: the kernel's present partitioning — 8 items of B blocks advance in lock step and share one top-k, nothing
    arrives from other groups of the same query.
A block is skippable when the largest score any of its postings can reach does not beat the k-th best so far (ties lose to
earlier docs under score desc / doc asc). Scores of one term are ordered by g = f / (f + cache[norm]) for any positive weight.
"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main(n_docs=10_000_000, n_terms=1_000_000, n_queries=96, k=10):
    from oracle import binding as oracle
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(n_docs, n_terms)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    ranks = indexgen.log_uniform_ranks(1024, 1, 10000, seed=0x51)[:n_queries] - 1
    L = oracle.lib()
    table = np.array([L.orc_norm_table(i) for i in range(256)], dtype=np.float32)
    avgdl = np.float32(np.float64(seg.sum_total_term_freq) / np.float64(seg.doc_count))
    k1, b = np.float32(1.2), np.float32(0.75)
    cache = (k1 * ((np.float32(1) - b) + b * (table / avgdl))).astype(np.float32)
    per_term = []
    for r in ranks:
        d, f = oseg.decode_term(seg.terms[int(r)])
        nb = len(d) // 128
        if nb:
            ff = f[:nb * 128].astype(np.float32)
            per_term.append((ff / (ff + cache[seg.norms[d[:nb * 128]]])).reshape(nb, 128))

    def feed(heap, row):
        for v in row:
            if len(heap) < k:
                heapq.heappush(heap, v)
            elif v > heap[0]:
                heapq.heapreplace(heap, v)

    total = sum(g.shape[0] for g in per_term)
    skipped = 0
    for g in per_term:
        heap, gmax = [], g.max(axis=1)
        for bi in range(g.shape[0]):
            if len(heap) == k and gmax[bi] <= heap[0]:
                skipped += 1
            else:
                feed(heap, g[bi])
    print("sequential threshold: %.1f%% of %d FullBlocks skippable" % (100.0 * skipped / total, total))
    for blocks_per_item in (8, 32, 128):
        skipped = 0
        for g in per_term:
            nb, gmax = g.shape[0], g.max(axis=1)
            for g0 in range(0, nb, 8 * blocks_per_item):
                heap = []
                items = [range(g0 + i * blocks_per_item, min(nb, g0 + (i + 1) * blocks_per_item)) for i in range(8)]
                for step in range(blocks_per_item):
                    for it in items:
                        if step < len(it):
                            bi = it[step]
                            if len(heap) == k and gmax[bi] <= heap[0]:
                                skipped += 1
                            else:
                                feed(heap, g[bi])
        print("group-local threshold, %3d blocks per item: %.1f%% skippable" % (blocks_per_item, 100.0 * skipped / total))
    # the same, but every group starts from the threshold its query's HEAD item (the first `blocks_per_item` blocks, run in an
    # earlier wave of workgroups) has published — what SharedTau already transports today
    for blocks_per_item in (8, 32, 128):
        skipped = 0
        for g in per_term:
            nb, gmax = g.shape[0], g.max(axis=1)
            head = []
            for bi in range(min(nb, blocks_per_item)):
                feed(head, g[bi])
            floor = head[0] if len(head) == k else -1.0
            for g0 in range(0, nb, 8 * blocks_per_item):
                heap = []
                items = [range(g0 + i * blocks_per_item, min(nb, g0 + (i + 1) * blocks_per_item)) for i in range(8)]
                for step in range(blocks_per_item):
                    for it in items:
                        if step < len(it):
                            bi = it[step]
                            thr = max(floor, heap[0]) if len(heap) == k else floor
                            if gmax[bi] <= thr:
                                skipped += 1
                            else:
                                feed(heap, g[bi])
        print("head item's threshold + group-local, %3d blocks per item: %.1f%% skippable" % (blocks_per_item, 100.0 * skipped / total))


if __name__ == "__main__":
    main()
