"""Developer analysis (not a test; run by hand: `python tests/analysis/maxscore_estimate.py`). Which share of the postings of the
bench's 10-term OR top-100 queries lies in lists that MaxScore would treat as non-essential? A list is
non-essential when the score bounds of it and of all cheaper-bounded lists together cannot reach the final top-100 threshold, so
a doc found only there can never enter: such lists need only be probed at candidates from the essential lists. Per-term bound =
weight * (k1 + 1) * g(freq 10, shortest doc of the corpus) — deliberately loose. Uses the ORACLE, hence lives under tests/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main(n_docs=10_000_000, n_terms=1_000_000, n_queries=24, k=100):
    from oracle import binding as oracle
    from rucene_amd import indexgen
    import rucene_amd
    seg = indexgen.build_zipf(n_docs, n_terms)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    searcher = oracle.Searcher([oseg])
    rows = (indexgen.log_uniform_ranks(1024 * 10, 1, 10000, seed=0x52).reshape(-1, 10) - 1)[:n_queries]
    L = oracle.lib()
    table = np.array([L.orc_norm_table(i) for i in range(256)], dtype=np.float32)
    avgdl = np.float32(np.float64(seg.sum_total_term_freq) / np.float64(seg.doc_count))
    k1, b = np.float32(1.2), np.float32(0.75)
    cache = (k1 * ((np.float32(1) - b) + b * (table / avgdl))).astype(np.float32)
    gmax = 10.0 / (10.0 + float(cache[seg.norms.max()]))
    total = non_essential = 0
    t0 = time.time()
    for row in rows:
        row = [int(x) for x in row]
        _, scores, _ = searcher.search(oracle.OP_OR, row, k, tie_mode=oracle.TIE_CANONICAL)
        theta = float(scores[-1]) if len(scores) == k else 0.0
        bounds = []
        for t in row:
            df = int(seg.terms[t]["doc_freq"])
            w, _, _ = rucene_amd.bm25_compute_weight(1.2, 0.75, n_docs, seg.doc_count, seg.sum_total_term_freq, [df])
            bounds.append((w * 2.2 * gmax, df))
        bounds.sort()
        acc = 0.0
        for u, df in bounds:
            if acc + u >= theta:
                break
            acc += u
            non_essential += df
        total += sum(df for _, df in bounds)
    print("OR top-%d, 10 terms: %.1f%% of %d postings lie in non-essential lists (%d queries, %.0f s)"
          % (k, 100.0 * non_essential / total, total, len(rows), time.time() - t0))


if __name__ == "__main__":
    main()
