"""Every query of the two-term phrase batch (bench.py configs.positions.phrase2) against the oracle's ExactPhraseScorer: hit counts,
doc ids and score bits. usage (GPU box): python scripts/phrase_check.py [n_queries] [docs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rucene_amd
from rucene_amd import indexgen
from oracle import binding as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
SEED = 0x527563656E65 ^ 0x51
seg = indexgen.build_zipf(docs, 1_000_000, positions=True)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic_positions(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
leaf.segment.attach_positions(leaf.pos_bytes)
leaf._pos_attached = True
ranks = indexgen.log_uniform_ranks(2 * 1024, 1, 1000, SEED ^ 0xF2).reshape(-1, 2)[:n] - 1
qs, ts = s.pack_phrases([rucene_amd.PhraseQuery([int(a), int(b)]) for a, b in ranks], leaf)
k = 10
hits, totals = leaf.segment.search_phrase_batch(qs, ts, k)
print("gpu total hits", int(totals.sum()))
ix = orc.PositionsIndex.from_files(seg.doc_bytes, seg.pos_bytes, seg.terms, leaf.term_positions)
bad = 0
t0 = time.time()
for i in range(n):
    d, sc, tot = ix.phrase_search([int(ranks[i, 0]), int(ranks[i, 1])], k, seg.norms, seg.max_doc, seg.doc_count, seg.sum_total_term_freq)
    ok = totals[i] == tot and bool((hits[i]["doc"][:d.size] == d).all()) and bool((hits[i]["score"][:d.size].view(np.int32) == sc.view(np.int32)).all())
    if not ok:
        bad += 1
        if bad <= 5:
            print("MISMATCH", i, ranks[i].tolist(), "gpu total", int(totals[i]), "oracle", tot, hits[i]["doc"][:5].tolist(), d[:5].tolist())
print("checked", n, "queries in %.1f s:" % (time.time() - t0), "all equal" if bad == 0 else "%d differ" % bad)
