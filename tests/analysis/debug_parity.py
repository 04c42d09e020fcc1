"""Developer check (run by hand on a GPU box): decode and search parity against the ORACLE at a chosen corpus size. Uses the oracle,\nhence lives under tests/."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rucene_amd
from rucene_amd import indexgen
from oracle import binding as orc
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seg = indexgen.build_zipf(docs, 1_000_000)
ctx = rucene_amd.Context()
leaf = rucene_amd.LeafReader.from_synthetic(seg)
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
oseg = orc.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
osr = orc.Searcher([oseg])
# decode parity for the biggest terms
for t in (0, 1, 2, 10, 100, 1000, 9999):
    d, f = leaf.segment.decode_terms(seg.terms[t:t+1])
    od, of = oseg.decode_term(seg.terms[t])
    bad = np.nonzero(d != od)[0]
    print("term", t, "df", seg.terms[t]["doc_freq"], "decode ok" if bad.size == 0 and (f == of).all() else ("MISMATCH first at %d (block %d) of %d" % (bad[0], bad[0] // 128, d.size)))
ranks = indexgen.log_uniform_ranks(1024, 1, 10000, 0x527563656E65 ^ 0x51)
tids = ranks - 1
qs = [rucene_amd.TermQuery(int(t)) for t in tids]
hits, totals = s.search_batch(qs, 10)
ops = np.zeros(1024, np.int32); offs = np.arange(1025, dtype=np.int32)
cd, cs, cc, ct, _, _ = osr.search_batch(ops, offs, tids, 10, tie_mode=orc.TIE_CANONICAL, threads=32)
nbad = 0
for i in range(1024):
    ok = (hits[i]["doc"] == cd[i]).all() and (hits[i]["score"].view(np.int32) == cs[i].view(np.int32)).all() and totals[i] == ct[i]
    if not ok:
        nbad += 1
        if nbad <= 6:
            print("query", i, "term", tids[i], "df", seg.terms[tids[i]]["doc_freq"], "totals", totals[i], ct[i])
            print("  gpu", hits[i]["doc"], hits[i]["score"])
            print("  cpu", cd[i], cs[i])
print("bad queries:", nbad)
