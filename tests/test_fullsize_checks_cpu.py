"""The property checkers of tests/fullsize_checks.py, exercised on a small index with the CPU oracle standing in for
the GPU path (they must accept a correct implementation before they are trusted to judge the HIP one at full size)."""
import numpy as np
import pytest

import fullsize_checks as fc


@pytest.fixture(scope="module")
def small(oracle):
    import __graft_entry__ as g
    g.build()
    import rucene_amd
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(60_000, 6_000)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    osr = oracle.Searcher([oseg])

    def decode(states):
        d, f, _secs = oseg.decode_terms(np.atleast_1d(states))
        return d, f

    def search(queries, k):
        hits = np.zeros((len(queries), k), dtype=rucene_amd.HIT_DTYPE)
        hits["doc"] = -1
        totals = np.zeros(len(queries), np.int64)
        for i, q in enumerate(queries):
            if isinstance(q, rucene_amd.TermQuery):
                op, tids = oracle.OP_TERM, [q.term]
            elif q.must_queries:
                op, tids = oracle.OP_AND, [c.term for c in q.must_queries]
            else:
                op, tids = oracle.OP_OR, [c.term for c in q.should_queries]
            d, s, total = osr.search(op, tids, k, tie_mode=oracle.TIE_CANONICAL)
            hits["doc"][i, :d.size], hits["score"][i, :d.size], totals[i] = d, s, total
        return hits, totals

    return rucene_amd, seg, decode, search


def test_checkers_accept_the_oracle(small):
    rucene_amd, seg, decode, search = small
    big = np.nonzero(seg.terms["doc_freq"] >= 128)[0]
    fc.check_decode(seg, decode, big)
    fc.check_term_queries(rucene_amd, seg, decode, search, [0, 1, 7, 300, 5_999], 10)
    fc.check_and_queries(rucene_amd, seg, decode, search, [[0, 1, 2], [3, 40, 7], [2, 2]], 10)
    fc.check_or_queries(rucene_amd, seg, decode, search, [[0, 9, 200], [5, 6, 7, 8, 1000]], 100)


def test_checkers_reject_a_wrong_answer(small):
    rucene_amd, seg, decode, search = small

    def off_by_one(queries, k):
        hits, totals = search(queries, k)
        hits = hits.copy()
        hits["doc"][0, 0] += 1
        return hits, totals

    with pytest.raises(AssertionError):
        fc.check_term_queries(rucene_amd, seg, decode, off_by_one, [3], 10)

    def lost_posting(states):
        d, f = decode(states)
        return d[:-1], f[:-1]

    with pytest.raises(AssertionError):
        fc.check_decode(seg, lost_posting, [0, 1])
