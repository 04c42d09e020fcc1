"""GPU path at BASELINE.json's full size (10 M docs, 1 M terms): (1) the ORACLE itself on sampled subsets of the bench's
own query batches — the multithreaded oracle does 1024 single-term queries in a fraction of a second, 64 conjunctions /
16 ten-clause disjunctions in a few seconds — and (2) size-independent properties through the C ABI: decoded postings
against the term table's own invariants (counts, strictly increasing doc ids, per-term freq checksums, idempotence) and
TERM / AND / OR top-k against a numpy re-derivation from those postings (tests/fullsize_checks.py; the checkers
themselves are validated against the oracle on a small index in tests/test_fullsize_checks_cpu.py)."""
import os
import numpy as np
import pytest

import fullsize_checks as fc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    import rucene_amd
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(10_000_000, 1_000_000)
    ctx = rucene_amd.Context()
    leaf = rucene_amd.LeafReader.from_synthetic(seg)
    searcher = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
    search = lambda qs, k: searcher.search_batch(qs, k)
    search.ctx = ctx
    yield rucene_amd, seg, (lambda st: leaf.segment.decode_terms(st)), search
    ctx.close()


def test_every_block_decoded_term_keeps_its_invariants(full):
    rucene_amd, seg, decode, search = full
    big = np.nonzero(seg.terms["doc_freq"] >= 128)[0]
    docs, freqs, starts, df = fc.check_decode(seg, decode, big)
    assert docs.size > 15_000_000                                      # ~19 M postings live in FullBlock terms


def test_single_term_topk_against_numpy(full):
    rucene_amd, seg, decode, search = full
    fc.check_term_queries(rucene_amd, seg, decode, search, [0, 1, 2, 5, 17, 100, 999, 9_999, 123_456, 999_999], 10)
    fc.check_term_queries(rucene_amd, seg, decode, search, [0, 3, 250, 40_000], 100)


def test_conjunction_topk_against_numpy(full):
    rucene_amd, seg, decode, search = full
    fc.check_and_queries(rucene_amd, seg, decode, search, [[0, 1, 2], [3, 10, 50], [7, 100, 900], [20, 21], [0, 5_000, 90_000]], 10)


def test_disjunction_topk_against_numpy(full):
    rucene_amd, seg, decode, search = full
    fc.check_or_queries(rucene_amd, seg, decode, search, [[0, 5], [1, 30, 400, 5_000, 70_000], [2, 3, 4, 6, 8, 9, 11, 13, 15]], 100)


# ---- the oracle at full size, on samples of bench.py's batches (same seeds, same rank distributions) --------------------
def _bench_queries(kind, n):
    import bench
    return bench.build_queries(1024, kind, bench.SEED_QUERIES)[:n]


def _against_oracle(full, oracle, kind, n, k):
    rucene_amd, seg, decode, search = full
    tids = _bench_queries(kind, n)
    T, B = rucene_amd.TermQuery, rucene_amd.BooleanQuery
    if kind == "term":
        qs, op = [T(int(t[0])) for t in tids], oracle.OP_TERM
    elif kind == "and3":
        qs, op = [B.build([T(int(x)) for x in t], []) for t in tids], oracle.OP_AND
    else:
        qs, op = [B.build([], [T(int(x)) for x in t]) for t in tids], oracle.OP_OR
    hits, totals = search(qs, k)
    osearcher = oracle.Searcher([oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)])
    ops = np.full(len(qs), op, np.int32)
    offs = (np.arange(len(qs) + 1) * tids.shape[1]).astype(np.int32)
    cd, cs, cc, ct, _, _ = osearcher.search_batch(ops, offs, np.ascontiguousarray(tids).reshape(-1), k, tie_mode=oracle.TIE_CANONICAL,
                                                  threads=os.cpu_count() or 1)
    assert (totals == ct).all()
    if kind == "or10":  # >= 10 clauses: the reference sums in heap order -> scores within 1e-5 relative (north_star's tolerance),
        from oracle import parity  # doc ids judged by the oracle's own score of every returned doc (oracle/parity.py)
        differing = parity.check_heap_order_batch(osearcher, op, tids, hits, totals, cd, cs, cc, ct, rtol=1e-5, what="or10")
        np.testing.assert_allclose(hits["score"], cs, rtol=1e-5, atol=0)
        print("or10 at full size: %d of %d returned docs differ from the oracle's rows (tie band only)" % (differing, int(cc.sum())))
    else:
        assert (hits["doc"] == cd).all()
        assert (hits["score"].view(np.int32) == cs.view(np.int32)).all()


def test_single_term_batch_equals_the_oracle_at_full_size(full, oracle):
    _against_oracle(full, oracle, "term", 1024, 10)


def test_pruned_term_batch_decodes_a_fraction_of_what_it_covers(full):
    """rgpu_last_search_counters at BASELINE size: the block-max frontier lets k_search_term leave most FullBlocks packed
    (bench.py's postings_decoded_per_sec is this figure, not the postings the queries cover)."""
    rucene_amd, seg, decode, search = full
    tids = _bench_queries("term", 1024)
    search([rucene_amd.TermQuery(int(t[0])) for t in tids], 10)
    c = search.ctx.last_search_counters()
    full_blocks = int((seg.terms["doc_freq"][tids.reshape(-1)] // 128).sum())
    assert c["postings_covered"] == int(seg.terms["doc_freq"][tids.reshape(-1)].sum())
    assert 0 < c["blocks_decoded"] < full_blocks // 5 and c["postings_decoded"] < c["postings_covered"] // 5


def test_conjunction_batch_equals_the_oracle_at_full_size(full, oracle):
    """All 1024 queries of bench.py's and3 batch (sampled parity is how round 4's phrase bench missed 568 short
    answers): the batched first probe, the survivor queue and the block-by-block path are all inside this batch."""
    _against_oracle(full, oracle, "and3", 1024, 10)


def test_disjunction_batch_matches_the_oracle_at_full_size(full, oracle):
    # ALL 1024 ten-clause disjunctions of bench.py's or10 batch (round 5 compared 256: VERDICT r5 item 8; the oracle walks ~2.7 M
    # postings per query — about ten seconds of the box's cores), under the TIE-BAND RULE of oracle/parity.py — never "bit-exact"
    _against_oracle(full, oracle, "or10", 1024, 100)


# ---- phrases at full size: the whole batch of bench.py's configs.positions.phrase2 ---------------------------------------------------
def test_phrase_batch_equals_the_oracle_at_full_size(oracle):
    """1024 two-term exact phrases (bench.py's own batch: ranks log-uniform 1..1000, ~100 M candidate slots) + sloppy ones behind them,
    over the 10 M-doc corpus indexed with positions, EVERY query against the oracle's ExactPhraseScorer / SloppyPhraseScorer: hit
    counts, doc ids, score bits. The size matters: a grid is described to the hardware in 32-bit work-item counts, and a kernel
    that takes one wavefront per candidate slot runs out of them at 67 M slots — round 4 found the exact-phrase kernel silently
    skipping everything behind that (568 of these 1024 queries answered short) while the first 96, all the bench compared, were fine."""
    import rucene_amd
    from rucene_amd import indexgen
    import bench
    seg = indexgen.build_zipf(10_000_000, 1_000_000, positions=True)
    ctx = rucene_amd.Context()
    try:
        leaf = rucene_amd.LeafReader.from_synthetic_positions(seg)
        searcher = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
        ranks = indexgen.log_uniform_ranks(2 * 1024, 1, 1000, bench.SEED_QUERIES ^ 0xF2).reshape(-1, 2) - 1
        # the sloppy queries come last: their slots lie behind the 2^32 / 64 mark of the batch
        sloppy = [(int(a), int(b), 1 + i % 3) for i, (a, b) in enumerate(ranks[600:664])]
        queries = [rucene_amd.PhraseQuery([int(a), int(b)]) for a, b in ranks] + [rucene_amd.PhraseQuery([a, b], slop=sl) for a, b, sl in sloppy]
        k = 10
        hits, totals = searcher.search_phrase_batch(queries, k)
        ix = oracle.PositionsIndex.from_files(seg.doc_bytes, seg.pos_bytes, seg.terms, leaf.term_positions)
        try:
            for i, q in enumerate(queries):
                d, s, total = ix.phrase_search(q.terms, k, seg.norms, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, slop=q.slop)
                assert totals[i] == total, (i, q.terms, q.slop, int(totals[i]), total)
                assert (hits[i]["doc"][:d.size] == d).all() and (hits[i]["doc"][d.size:] == -1).all(), (i, q.terms, q.slop)
                assert (hits[i]["score"][:d.size].view(np.int32) == s.view(np.int32)).all(), (i, q.terms, q.slop)
        finally:
            ix.close()
        assert int(totals[:1024].sum()) > 1_000_000
    finally:
        ctx.close()
