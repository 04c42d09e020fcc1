"""GPU path at BASELINE.json's full size (10 M docs, 1 M terms — far beyond what the per-doc CPU oracle covers in test
time): size-independent properties through the C ABI. Decoded postings against the term table's own invariants
(counts, strictly increasing doc ids, per-term freq checksums, idempotence) and TERM / AND / OR top-k against a numpy
re-derivation from those postings (tests/fullsize_checks.py; the checkers themselves are validated against the
oracle on a small index in tests/test_fullsize_checks_cpu.py)."""
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
    yield rucene_amd, seg, (lambda st: leaf.segment.decode_terms(st)), (lambda qs, k: searcher.search_batch(qs, k))
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
