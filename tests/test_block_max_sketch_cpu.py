"""The claim behind the block-max sketches (rucene_amd/csrc/kernels/search_term.hpp: k_term_sketch, sketch_floor), restated on the CPU
with the oracle: of every 128-posting block of a term take the posting with the block's largest freq and, among those, the largest
norm byte — what the directory's frontier word names. Those are real postings of different blocks, so the k-th best of THEIR scores
can never exceed the query's k-th best score, under any similarity; and for a long Zipfian list it is close to it — close enough that
few blocks hold a posting at or above it."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def corpus(oracle):
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(1_200_000, 50_000)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    return seg, oseg


@pytest.mark.parametrize("k1,b", [(1.2, 0.75), (2.0, 0.3)])
def test_kth_best_block_champion_is_a_valid_and_tight_threshold(corpus, oracle, k1, b):
    seg, oseg = corpus
    osearcher = oracle.Searcher([oseg], k1=k1, b=b)
    norms = np.asarray(seg.norms)
    long_terms = [t for t in range(1, 30) if int(seg.terms["doc_freq"][t]) >= 64 * 128]
    assert len(long_terms) >= 10
    for t in long_terms[:: 3]:
        docs, freqs = oseg.decode_term(seg.terms[t])
        nblocks = docs.size // 128
        d = docs[: nblocks * 128].reshape(nblocks, 128)
        f = freqs[: nblocks * 128].reshape(nblocks, 128).astype(np.int64)
        nb = norms[d].astype(np.int64)
        # per block: largest freq, then the largest norm byte among the postings with that freq (a larger byte = a shorter doc)
        key = f * 256 + nb
        pick = key.argmax(axis=1)
        champions = d[np.arange(nblocks), pick]
        champ_scores = np.sort(osearcher.score_docs(oracle.OP_TERM, [t], champions)[0])[::-1]
        per_posting = osearcher.score_docs(oracle.OP_TERM, [t], docs)[0]
        all_scores = np.sort(per_posting)[::-1]
        for k in (1, 10, 64, 128):
            if k > nblocks:
                continue   # (fewer blocks than k: no threshold from the sketch)
            threshold = champ_scores[k - 1]
            kth_best = all_scores[k - 1]
            assert threshold <= kth_best, (t, k, threshold, kth_best)        # valid: k real postings of k blocks reach it
            if k == 10:
                # tight: the blocks that hold a posting at or above the threshold are a small part of the list
                block_best = per_posting[: nblocks * 128].reshape(nblocks, 128).max(axis=1)
                at_or_above = int((block_best >= threshold).sum())
                assert at_or_above <= max(64, nblocks // 8), (t, at_or_above, nblocks)
