"""BASELINE.json `configs`, case by case, and where each one is pinned:

  configs[0]  single-term BM25 query, 100 k-doc synthetic index, Rucene's CPU IndexSearcher (plumbing, no GPU)  -> HERE (the oracle
              is the CPU IndexSearcher: its result against an independent numpy restatement of TermScorer + BM25 + the collector,
              and — `-m gpu` — the same query through the C ABI, bit-exact)
  configs[1]  1024 single-term queries, 10 M docs   -> tests/test_gpu_fullsize.py (the whole batch against the oracle) + bench.py's headline
  configs[2]  3-term AND, 10 M docs                 -> tests/test_gpu_fullsize.py (all 1024 queries) + bench.py `configs.and3` (all 1024)
  configs[3]  10-term OR top-100, 10 M docs         -> tests/test_gpu_fullsize.py (256 queries, heap-order rule) + bench.py `configs.or10`
  configs[4]  100 M docs in 8 shards, 3-term AND    -> bench.py --gpus 8 (`configs.and3` on every rank's shard); the record /
              merge half on one GPU: tests/test_gpu_parity.py::test_shard_records_merge_like_finish_parallel; world size 2 on CPU:
              tests/test_dist_gloo.py
"""
import math

import numpy as np
import pytest

N0, VOCAB0 = 100_000, 20_000   # config 0 — N = 100 000 docs, one field, Zipf df(r) = 0.2 N / r


def _byte315_to_float(b):
    """SmallFloat::byte315_to_float (util/small_float.rs:28-36)."""
    if b == 0:
        return np.float32(0.0)
    bits = (int(b) & 0xFF) << (24 - 3)
    bits += (63 - 15) << 24
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def _numpy_term_search(docs, freqs, norms, max_doc, doc_count, sum_ttf, k, k1=np.float32(1.2), b=np.float32(0.75)):
    """TermScorer + BM25Similarity + TopDocsCollector, restated independently of the oracle's C++ in numpy f32 / f64:
    bm25_similarity.rs:99-114 (idf in f64 -> f32), :72-83 (avgdl), :33-43 (NORM_TABLE), :158-165 (cache), :203-212 (score),
    collector/top_docs.rs (score desc, doc asc — the canonical tie rule)."""
    df = docs.size
    idf = np.float32(math.log(1.0 + (float(doc_count) - float(df) + 0.5) / (float(df) + 0.5)))
    avgdl = np.float32(float(sum_ttf) / float(doc_count))
    table = np.zeros(256, dtype=np.float32)
    for i in range(1, 256):
        f = _byte315_to_float(i)
        table[i] = np.float32(1.0) / (f * f)
    table[0] = np.float32(1.0) / table[255]
    cache = (k1 * ((np.float32(1.0) - b) + b * (table / avgdl))).astype(np.float32)
    weight = idf * np.float32(1.0)
    wk = np.float32(weight * (k1 + np.float32(1.0)))
    fr = freqs.astype(np.float32)
    scores = ((wk * fr).astype(np.float32) / (fr + cache[norms[docs]]).astype(np.float32)).astype(np.float32)
    order = np.lexsort((docs, -scores.astype(np.float64)))[:k]
    return docs[order], scores[order], df


@pytest.fixture(scope="module")
def config0(oracle):
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(N0, VOCAB0)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    return seg, oseg, oracle.Searcher([oseg])


@pytest.mark.parametrize("rank", [1, 7, 100, 3000])
def test_config0_single_term_query_on_the_cpu_searcher(oracle, config0, rank):
    seg, oseg, osearcher = config0
    tid = rank - 1
    docs, freqs = oseg.decode_term(seg.terms[tid])
    target = max(1, min(N0 // 2, round(0.2 * N0 / rank)))   # gaps are Geometric(df / N), the list is cut at N: about `target` postings
    assert docs.size == seg.terms[tid]["doc_freq"] and 0.9 * target - 3 <= docs.size <= target
    for k in (10, 100):
        want_d, want_s, total = _numpy_term_search(docs, freqs, seg.norms, N0, seg.doc_count, seg.sum_total_term_freq, k)
        d, s, t = osearcher.search(oracle.OP_TERM, [tid], k, tie_mode=oracle.TIE_CANONICAL)
        assert t == total
        assert (d == want_d).all()
        assert (s.view(np.int32) == want_s.view(np.int32)).all()   # the same f32 operations in the same order
        # the reference's own heap (std BinaryHeap under ScoreDoc's reversed order): same scores, same docs above the k-th score
        hd, hs, ht = osearcher.search(oracle.OP_TERM, [tid], k, tie_mode=oracle.TIE_RUST_HEAP)
        assert ht == total and (np.sort(hs) == np.sort(s)).all()
        assert set(hd[hs > s[-1]]) == set(d[s > s[-1]])


@pytest.mark.gpu
def test_config0_on_the_gpu(oracle, config0):
    import rucene_amd
    seg, oseg, osearcher = config0
    ctx = rucene_amd.Context()
    try:
        searcher = rucene_amd.GpuIndexSearcher([rucene_amd.LeafReader.from_synthetic(seg)], ctx=ctx)
        tids = [0, 6, 99, 2999, VOCAB0 - 1]
        for k in (10, 100):
            hits, totals = searcher.search_batch([rucene_amd.TermQuery(t) for t in tids], k)
            for i, t in enumerate(tids):
                d, s, total = osearcher.search(oracle.OP_TERM, [t], k, tie_mode=oracle.TIE_CANONICAL)
                assert totals[i] == total
                assert (hits[i]["doc"][:d.size] == d).all() and (hits[i]["doc"][d.size:] == -1).all()
                assert (hits[i]["score"][:d.size].view(np.int32) == s.view(np.int32)).all()
    finally:
        ctx.close()
