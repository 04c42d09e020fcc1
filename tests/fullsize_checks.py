"""Size-independent property checks for the full BASELINE-size index (10 M docs): nothing here calls the oracle —
decoded postings are checked against the term table's own invariants, and search results against a numpy
re-derivation of BM25 / intersection / union from the decoded postings (f32 elementwise arithmetic in numpy is IEEE
correctly rounded, i.e. bit-identical to the reference's operation order when written in the same order).
`decode(states) -> (docs, freqs)` and `search(queries, k) -> (hits, totals)` are whatever is under test."""
import numpy as np

K1, B = 1.2, 0.75


def check_decode(seg, decode, term_ids):
    st = seg.terms[np.asarray(term_ids)]
    docs, freqs = decode(st)
    df = st["doc_freq"].astype(np.int64)
    assert docs.size == int(df.sum()) == freqs.size
    starts = np.concatenate([[0], np.cumsum(df)[:-1]])
    assert (docs >= 0).all() and (docs < seg.max_doc).all()
    assert (freqs >= 1).all() and (freqs <= 10).all()                  # write-time clamp (codec/postings/mod.rs:82)
    inc = np.diff(docs.astype(np.int64)) > 0
    boundary = np.zeros(docs.size - 1, dtype=bool)
    boundary[(starts[1:] - 1)[starts[1:] - 1 < docs.size - 1]] = True  # a new term may restart at a lower doc id
    assert (inc | boundary).all(), "doc ids must increase strictly inside a term"
    assert (np.add.reduceat(freqs.astype(np.int64), starts) == st["total_term_freq"]).all()   # checksum per term
    d2, f2 = decode(st)
    assert (d2 == docs).all() and (f2 == freqs).all()                  # idempotent
    return docs, freqs, starts, df


def _weights(rucene_amd, seg, term):
    w, _idf, cache = rucene_amd.bm25_compute_weight(K1, B, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, [int(seg.terms[term]["doc_freq"])])
    return np.float32(w), cache.astype(np.float32)


def _scores(rucene_amd, seg, term, docs, freqs):
    w, cache = _weights(rucene_amd, seg, term)
    wk = np.float32(w * np.float32(np.float32(K1) + np.float32(1.0)))   # weight * (k1 + 1), formed in f32 first
    f = freqs.astype(np.float32)
    return (wk * f) / (f + cache[seg.norms[docs]])                      # bm25_similarity.rs:203-212, left to right


def _topk(docs, scores, k):
    order = np.lexsort((docs, -scores.astype(np.float64)))[:k]          # score desc, then doc asc (canonical tie rule)
    return docs[order], scores[order]


def _same(hits_row, total, want_docs, want_scores, want_total, what):
    n = want_docs.size
    assert int(total) == int(want_total), (what, int(total), int(want_total))
    assert (hits_row["doc"][:n] == want_docs).all(), (what, hits_row["doc"][:n], want_docs)
    assert (hits_row["score"][:n].view(np.int32) == want_scores.astype(np.float32).view(np.int32)).all(), what
    assert (hits_row["doc"][n:] == -1).all(), what
    s = hits_row["score"][:n]
    assert (np.diff(s) <= 0).all()                                      # sorted
    tie = np.diff(s) == 0
    assert (np.diff(hits_row["doc"][:n])[tie] > 0).all()                # ties by doc id ascending


def check_term_queries(rucene_amd, seg, decode, search, terms, k):
    T = rucene_amd.TermQuery
    hits, totals = search([T(int(t)) for t in terms], k)
    for i, t in enumerate(terms):
        docs, freqs = decode(seg.terms[[int(t)]])
        wd, ws = _topk(docs, _scores(rucene_amd, seg, int(t), docs, freqs), k)
        _same(hits[i], totals[i], wd, ws, docs.size, ("TERM", int(t)))


def check_and_queries(rucene_amd, seg, decode, search, term_rows, k):
    T, Bq = rucene_amd.TermQuery, rucene_amd.BooleanQuery
    hits, totals = search([Bq.build([T(int(t)) for t in row], []) for row in term_rows], k)
    for i, row in enumerate(term_rows):
        order = sorted(range(len(row)), key=lambda j: int(seg.terms[int(row[j])]["doc_freq"]))   # stable cost sort
        lists = [decode(seg.terms[[int(row[j])]]) for j in order]
        common = lists[0][0]
        for d, _ in lists[1:]:
            common = np.intersect1d(common, d, assume_unique=True)
        total = np.zeros(common.size, np.float32)
        for n, (j, (d, f)) in enumerate(zip(order, lists)):
            s = _scores(rucene_amd, seg, int(row[j]), d, f)[np.searchsorted(d, common)]
            total = s if n == 0 else (total + s).astype(np.float32)     # lead1, lead2, others... (conjunction_scorer.rs:87-95)
        wd, ws = _topk(common, total, k)
        _same(hits[i], totals[i], wd, ws, common.size, ("AND", list(map(int, row))))


def check_or_queries(rucene_amd, seg, decode, search, term_rows, k):
    T, Bq = rucene_amd.TermQuery, rucene_amd.BooleanQuery
    assert all(len(r) < 10 for r in term_rows), "clause-order summation only holds below 10 clauses"
    hits, totals = search([Bq.build([], [T(int(t)) for t in row]) for row in term_rows], k)
    for i, row in enumerate(term_rows):
        lists = [decode(seg.terms[[int(t)]]) for t in row]
        union = np.unique(np.concatenate([d for d, _ in lists]))
        acc = np.zeros(union.size, np.float32)
        for t, (d, f) in zip(row, lists):                               # child order == SimpleQueue order
            idx = np.searchsorted(union, d)
            acc[idx] = (acc[idx] + _scores(rucene_amd, seg, int(t), d, f)).astype(np.float32)
        wd, ws = _topk(union, acc, k)
        _same(hits[i], totals[i], wd, ws, union.size, ("OR", list(map(int, row))))
