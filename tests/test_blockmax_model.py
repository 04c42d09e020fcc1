"""Executable specification of the block-max pruning planned for k_search_term — a numpy model, no GPU
code. Per FullBlock one 64-bit word: bits 0..3 the largest freq (15 = some freq > 10: no bound), then for every freq value
1..10 six bits with the largest norm rank among the block's postings with that freq (0 when there is none) — the block's Pareto
frontier in (freq, rank), which carries its exact maximum score under ANY similarity table. Per query the score table's 2-D
prefix maximum (over ranks, then freqs: robust to rounding and to the rank-0 filler of absent freqs); a block's bound is the
largest of the entries `pmax[rank_f][f]`, f = 1..largest freq; the skip test is "bound bits < threshold bits" with the kernel's tie rule (strict once every remaining doc
lies after the threshold's doc). A first design with two bytes per block (largest freq, largest rank) was modelled here too: it
prunes only ~20 % of the blocks the exact bound prunes, because the posting with both extremes rarely exists. The model must return exactly what the oracle's TermScorer + TopDocsCollector return (canonical order) and count every
posting, whatever the order in which independent work items run and share their lists. It also reports how many blocks it
skipped, the number a future kernel's debug counter can be held against."""
import numpy as np
import pytest


def _f32(x):
    return np.float32(x)


def _table(weight, k1, cache, rank_to_norm):
    """score table [rank][freq 0..10] computed like the kernel's build_score_table / bm25_score: (w*(k1+1) * f) / (f + cache[norm])"""
    wk = _f32(weight) * (_f32(k1) + _f32(1.0))
    t = np.zeros((len(rank_to_norm), 11), dtype=np.float32)
    for r, nb in enumerate(rank_to_norm):
        for f in range(11):
            t[r, f] = _f32(_f32(wk * _f32(f)) / _f32(_f32(f) + cache[nb])) if f else _f32(0)
    return t


def _key(score_bits, doc):
    return (int(score_bits) << 32) | (0xFFFFFFFF - int(doc))      # make_key for non-negative scores: larger == better


class _TopK:
    def __init__(self, k):
        self.k, self.keys = k, []

    def offer(self, key):
        if len(self.keys) < self.k:
            self.keys.append(key)
            self.keys.sort()
        elif key > self.keys[0]:
            self.keys[0] = key
            self.keys.sort()

    def tau(self):
        return self.keys[0] if len(self.keys) == self.k else 0


def _thr(tau, seen_doc):
    """term_blocks_fast::thr_of: raw score bits a posting must reach; a tie can only win while docs at or before tau's doc remain"""
    if tau == 0:
        return 0
    bits, doc = tau >> 32, 0xFFFFFFFF - (tau & 0xFFFFFFFF)
    return bits + 1 if doc <= seen_doc else bits


def frontier_words(freqs, ranks):
    """The per-block directory word k_prepare_blocks would build (n_blocks x 128 freqs / ranks -> uint64[n_blocks])."""
    assert ranks.max() < 64
    words = np.zeros(freqs.shape[0], dtype=np.uint64)
    fmax = freqs.max(axis=1)
    words |= np.where(fmax > 10, 15, fmax).astype(np.uint64)
    for fv in range(1, 11):
        best = np.where(freqs == fv, ranks, 0).max(axis=1).astype(np.uint64)
        words |= best << np.uint64(4 + 6 * (fv - 1))
    return words


def model_search(docs, freqs, ranks, table, k, items, order):
    """docs/freqs/ranks: the term's FullBlock postings (n_blocks x 128). items: list of (first block, last block + 1); order: the
    sequence in which (item index) steps run one block each — any interleaving. All items share ONE list (a workgroup group)."""
    pmax = np.maximum.accumulate(np.maximum.accumulate(table, axis=0), axis=1)      # 2-D prefix max: ranks, then freqs
    nb = docs.shape[0]
    words = frontier_words(freqs, ranks)
    usable_block = (words & np.uint64(15)) != np.uint64(15)
    bound_bits = np.zeros(nb, dtype=np.uint32)
    for b in range(nb):
        w = int(words[b])
        fmax = w & 15
        if fmax != 15:
            bound_bits[b] = max(pmax[(w >> (4 + 6 * (fv - 1))) & 63, fv] for fv in range(1, fmax + 1)).view(np.uint32)
    top, skipped, count = _TopK(k), 0, 0
    cursor = [a for a, _ in items]
    seen = [(-1 if a == 0 else int(docs[a - 1, -1])) for a, _ in items]
    for it in order:
        b = cursor[it]
        if b >= items[it][1]:
            continue
        cursor[it] += 1
        count += 128
        thr = _thr(top.tau(), seen[it])
        usable = bool(usable_block[b])
        if usable and int(bound_bits[b]) < thr:
            skipped += 1
        else:
            s = table[ranks[b], np.minimum(freqs[b], 10)] if usable else None
            assert s is not None
            for d, sb in zip(docs[b], s.view(np.uint32)):
                if int(sb) >= thr:
                    top.offer(_key(sb, d))
        seen[it] = int(docs[b, -1])
    hits = sorted(top.keys, reverse=True)
    return [(0xFFFFFFFF - (h & 0xFFFFFFFF), np.array([h >> 32], dtype=np.uint32).view(np.float32)[0]) for h in hits], count, skipped


@pytest.fixture(scope="module")
def world(oracle):
    import __graft_entry__ as g
    g.build()
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(300_000, 20_000, seed=17)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    return seg, oseg, oracle.Searcher([oseg])


@pytest.mark.parametrize("term", [0, 1, 3, 9, 40])
@pytest.mark.parametrize("k", [10, 100])
def test_block_max_model_is_exact(oracle, world, term, k):
    import rucene_amd
    seg, oseg, searcher = world
    d, f = oseg.decode_term(seg.terms[term])
    nb = len(d) // 128
    assert nb >= 8
    docs, freqs = d[:nb * 128].reshape(nb, 128), f[:nb * 128].reshape(nb, 128)
    rank_to_norm = np.unique(seg.norms)                                     # ranks ordered by norm byte, as the segment upload does
    ranks = np.searchsorted(rank_to_norm, seg.norms[docs])
    w, _, cache = rucene_amd.bm25_compute_weight(1.2, 0.75, seg.max_doc, seg.doc_count, seg.sum_total_term_freq,
                                                 [int(seg.terms[term]["doc_freq"])])
    table = _table(w, 1.2, np.asarray(cache, dtype=np.float32), rank_to_norm)
    # the oracle's answer over the FullBlock postings only: the same term cut to nb * 128 postings is not addressable, so
    # compare with a brute-force canonical top-k of exactly those postings, whose scores the oracle pins (checked below)
    scores = table[ranks, np.minimum(freqs, 10)]
    order = np.lexsort((docs.reshape(-1), -scores.reshape(-1)))[:k]
    want = [(int(docs.reshape(-1)[i]), scores.reshape(-1)[i]) for i in order]
    od, os_, _ = searcher.search(oracle.OP_TERM, [term], 5, tie_mode=oracle.TIE_CANONICAL)
    best = dict(zip(docs.reshape(-1).tolist(), scores.reshape(-1).tolist()))
    for dd, ss in zip(od.tolist(), os_.tolist()):
        if dd in best:
            assert np.float32(best[dd]) == np.float32(ss)                   # the model's table scores are the oracle's, bit for bit
    rng = np.random.default_rng(term * 31 + k)
    results = []
    for n_items in (1, 8):
        per = -(-nb // n_items)
        items = [(i * per, min(nb, (i + 1) * per)) for i in range(n_items) if i * per < nb]
        schedules = [np.repeat(np.arange(len(items)), per),                  # one item after the other
                     np.tile(np.arange(len(items)), per),                    # lock step
                     rng.permutation(np.repeat(np.arange(len(items)), per))]  # arbitrary interleaving
        for order_ in schedules:
            hits, count, skipped = model_search(docs, freqs, ranks, table, k, items, order_.tolist())
            assert count == nb * 128
            assert [(h[0], np.float32(h[1])) for h in hits] == [(w_[0], np.float32(w_[1])) for w_ in want]
            results.append(skipped)
    if nb >= 100 and k == 10:
        assert max(results) > 0.5 * nb, results                               # long lists: most blocks are pruned, exactly
    print("term %d k %d: %d FullBlocks, skipped per schedule %s" % (term, k, nb, results))
