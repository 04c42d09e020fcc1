"""Executable specification of MaxScore pruning for the OR path — a numpy model, no GPU code. (The GPU kernel built on the idea is
k_or_lazy, csrc/kernels/search_or_lazy.hpp: its non-essential clauses are the ones with doc bitmaps, and its candidate test uses the
bounds of the clauses a doc is actually in; this model keeps the textbook form.)

  1. theta0 = a LOWER bound of the final k-th best score: the k-th best partial score over the docs of one clause (any k docs'
     full scores are at least their partial scores when weights are non-negative).
  2. Clauses ordered by their score bound (exact list maximum); the non-essential set is the longest prefix whose bounds, summed
     in f32 IN CLAUSE ORDER, stay strictly below theta0. Floating-point addition of non-negative numbers is monotone, so that sum
     bounds the computed score of every doc that occurs in non-essential clauses only: such a doc can never enter the top-k.
  3. Candidates = docs of the essential clauses; each candidate's score is the f32 sum over ALL its clauses in clause order (the
     order DisjunctionSumScorer's SimpleQueue uses below 10 clauses), the non-essential ones reached by probing.
  4. total_hits = size of the union of all clauses: counted exhaustively, without scores.
The model must equal the oracle's DisjunctionSumScorer + TopDocsCollector (canonical order) bit for bit."""
import numpy as np
import pytest

from test_blockmax_model import _table


def model_or(clauses, k):
    """clauses: [(docs int32[], scores f32[])] in clause order -> (top-k [(doc, score)], total_hits, postings scored, postings all)"""
    m = len(clauses)
    ub = [c[1].max() if len(c[1]) else np.float32(0) for c in clauses]
    seed = min(range(m), key=lambda j: len(clauses[j][0]) if len(clauses[j][0]) >= k else 1 << 62)
    theta0 = np.sort(clauses[seed][1])[-k] if len(clauses[seed][0]) >= k else np.float32(0)
    by_bound = sorted(range(m), key=lambda j: (ub[j], j))
    non_essential = set()
    for j in by_bound:
        trial = non_essential | {j}
        acc = np.float32(0)
        for c in range(m):                       # f32, clause order — the order the real sum is formed in
            if c in trial:
                acc = np.float32(acc + ub[c])
        if acc < theta0:
            non_essential = trial
        else:
            break
    essential = [j for j in range(m) if j not in non_essential]
    cand = np.unique(np.concatenate([clauses[j][0] for j in essential])) if essential else np.zeros(0, np.int32)
    total = np.unique(np.concatenate([c[0] for c in clauses])).size
    score = np.zeros(cand.size, dtype=np.float32)
    for c in range(m):                           # clause order
        docs, sc = clauses[c]
        pos = np.searchsorted(docs, cand)
        hit = (pos < docs.size) & (docs[np.minimum(pos, docs.size - 1)] == cand)
        score[hit] = (score[hit] + sc[pos[hit]]).astype(np.float32)
    order = np.lexsort((cand, -score))[:k]
    scored = sum(len(clauses[j][0]) for j in essential)
    return [(int(cand[i]), score[i]) for i in order], total, scored, sum(len(c[0]) for c in clauses)


@pytest.fixture(scope="module")
def world(oracle):
    import __graft_entry__ as g
    g.build()
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(400_000, 30_000, seed=23)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    return seg, oseg, oracle.Searcher([oseg])


def _clause(seg, oseg, term):
    import rucene_amd
    d, f = oseg.decode_term(seg.terms[term])
    rank_to_norm = np.unique(seg.norms)
    w, _, cache = rucene_amd.bm25_compute_weight(1.2, 0.75, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, [int(seg.terms[term]["doc_freq"])])
    table = _table(w, 1.2, np.asarray(cache, dtype=np.float32), rank_to_norm)
    assert f.max() <= 10
    return d.astype(np.int32), table[np.searchsorted(rank_to_norm, seg.norms[d]), f]


@pytest.mark.parametrize("k", [10, 100])
def test_maxscore_model_is_exact(oracle, world, k):
    from rucene_amd import indexgen
    seg, oseg, searcher = world
    rows = indexgen.log_uniform_ranks(9 * 40, 1, 3000, seed=77).reshape(-1, 9) - 1
    saved = []
    for i, row in enumerate(rows):
        terms = [int(t) for t in row[:2 + i % 8]]                  # 2 .. 9 clauses: SimpleQueue (clause-order sums)
        got, total, scored, every = model_or([_clause(seg, oseg, t) for t in terms], k)
        od, os_, ot = searcher.search(oracle.OP_OR, terms, k, tie_mode=oracle.TIE_CANONICAL)
        assert total == ot, terms
        assert [g[0] for g in got] == od.tolist(), terms
        assert np.array([g[1] for g in got], dtype=np.float32).view(np.uint32).tolist() == os_.view(np.uint32).tolist(), terms
        saved.append(1.0 - scored / every)
    print("k=%d: postings spared from scoring: median %.0f%%, mean %.0f%%" % (k, 100 * np.median(saved), 100 * np.mean(saved)))
    assert np.mean(saved) > 0.3
