"""SloppyPhraseScorer as restated in oracle/sloppy_phrase.hpp (PhraseQuery with slop > 0).

Pinned by the one test the reference holds for phrases (query/phrase_query.rs:511-637): three documents, the phrase "quick
fox" with slop 0 / 1 / 2 / 3 -> total_hits 1 / 2 / 2 / 3. Everything finer is parity-unpinned (the source text is the only
authority), so the C++ restatement is held against a SECOND, independent restatement of the same source lines written
here in Python (its own emulation of Rust's BinaryHeap included): two transcriptions must agree on every doc's sloppy
frequency, with and without repeated terms."""
import numpy as np
import pytest

F32 = np.float32


# ---- the reference's algorithm once more, in Python (scorer/phrase_scorer.rs:537-871; util/external/binary_heap.rs:121-210) ----
class _PP:
    def __init__(self, positions, offset, ord_, term):
        self.positions, self.offset, self.ord, self.term = positions, offset, ord_, term
        self.position = self.count = 0
        self.at = 0
        self.rpt_group, self.rpt_ind = -1, 0

    def first_position(self):
        self.count, self.at = len(self.positions), 0
        self.next_position()

    def next_position(self):
        if self.count > 0:
            self.count -= 1
            self.position = self.positions[self.at] - self.offset
            self.at += 1
            return True
        return False

    def key(self):
        return (self.position, self.offset, self.ord)


class _Sloppy:
    """One instance per (query, leaf): init_first_time state persists from doc to doc, as in the reference."""

    def __init__(self, offsets, terms, slop):
        self.offsets, self.terms, self.slop = offsets, terms, slop
        self.checked_rpts = self.has_rpts = False
        self.rpt_group = []
        self.group_of = [-1] * len(terms)
        self.ind_of = [0] * len(terms)

    # std BinaryHeap with PPElement's reversed order: a <= b  <=>  key(a) >= key(b)
    def _le(self, a, b):
        return self.pps[a].key() >= self.pps[b].key()

    def _sift_up(self, start, pos):
        elt = self.pq[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if self._le(elt, self.pq[parent]):
                break
            self.pq[pos] = self.pq[parent]
            pos = parent
        self.pq[pos] = elt

    def _push(self, i):
        self.pq.append(i)
        self._sift_up(0, len(self.pq) - 1)

    def _pop(self):
        item = self.pq.pop()
        if self.pq:
            item, self.pq[0] = self.pq[0], item
            pos, end = 0, len(self.pq)
            elt = self.pq[0]
            child = 1
            while child < end:
                right = child + 1
                if right < end and self._le(self.pq[child], self.pq[right]):   # !(child > right)
                    child = right
                self.pq[pos] = self.pq[child]
                pos = child
                child = 2 * pos + 1
            self.pq[pos] = elt
            self._sift_up(0, pos)
        return item

    def _advance_pp(self, i):
        if not self.pps[i].next_position():
            return False
        self.end = max(self.end, self.pps[i].position)
        return True

    def _tp(self, i):
        return self.pps[i].position + self.pps[i].offset

    def _collide(self, i):
        for j in self.rpt_group[self.pps[i].rpt_group]:
            if j != i and self._tp(j) == self._tp(i):
                return self.pps[j].rpt_ind
        return -1

    def _lesser(self, a, b):
        x, y = self.pps[a], self.pps[b]
        return a if (x.position < y.position or (x.position == y.position and x.offset < y.offset)) else b

    def _advance_rpts(self, pp):
        if self.pps[pp].rpt_group < 0:
            return True
        rg = self.rpt_group[self.pps[pp].rpt_group]
        bits = set()
        k0, cur = self.pps[pp].rpt_ind, pp
        while True:
            k = self._collide(cur)
            if k < 0:
                break
            cur = self._lesser(cur, rg[k])
            if not self._advance_pp(cur):
                return False
            if k != k0:
                bits.add(k)
        stack = []
        while bits:
            p2 = self._pop()
            stack.append(p2)
            q = self.pps[p2]
            if q.rpt_group >= 0 and q.rpt_ind < len(rg) and q.rpt_ind in bits:
                bits.discard(q.rpt_ind)
        for p2 in reversed(stack):
            self._push(p2)
        return True

    def _fill_queue(self):
        self.pq = []
        for i, p in enumerate(self.pps):
            self.end = max(self.end, p.position)
            self._push(i)

    def _advance_repeat_groups(self):
        for rg in self.rpt_group:
            for j in range(1, len(rg)):
                for _ in range(j):
                    if not self.pps[rg[j]].next_position():
                        return False
        return True

    def _init(self):
        self.end = -2 ** 31
        if not self.checked_rpts:
            self.checked_rpts = True
            for p in self.pps:
                p.first_position()
            cnt, rpt = {}, {}
            for t in self.terms:
                cnt[t] = cnt.get(t, 0) + 1
                if cnt[t] == 2:
                    rpt[t] = len(rpt)
            self.has_rpts = bool(rpt)
            if self.has_rpts:
                rpp = [i for i, t in enumerate(self.terms) if t in rpt]
                res = []
                for a, i1 in enumerate(rpp):
                    if self.pps[i1].rpt_group >= 0:
                        continue
                    for i2 in rpp[a + 1:]:
                        if self.pps[i2].rpt_group >= 0 or self.pps[i2].offset == self.pps[i1].offset or self._tp(i2) != self._tp(i1):
                            continue
                        if self.pps[i1].rpt_group < 0:
                            self.pps[i1].rpt_group = len(res)
                            res.append([i1])
                        self.pps[i2].rpt_group = self.pps[i1].rpt_group
                        res[self.pps[i1].rpt_group].append(i2)
                for rg in res:
                    rg.sort(key=lambda i: self.pps[i].offset)
                    for j, i in enumerate(rg):
                        self.pps[i].rpt_ind = j
                    self.rpt_group.append(rg)
                self.group_of = [p.rpt_group for p in self.pps]
                self.ind_of = [p.rpt_ind for p in self.pps]
                if not self._advance_repeat_groups():
                    return False
            self._fill_queue()
            return True
        for p in self.pps:
            p.first_position()
        if not self.has_rpts:
            self.pq = []
            for i, p in enumerate(self.pps):
                self.end = max(self.end, p.position)
                self._push(i)
            return True
        if not self._advance_repeat_groups():
            return False
        self._fill_queue()
        return True

    def phrase_freq(self, doc_positions):
        """doc_positions[i]: ascending positions of term i in the candidate doc."""
        self.pps = [_PP(ps, o, i, t) for i, (ps, o, t) in enumerate(zip(doc_positions, self.offsets, self.terms))]
        for p, g, k in zip(self.pps, self.group_of, self.ind_of):
            p.rpt_group, p.rpt_ind = g, k
        if not self._init():
            return F32(0)
        freq = F32(0)
        pp = self._pop()
        match_length = self.end - self.pps[pp].position
        nxt = self.pps[self.pq[0]].position
        while self._advance_pp(pp):
            if self.has_rpts and not self._advance_rpts(pp):
                break
            if self.pps[pp].position > nxt:
                if match_length <= self.slop:
                    freq = F32(freq + F32(1.0) / F32(F32(match_length) + F32(1.0)))
                self._push(pp)
                pp = self._pop()
                nxt = self.pps[self.pq[0]].position
                match_length = self.end - self.pps[pp].position
            else:
                match_length = min(match_length, self.end - self.pps[pp].position)
        if match_length <= self.slop:
            freq = F32(freq + F32(1.0) / F32(F32(match_length) + F32(1.0)))
        return freq


def _python_sloppy(postings, term_ids, offsets, slop):
    """[(doc, sloppy freq)] for the docs that hold every term, in doc order, skipping freq <= f32::EPSILON."""
    by_term = [dict(postings[t]) for t in term_ids]
    common = sorted(set.intersection(*[set(b) for b in by_term]))
    sc = _Sloppy(list(offsets), list(term_ids), slop)
    out = []
    for d in common:
        f = sc.phrase_freq([b[d] for b in by_term])
        if f > np.finfo(np.float32).eps:
            out.append((d, f))
    return out


# ---- the reference's own test ------------------------------------------------------------------------------------------------
def test_reference_phrase_query_test_quick_fox(oracle):
    """query/phrase_query.rs:511-637."""
    texts = ["The quick brown fox jumps over a lazy dog", "The quick fox jumps over a lazy dog", "The fox jumps quick over a lazy dog"]
    vocab = sorted({w.lower() for t in texts for w in t.split()})
    postings = [[] for _ in vocab]
    for d, t in enumerate(texts):
        words = [w.lower() for w in t.split()]
        for v, w in enumerate(vocab):
            ps = [i for i, x in enumerate(words) if x == w]
            if ps:
                postings[v].append((d, ps))
    ix = oracle.PositionsIndex(3, postings)
    norms = np.array([oracle.lib().orc_bm25_encode_norm(1.0, len(t.split())) for t in texts], dtype=np.uint8)
    q = [vocab.index("quick"), vocab.index("fox")]
    for slop, want in ((0, 1), (1, 2), (2, 2), (3, 3)):
        docs, scores, total = ix.phrase_search(q, 10, norms, 3, 3, sum(len(t.split()) for t in texts), slop=slop)
        assert total == want == docs.size, (slop, total)
    # and the frequencies behind them: doc 1 exact (distance 0), doc 0 one word apart (1), doc 2 reversed (3)
    d, f = ix.sloppy_freqs(q, 3)
    assert d.tolist() == [0, 1, 2] and f.tolist() == [F32(0.5), F32(1.0), F32(0.25)]
    ix.close()


# ---- two transcriptions of the same source -------------------------------------------------------------------------------------
def _random_positions_index(rng, max_doc, n_terms, density, max_freq, span):
    postings = []
    for _ in range(n_terms):
        docs = np.nonzero(rng.random(max_doc) < density)[0].tolist()
        plist = []
        for d in docs:
            f = int(rng.integers(1, max_freq + 1))
            plist.append((d, np.sort(rng.choice(span, size=min(f, span), replace=False)).tolist()))
        postings.append(plist)
    return postings


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_cpp_and_python_restatements_agree(oracle, seed):
    rng = np.random.default_rng(seed)
    max_doc = 3000
    postings = _random_positions_index(rng, max_doc, 6, 0.35, 6, 40)
    ix = oracle.PositionsIndex(max_doc, postings)
    phrases = [([0, 1], None), ([2, 3, 4], None), ([0, 1, 2, 3], None), ([5, 4], [0, 3]), ([1, 3, 5], [0, 2, 3]),
               # repeated terms: "a b a", "a a", "a b a b", "a a a", a gap between the repeats
               ([0, 1, 0], None), ([2, 2], None), ([0, 1, 0, 1], None), ([3, 3, 3], None), ([4, 5, 4], [0, 1, 4]), ([1, 0, 1, 2, 1], None)]
    checked = matched = 0
    for tids, offs in phrases:
        offs = list(range(len(tids))) if offs is None else offs
        for slop in (1, 2, 5, 12):
            d, f = ix.sloppy_freqs(tids, slop, offsets=offs)
            want = _python_sloppy(postings, tids, offs, slop)
            assert d.tolist() == [x for x, _ in want], (tids, offs, slop)
            assert f.view(np.int32).tolist() == np.array([x for _, x in want], dtype=np.float32).view(np.int32).tolist(), (tids, offs, slop)
            checked += 1
            matched += d.size
    assert checked == len(phrases) * 4 and matched > 2000
    ix.close()


def test_sloppy_with_slop_zero_finds_the_exact_scorers_docs(oracle):
    """Not the reference's dispatch (slop 0 goes to ExactPhraseScorer), but a property worth having: on phrases without repeated
    terms the sloppy scorer run with slop 0 matches exactly the docs the exact scorer matches."""
    rng = np.random.default_rng(7)
    max_doc = 4000
    postings = _random_positions_index(rng, max_doc, 5, 0.4, 5, 25)
    ix = oracle.PositionsIndex(max_doc, postings)
    for tids in ([0, 1], [1, 2, 3], [4, 0], [0, 2, 4, 1]):
        exact = [d for d, _ in ix.phrase_freqs(tids)]
        d, f = ix.sloppy_freqs(tids, 0)
        assert d.tolist() == exact and d.size > 0
    ix.close()


# ---- BulkScorer's two-phase loop around the sloppy scorer (bulk_scorer.rs:91-113) ----------------------------------------------
def _python_two_phase(postings, term_ids, offsets, slop, live, next_limit):
    """score_range_in_docs_set, two-phase arm: the live-docs test comes before matches() (so the scorer's first-doc
    initialisation happens on the first LIVE candidate), every approximation counts towards `next`, and the leaf is abandoned
    once more than next_limit approximations went by without a collected doc."""
    by_term = [dict(postings[t]) for t in term_ids]
    common = sorted(set.intersection(*[set(b) for b in by_term]))
    sc = _Sloppy(list(offsets), list(term_ids), slop)
    out, nxt = [], 0
    for d in common:
        if live is None or live[d]:
            f = sc.phrase_freq([b[d] for b in by_term])
            if f > np.finfo(np.float32).eps:
                out.append((d, f))
        nxt += 1
        if not out and nxt > next_limit:
            break
    return out


def _live_words(live):
    words = np.zeros((live.size + 63) // 64, dtype=np.uint64)
    for d in np.nonzero(live)[0]:
        words[d >> 6] |= np.uint64(1) << np.uint64(d & 63)
    return words


@pytest.mark.parametrize("seed", [11, 12])
def test_two_phase_loop_live_docs_and_next_limit(oracle, seed):
    """Sloppy phrases go through BulkScorer's two-phase arm: deleted docs are never handed to matches(), and a leaf whose first
    next_limit + 1 conjunction matches produce no collected doc yields nothing at all (DEFAULT_DISMATCH_NEXT_LIMIT,
    searcher.rs:47). The exact scorer is not two-phase in the reference: no limit applies to it."""
    rng = np.random.default_rng(seed)
    max_doc = 2500
    postings = _random_positions_index(rng, max_doc, 5, 0.5, 5, 60)
    # a stretch at the front where terms 0 and 1 sit far apart: conjunction matches, no phrase match for a small slop
    for t, at in ((0, 0), (1, 50)):
        postings[t] = [(d, [at]) if d < 400 else (d, ps) for d, ps in postings[t]]
    ix = oracle.PositionsIndex(max_doc, postings)
    norms = rng.integers(95, 125, size=max_doc).astype(np.uint8)
    live = rng.random(max_doc) < 0.8
    words = _live_words(live)
    checked = cut = 0
    for tids, offs in (([0, 1], None), ([1, 0], None), ([0, 1, 2], None), ([0, 1, 0], None), ([3, 4], None), ([2, 2], None)):
        offs = list(range(len(tids))) if offs is None else offs
        for slop in (1, 3):
            for lv, lw in ((None, None), (live, words)):
                unlimited = _python_two_phase(postings, tids, offs, slop, lv, 1 << 40)
                for limit in (None, 0, 5, 60, 150, 10_000):
                    want = _python_two_phase(postings, tids, offs, slop, lv, 500_000 if limit is None else limit)
                    docs, scores, total = ix.phrase_search(tids, max_doc, norms, max_doc, max_doc, 60 * max_doc, offsets=offs, slop=slop,
                                                           live_docs=lw, next_limit=limit)
                    assert total == len(want) and sorted(docs.tolist()) == [d for d, _ in want], (tids, slop, limit, lv is not None)
                    assert len(want) in (0, len(unlimited))  # all or nothing
                    cut += 1 if (unlimited and not want) else 0
                    checked += 1
    assert checked == 6 * 2 * 2 * 6 and cut >= 8
    # the exact scorer: live docs filter its matches, next_limit does not exist for it
    exact = [d for d, _ in ix.phrase_freqs([3, 4])]
    docs, _, total = ix.phrase_search([3, 4], max_doc, norms, max_doc, max_doc, 60 * max_doc, live_docs=words, next_limit=0)
    assert total == sum(1 for d in exact if live[d]) > 0 and sorted(docs.tolist()) == [d for d in exact if live[d]]
    ix.close()
