"""Positions side of the Lucene50 postings (".pos" file, position pointers in the ".doc" skip entries, BlockPostingIterator),
as restated in oracle/positions.hpp — groundwork for SURVEY §8(f)3 (positions + PhraseScorer); no product code reads
positions yet. The reference holds no test for any of it (parity unpinned), so the writer/reader pair is checked against
brute force over the input postings: every doc, freq and position, through next() and advance(), with positions read
for every doc, for some docs (the iterator must lazily skip the unread ones, across whole blocks) or partially."""
import numpy as np
import pytest


def _make_postings(rng, max_doc, dfs, max_freq):
    postings = []
    for df in dfs:
        docs = np.sort(rng.choice(max_doc, size=df, replace=False)).tolist()
        plist = []
        for d in docs:
            f = int(rng.integers(1, max_freq + 1)) if rng.random() < 0.7 else 1
            ps = np.sort(rng.choice(5000, size=f, replace=False)).tolist()
            plist.append((d, ps))
        postings.append(plist)
    return postings


@pytest.fixture(scope="module")
def index(oracle):
    rng = np.random.default_rng(12)
    dfs = [1, 2, 127, 128, 129, 255, 256, 300, 1024, 1025, 1200, 9000, 70_000]   # 70 000 docs -> 3 skip levels
    postings = _make_postings(rng, 200_000, dfs[:-1], 40) + _make_postings(rng, 200_000, dfs[-1:], 3)
    # a term whose total_term_freq is exactly 128 and one with exactly 256 (last_pos_block_fp corner cases)
    postings.append([(10 + i, [i, i + 7]) for i in range(64)])
    postings.append([(5 + 3 * i, [1, 2, 3, 4]) for i in range(64)])
    return oracle.PositionsIndex(200_000, postings), postings


def test_term_states(index):
    ix, postings = index
    for t, plist in enumerate(postings):
        st = ix.term_state(t)
        ttf = sum(len(ps) for _, ps in plist)
        assert st["doc_freq"] == len(plist) and st["total_term_freq"] == ttf
        assert (st["singleton_doc_id"] == plist[0][0]) if len(plist) == 1 else (st["singleton_doc_id"] == -1)
        assert (st["skip_offset"] > 0) == (len(plist) > 128)
        assert (st["last_pos_block_offset"] >= 0) == (ttf > 128)
    assert ix.sizes()[1] > 100_000


@pytest.mark.parametrize("version", [1, 0])
def test_next_reads_everything(oracle, index, version):
    ix, postings = index
    if version == 0:          # legacy PackedInts blocks
        postings = postings[:9]
        ix = oracle.PositionsIndex(200_000, postings, version=0)
    for t, plist in enumerate(postings):
        got = ix.iterate(t)
        assert [(d, f) for d, f, _ in got] == [(d, len(ps)) for d, ps in plist], t
        assert [p for _, _, p in got] == [ps for _, ps in plist], t


@pytest.mark.parametrize("read_every,max_positions", [(2, -1), (7, -1), (1, 1), (3, 2), (50, -1), (0, 0)])
def test_lazy_position_skipping(index, read_every, max_positions):
    """Positions are pulled only for some docs / only partly: pos_pending_count grows and skip_positions() must jump inside the
    buffered block, over whole blocks (ForUtil::skip_block) and into the vint tail."""
    ix, postings = index
    for t, plist in enumerate(postings):
        got = ix.iterate(t, read_every=read_every, max_positions=max_positions)
        assert [(d, f) for d, f, _ in got] == [(d, len(ps)) for d, ps in plist]
        for i, ((d, ps), (_, _, rp)) in enumerate(zip(plist, got)):
            if read_every > 0 and i % read_every == 0:
                want = ps if max_positions < 0 else ps[:max_positions]
                assert rp == want, (t, i, d)
            else:
                assert rp == []


def test_advance_lands_on_the_right_doc_with_the_right_positions(index):
    ix, postings = index
    rng = np.random.default_rng(3)
    for t, plist in enumerate(postings):
        docs = np.array([d for d, _ in plist])
        by_doc = dict(plist)
        for step in (1, 30, 129, 5000, 60_000):
            targets, cur = [], -1
            while True:
                tgt = cur + 1 + int(rng.integers(0, step))
                j = int(np.searchsorted(docs, tgt))
                targets.append(tgt)
                if j >= len(docs):
                    break
                cur = int(docs[j])
            for read_every in (1, 3):
                got = ix.iterate(t, targets=targets, read_every=read_every)
                assert len(got) == len(targets)
                for i, (tgt, (d, f, rp)) in enumerate(zip(targets, got)):
                    j = int(np.searchsorted(docs, tgt))
                    if j >= len(docs):
                        assert d == oracle_no_more_docs()
                    else:
                        assert d == docs[j] and f == len(by_doc[d]), (t, step, tgt)
                        if i % read_every == 0:
                            assert rp == by_doc[d], (t, step, tgt)


def oracle_no_more_docs():
    return 2**31 - 1


def test_rejects_bad_input(oracle):
    with pytest.raises(oracle.OracleError):
        oracle.PositionsIndex(100, [[(5, [1]), (5, [2])]])            # docs out of order
    with pytest.raises(oracle.OracleError):
        oracle.PositionsIndex(100, [[(5, [-1])]])                     # position < 0
    ix = oracle.PositionsIndex(100, [[(5, [3, 9])], []])
    assert ix.iterate(0) == [(5, 2, [3, 9])] and ix.iterate(1) == []


# ---- exact PhraseQuery (oracle/phrase.hpp) ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def corpus(oracle):
    """Random token streams over a small vocabulary (so that phrases really occur and terms repeat inside a doc)."""
    rng = np.random.default_rng(21)
    max_doc, vocab = 6000, 12
    docs = [rng.integers(0, vocab, size=int(rng.integers(1, 60))).tolist() if rng.random() < 0.9 else [] for _ in range(max_doc)]
    postings = [[] for _ in range(vocab + 1)]            # the last term never occurs
    for d, toks in enumerate(docs):
        where = {}
        for p, t in enumerate(toks):
            where.setdefault(t, []).append(p)
        for t, ps in where.items():
            postings[t].append((d, ps))
    return oracle.PositionsIndex(max_doc, postings), docs, postings


def _brute_phrase(docs, terms, offsets):
    out = []
    for d, toks in enumerate(docs):
        n = 0
        for start in range(-max(offsets), len(toks)):
            if all(0 <= start + o < len(toks) and toks[start + o] == t for t, o in zip(terms, offsets)):
                n += 1
        if n:
            out.append((d, n))
    return out


def test_exact_phrase_freq_matches_brute_force(corpus):
    ix, docs, postings = corpus
    rng = np.random.default_rng(5)
    phrases = [[0, 1], [1, 0], [3, 3], [2, 2, 2], [4, 5, 6], [7, 7, 8, 7], [0, 1, 2, 3], [9, 10, 11, 0, 1], [5, 12], [12, 5]]
    phrases += [rng.integers(0, 12, size=int(rng.integers(2, 5))).tolist() for _ in range(40)]
    for terms in phrases:
        offsets = list(range(len(terms)))
        assert ix.phrase_freqs(terms) == _brute_phrase(docs, terms, offsets), terms
    # phrases with position gaps ("a ? b"), as PhraseQuery::new(terms, positions) allows
    for terms, offsets in (([0, 1], [0, 2]), ([3, 4, 5], [0, 1, 3]), ([2, 2], [0, 5])):
        assert ix.phrase_freqs(terms, offsets) == _brute_phrase(docs, terms, offsets), (terms, offsets)


def test_exact_phrase_scores_and_topk(oracle, corpus):
    """score = BM25(phrase freq) with idf summed over the phrase's terms (phrase_query.rs:136-186, bm25_similarity.rs:99-114,
    203-212): recomputed here in numpy float32 in the reference's evaluation order."""
    ix, docs, postings = corpus
    max_doc = len(docs)
    lens = np.array([max(len(t), 1) for t in docs])
    L = oracle.lib()
    norms = np.array([L.orc_bm25_encode_norm(1.0, int(l)) for l in lens], dtype=np.uint8)
    sum_ttf = int(sum(len(t) for t in docs))
    doc_count = int(sum(1 for t in docs if t))
    k1, b = np.float32(1.2), np.float32(0.75)
    avgdl = np.float32(np.float64(sum_ttf) / np.float64(doc_count))
    table = np.array([L.orc_norm_table(i) for i in range(256)], dtype=np.float32)
    cache = (k1 * ((np.float32(1) - b) + b * (table / avgdl))).astype(np.float32)
    for terms in ([0, 1], [4, 5, 6], [2, 2, 2], [7, 8]):
        dfs = [len(postings[t]) for t in terms]
        idf = np.float32(0)
        for df in dfs:
            idf = np.float32(idf + np.float32(np.log(1.0 + (np.float64(doc_count) - df + 0.5) / (df + 0.5))))
        want = []
        for d, f in _brute_phrase(docs, terms, list(range(len(terms)))):
            freq = np.float32(f)
            score = np.float32(np.float32(np.float32(idf * (k1 + np.float32(1))) * freq) / np.float32(freq + cache[norms[d]]))
            want.append((float(score), d))
        want.sort(key=lambda x: (-x[0], x[1]))
        gd, gs, total = ix.phrase_search(terms, 10, norms, max_doc, doc_count, sum_ttf)
        assert total == len(want)
        assert gd.tolist() == [d for _, d in want[:10]]
        assert gs.tolist() == [s for s, _ in want[:10]]
    # a phrase with an absent term matches nothing (PhraseWeight::create_scorer -> None)
    gd, gs, total = ix.phrase_search([0, 12], 10, norms, max_doc, doc_count, sum_ttf)
    assert total == 0 and len(gd) == 0
    with pytest.raises(oracle.OracleError):
        ix.phrase_freqs([3])                       # fewer than 2 terms


# ---- the synthetic index writer's positions output vs the restated Lucene50PostingsWriter --------------------------------
@pytest.mark.parametrize("version", [1, 0])
def test_synthetic_positions_writer_is_byte_identical_to_the_restated_writer(oracle, index, version):
    """Two independent implementations of the same format — rucene_amd/csrc/indexgen (bulk: whole term at a time) and
    oracle/positions.hpp (line-faithful: doc by doc, position by position) — must produce the same .doc and .pos bytes
    and the same term pointers."""
    import __graft_entry__ as g
    g.build()
    from rucene_amd import indexgen
    _, postings = index
    postings = postings if version == 1 else postings[:9]
    ix = oracle.PositionsIndex(200_000, postings, version=version)
    seg = indexgen.build_explicit_positions(200_000, postings, version=version)
    odoc, opos = ix.files()
    assert seg.pos_bytes.tobytes() == opos
    assert seg.doc_bytes.tobytes() == odoc
    for t in range(len(postings)):
        st = ix.term_state(t)
        for name in ("doc_start_fp", "skip_offset", "total_term_freq", "doc_freq", "singleton_doc_id"):
            assert int(seg.terms[t][name]) == st[name], (t, name)
        assert int(seg.pos_start_fp[t]) == st["pos_start_fp"] and int(seg.last_pos_block_offset[t]) == st["last_pos_block_offset"], t
