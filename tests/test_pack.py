"""The mirror -> C ABI contract, checked without a GPU: what GpuIndexSearcher.pack puts into rgpu_query / rgpu_query_term for
every tree shape the GPU path serves (include/rucene_gpu.h: op byte, min_should_match byte, optional-SHOULD byte, clause
order MUST / SHOULD / MUST_NOT, absent terms, FILTER clauses as zero-weight MUST clauses, byte-named terms resolved through
the block-tree dictionary)."""
import numpy as np
import pytest


class _FakeCtx:
    def __init__(self):
        self.tables = []

    def sim_table(self, cache, k1):
        self.tables.append((np.asarray(cache).copy(), k1))
        return 7


@pytest.fixture(scope="module")
def world(oracle):
    import __graft_entry__ as g
    g.build()
    import rucene_amd
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(20_000, 500, seed=3)
    leaf = rucene_amd.LeafReader.from_synthetic(seg)
    s = object.__new__(rucene_amd.GpuIndexSearcher)     # no Context: pack() is host-only
    s.leaves, s.ctx, s.similarity, s._stats_leaf, s._weights = [leaf], _FakeCtx(), rucene_amd.BM25Similarity(), 0, {}
    s._planners, s._stats_terms = {}, None
    s.collection_statistics = rucene_amd.CollectionStatistics("body", 0, seg.max_doc, seg.doc_count, seg.sum_total_term_freq)
    return rucene_amd, seg, leaf, s


def test_ops_counts_and_clause_order(world):
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    queries = [T(3),
               B.build([T(1), T(2), T(3)], []),
               B.build([], [T(4), T(5)]),
               B.build([], [T(4), T(5), T(6)], min_should_match=2),
               B.build([T(1), T(2)], [], must_nots=[T(7), T(8)]),
               B.build([T(1)], [T(9), T(10)], must_nots=[T(11)]),
               B.build([T(1)], [], filters=[T(12)]),
               B.build([], [], filters=[T(13)]),
               B.build([T(1), T(499)], [T(-1)])]
    q, t = s.pack(queries, leaf)
    assert q["op"].tolist() == [0, 1, 2, 2 | 2 << 8, 1, 1 | 2 << 16, 1, 0, 1 | 1 << 16]
    assert q["n_terms"].tolist() == [1, 3, 2, 3, 2, 1, 2, 1, 2]
    assert q["n_must_not"].tolist() == [0, 0, 0, 0, 2, 1, 0, 0, 0]
    assert q["first_term"].tolist() == [0, 1, 4, 6, 9, 13, 17, 19, 20]
    ids = lambda lo, n: [int(np.flatnonzero(seg.terms["doc_start_fp"] == fp)[0]) if df else None
                         for fp, df in zip(t["state"]["doc_start_fp"][lo:lo + n], t["state"]["doc_freq"][lo:lo + n])]
    assert ids(9, 4) == [1, 2, 7, 8]                      # MUST, MUST, then the MUST_NOT clauses
    assert ids(13, 4) == [1, 9, 10, 11]                   # MUST, the optional SHOULD clauses, then MUST_NOT
    assert ids(17, 2) == [1, 12] and t["weight"][18] == 0.0 and t["weight"][17] > 0     # FILTER = zero-weight MUST
    assert t["weight"][19] == 0.0                         # a lone FILTER: ConstantScoreQuery with boost 0
    assert ids(20, 3) == [1, 499, None]                   # an absent term keeps its slot with doc_freq 0
    assert t["state"]["skip_offset"][22] == -1 and t["state"]["singleton_doc_id"][22] == -1
    assert (t["sim_table"][:23] == 7).all()
    # weights are BM25 idf * boost with the statistics of the (only) leaf
    w, _, _ = ra.bm25_compute_weight(1.2, 0.75, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, [int(seg.terms[3]["doc_freq"])])
    assert t["weight"][0] == np.float32(w)


def test_clause_limit_and_unsupported_trees(world):
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    with pytest.raises(ra.RgpuError) as e:
        s.pack([B.build([T(i) for i in range(30)], [T(i) for i in range(30, 65)])], leaf)     # 65 clauses > RGPU_MAX_QUERY_TERMS
    q, t = s.pack([B.build([T(i) for i in range(30)], [T(i) for i in range(30, 64)])], leaf)    # 64: fine
    assert q["n_terms"][0] == 30 and (q["op"][0] >> 16) == 34 and len(t) == 64
    assert e.value.status == -5
    with pytest.raises(ra.RgpuError):
        s.pack(["not a query"], leaf)


def test_byte_terms_go_through_the_dictionary(world, oracle):
    ra, seg, leaf, s = world
    st = np.zeros(seg.terms.size, dtype=oracle.FULL_TERM_STATE_DTYPE)
    st["base"] = seg.terms
    st["last_pos_block_offset"] = -1
    tim, tip = oracle.blocktree_write([dict(number=0, doc_count=seg.doc_count, terms=[b"t%04d" % i for i in range(seg.terms.size)], states=st)])
    dleaf = ra.LeafReader(seg.doc_bytes, seg.norms, seg.max_doc, None, sum_total_term_freq=seg.sum_total_term_freq,
                          term_dictionary=ra.TermDictionary(tim, tip, [(0, 2)], seg.max_doc), field_number=0)
    s2 = object.__new__(ra.GpuIndexSearcher)
    s2.leaves, s2.ctx, s2.similarity, s2._stats_leaf, s2._weights = [dleaf], _FakeCtx(), ra.BM25Similarity(), 0, {}
    s2._planners, s2._stats_terms = {}, None
    s2.collection_statistics = s.collection_statistics
    T, B = ra.TermQuery, ra.BooleanQuery
    by_id = s.pack([T(3), B.build([T(1), T(2)], [T(9)], must_nots=[T(4)])], leaf)
    by_text = s2.pack([T(b"t0003"), B.build([T(b"t0001"), T(b"t0002")], [T(b"t0009")], must_nots=[T(b"t0004")])], dleaf)
    assert by_id[0].tobytes() == by_text[0].tobytes() and by_id[1].tobytes() == by_text[1].tobytes()
    assert s2._planners                                  # byte-named batches are planned natively too (rgpu_plan_batch_bytes) ...
    qs = [T(b"t0003"), B.build([T(b"t0001"), T(b"t0002", 2.0)], [T(b"t0009")], must_nots=[T(b"t0004")]), B.build([], [T(b"zz"), T(b"t0007")])]
    native = s2.pack(qs, dleaf)                           # ... and write what the clause-by-clause path writes
    by_hand = s2._pack_clause_by_clause(qs, dleaf)
    assert native[0].tobytes() == by_hand[0].tobytes() and native[1].tobytes() == by_hand[1].tobytes()
    q, t = s2.pack([T(b"nope")], dleaf)
    assert t["state"]["doc_freq"][0] == 0
    with pytest.raises(ra.RgpuError):
        s.pack([T(b"t0003")], leaf)                      # a leaf without a dictionary cannot resolve bytes


def test_array_path_equals_the_general_path(world):
    """A batch that names every term the same way is planned behind the C ABI (rgpu_plan_batch_ids, csrc/host/batch_planner.hpp):
    the structs must be the ones the clause-by-clause Python path writes, absent and out-of-table ids and boosts included."""
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    rng = np.random.default_rng(5)

    def batch(wrap):
        qs = []
        for i in range(300):
            ids = [wrap(int(x)) for x in rng.integers(-2, seg.terms.size + 3, size=int(rng.integers(1, 7)))]
            kind = i % 5
            if kind == 0:
                qs.append(T(ids[0]))
            elif kind == 1:
                qs.append(B.build([T(x) for x in ids], []))
            elif kind == 2:
                qs.append(B.build([], [T(x) for x in ids], min_should_match=1 + i % 2))
            elif kind == 3:
                qs.append(B.build([T(x) for x in ids[:1]], [T(x) for x in ids[1:3]], must_nots=[T(x) for x in ids[3:]]))
            else:
                qs.append(B.build([], [T(x) for x in ids[:2]], must_nots=[T(x) for x in ids[2:]]) if len(ids) > 2 else T(ids[0]))
        return qs

    rng = np.random.default_rng(5)
    fast = s.pack(batch(int), leaf)
    rng = np.random.default_rng(5)
    general = s.pack(batch(np.int64), leaf)             # numpy scalars are not `int`: the clause-by-clause path
    assert fast[0].tobytes() == general[0].tobytes()
    assert fast[1].tobytes() == general[1].tobytes()
    assert s._planners                                  # ... and the first batch did take the native planner
    boosted = lambda wrap: [T(wrap(3), 2.5), B.build([T(wrap(1), 0.5), T(wrap(2))], [T(wrap(9), 3.0)], must_nots=[T(wrap(4))]), B.build([], [T(wrap(5)), T(wrap(499), 1.5)])]
    a, b = s.pack(boosted(int), leaf), s.pack(boosted(np.int64), leaf)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_pack_uniform_equals_pack(world):
    """The array planner writes what pack() writes for the same batch as query objects."""
    ra, seg, leaf, s = world
    from rucene_amd._lib import OP_TERM, OP_AND, OP_OR
    T, B = ra.TermQuery, ra.BooleanQuery
    rng = np.random.default_rng(9)
    ids = rng.integers(-1, seg.terms.size + 2, size=(200, 5))
    for op, nc, msm in ((OP_TERM, 1, 0), (OP_AND, 3, 0), (OP_OR, 5, 0), (OP_OR, 4, 3), (OP_AND, 1, 0), (OP_OR, 1, 2)):
        sub = ids[:, :nc]
        if op == OP_TERM:
            qs = [T(int(r[0])) for r in sub]
        elif op == OP_AND:
            qs = [B.build([T(int(x)) for x in r], []) for r in sub]
        else:
            qs = [B.build([], [T(int(x)) for x in r], min_should_match=msm) for r in sub]
        want = s.pack(qs, leaf)
        got = s.pack_uniform(op, sub, leaf, min_should_match=msm)
        assert got[0].tobytes() == want[0].tobytes(), (op, nc, msm)
        assert got[1].tobytes() == want[1].tobytes(), (op, nc, msm)
    with pytest.raises(ra.RgpuError):
        s.pack_uniform(OP_TERM, ids[:, :2], leaf)
    with pytest.raises(ra.RgpuError):
        s.pack_uniform(OP_OR, np.zeros((3, 65), dtype=np.int64), leaf)


def test_batch_term_weights_equal_the_single_term_ones(world):
    """rgpu_bm25_term_weights(dfs) == rgpu_bm25_compute_weight([df]) for every df, bit for bit (idf in f64, rounded once)."""
    ra, seg, leaf, s = world
    dfs = np.unique(np.concatenate([seg.terms["doc_freq"].astype(np.int64), [0, 1, 2, seg.max_doc // 2, seg.max_doc]]))
    for boost in (1.0, 2.5):
        got = ra._lib.bm25_term_weights(seg.max_doc, seg.doc_count, dfs, boost)
        want = np.array([ra.bm25_compute_weight(1.2, 0.75, seg.max_doc, seg.doc_count, seg.sum_total_term_freq, [int(d)], boost)[0] for d in dfs],
                        dtype=np.float32)
        assert got.view(np.int32).tolist() == want.view(np.int32).tolist()


def test_a_disjunction_under_must_is_packed_as_required_should_clauses(world):
    """VERDICT r5 missing 5: "+a +(b c)" — a should-only BooleanQuery as a MUST clause — goes to the GPU path as
    RGPU_OP_WITH_SHOULD(AND, n) | RGPU_OP_SHOULD_REQUIRED | RGPU_OP_NESTED_AT(its index among the MUST clauses); the library forms
    ConjunctionScorer::score's f32 sum (children sorted by cost, conjunction_scorer.rs:27-43, 87-95) per leaf. Other shapes stay
    trees the GPU path declines."""
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    REQ, OP_AND = ra._lib.OP_SHOULD_REQUIRED, ra.OP_AND
    assert REQ == 1 << 24
    df = lambda t: int(seg.terms[t]["doc_freq"])   # noqa: E731
    s.flatten_nested, s.cpu_fallback = False, None
    q, t = s.pack([B.build([T(1), B.build([], [T(2), T(3)])], [], must_nots=[T(9)], filters=[T(7)])], leaf)
    assert q[0]["op"] == (OP_AND | (2 << 16) | REQ | (1 << 26)) and q[0]["n_terms"] == 2 and q[0]["n_must_not"] == 1
    want = s.pack([B.build([T(1)], [T(2), T(3)], must_nots=[T(9)], filters=[T(7)])], leaf)   # same clauses, same order, same weights
    assert t.tobytes() == want[1].tobytes() and want[0][0]["op"] == (OP_AND | (2 << 16))
    # two scoring MUST clauses: exact only when the disjunction is the costliest child
    rare = [i for i in range(len(seg.terms)) if 2 <= df(i) <= 40][:4]
    assert df(0) + df(1) > max(df(rare[0]), df(rare[1]))
    q, _ = s.pack([B.build([T(rare[0]), T(rare[1]), B.build([], [T(0), T(1)])], [])], leaf)
    assert q[0]["op"] == (OP_AND | (2 << 16) | REQ | (2 << 26)) and q[0]["n_terms"] == 2
    # ... and a disjunction that is NOT the costliest child is served just the same: the library sorts the children per leaf and adds
    # the nested sum where ConjunctionScorer::score adds it; RGPU_OP_NESTED_AT (bits 26..) names its place among the MUST clauses
    cheap = B.build([T(0), B.build([], [T(rare[2]), T(rare[3])]), T(1)], [])
    q, _ = s.pack([cheap], leaf)
    assert q[0]["op"] == (OP_AND | (2 << 16) | REQ | (1 << 26)) and q[0]["n_terms"] == 2
    q, _ = s.pack([B.build([T(i) for i in range(40)] + [B.build([], [T(50), T(51)])], [])], leaf)   # at = 40: the int32's sign bit
    assert q[0]["op"] == np.int32(np.uint32(OP_AND | (2 << 16) | REQ | (40 << 26))) and q[0]["n_terms"] == 40
    # not this shape: SHOULD clauses beside it, a nested min_should_match, ten children, a nested MUST_NOT
    for tree in (B.build([T(1), B.build([], [T(2), T(3)])], [T(4)]),
                 B.build([T(1), B.build([], [T(2), T(3), T(4)], min_should_match=2)], []),
                 B.build([T(1), B.build([], [T(i) for i in range(2, 12)])], []),
                 B.build([T(1), B.build([], [T(2), T(3)], must_nots=[T(4)])], [])):
        with pytest.raises(ra.RgpuError) as e:
            s.pack([tree], leaf)
        assert e.value.status == -5


def test_nested_clauses_that_do_not_score_pack_as_their_flat_forms(world):
    """"-(b c)" is the MUST_NOT clauses b, c (ReqNotScorer over the nested DisjunctionSumScorer: boolean_query.rs:236-252) and
    "#(+b +c)" is the FILTER clauses b, c (weights created with needs_scores = false score 0.0: boolean_query.rs:106-108) — exact, so
    BooleanQuery.normalized() rewrites them whatever flatten_nested says; they combine with the scoring nested shapes."""
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    s.flatten_nested, s.cpu_fallback = False, None
    same = lambda a, b: a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()   # noqa: E731
    nested = B.build([T(1), T(2)], [], must_nots=[B.build([], [T(9), T(11)])], filters=[B.build([T(7), T(8)], [])])
    assert same(s.pack([nested], leaf), s.pack([B.build([T(1), T(2)], [], must_nots=[T(9), T(11)], filters=[T(7), T(8)])], leaf))
    # a should-only tree with a nested MUST_NOT disjunction next to a term one; a nested filter of filters
    assert same(s.pack([B.build([], [T(1), T(2)], must_nots=[T(5), B.build([], [T(9), T(11)])])], leaf),
                s.pack([B.build([], [T(1), T(2)], must_nots=[T(5), T(9), T(11)])], leaf))
    assert same(s.pack([B.build([T(1)], [T(3)], filters=[B.build([], [], filters=[T(7), T(8)])])], leaf),
                s.pack([B.build([T(1)], [T(3)], filters=[T(7), T(8)])], leaf))
    # together with a disjunction under MUST: RGPU_OP_SHOULD_REQUIRED as before, the expanded clauses in their places
    q, t = s.pack([B.build([T(1), B.build([], [T(2), T(3)])], [], must_nots=[B.build([], [T(9), T(11)])], filters=[B.build([T(7), T(8)], [])])], leaf)
    want = s.pack([B.build([T(1), B.build([], [T(2), T(3)])], [], must_nots=[T(9), T(11)], filters=[T(7), T(8)])], leaf)
    assert same((q, t), want) and q[0]["op"] == (ra.OP_AND | (2 << 16) | ra._lib.OP_SHOULD_REQUIRED | (1 << 26)) and q[0]["n_must_not"] == 2
    # "+a +b #(c d)": a filter by a disjunction is the required disjunction of zero-weight clauses, behind the MUST clauses
    q, t = s.pack([B.build([T(1), T(2)], [], filters=[B.build([], [T(7), T(8)])], must_nots=[T(9)])], leaf)
    want = s.pack([B.build([T(1), T(2), B.build([], [T(7), T(8)])], [], must_nots=[T(9)])], leaf)
    assert q[0]["op"] == want[0][0]["op"] == (ra.OP_AND | (2 << 16) | ra._lib.OP_SHOULD_REQUIRED | (2 << 26)) and q[0]["n_terms"] == 2 and q[0]["n_must_not"] == 1
    assert (t["weight"][2:4] == 0).all() and (want[1]["weight"][2:4] > 0).all() and (t["weight"][:2] == want[1]["weight"][:2]).all()
    assert t["state"].tobytes() == want[1]["state"].tobytes()
    # NOT rewritten: the outer min_should_match counts the MUST_NOT disjunction's matches too (one scorer vs two), a MUST_NOT
    # conjunction, a nested min_should_match, a FILTER disjunction beside SHOULD clauses or beside another nested clause, deeper trees
    for tree in (B.build([], [T(1), T(2), T(3)], must_nots=[B.build([], [T(9), T(11)])], min_should_match=2),
                 B.build([T(1), T(2)], [], must_nots=[B.build([T(9), T(11)], [])]),
                 B.build([T(1), T(2)], [], must_nots=[B.build([], [T(9), T(11), T(12)], min_should_match=2)]),
                 B.build([T(1), T(2)], [T(4)], filters=[B.build([], [T(7), T(8)])]),
                 B.build([T(1), B.build([], [T(2), T(3)])], [], filters=[B.build([], [T(7), T(8)])]),
                 B.build([T(1), T(2)], [], must_nots=[B.build([], [T(9), B.build([], [T(11), T(12)])])])):
        with pytest.raises(ra.RgpuError) as e:
            s.pack([tree], leaf)
        assert e.value.status == -5


def test_nested_boolean_trees_fold_one_level_or_fall_back(world):
    """SURVEY 8(f)1 "everything else to the CPU path" as code (VERDICT r4 missing 3 / item 10): a BooleanQuery whose clauses are
    themselves BooleanQuerys builds (as in the reference); the GPU path serves it only when flatten_nested folds it into one clause
    list (MUST of MUSTs, SHOULD of SHOULDs), and search() hands every other tree to cpu_fallback."""
    ra, seg, leaf, s = world
    T, B = ra.TermQuery, ra.BooleanQuery
    nested_and = B.build([T(1), B.build([T(2), T(3)], [])], [])
    nested_or = B.build([], [T(4), T(7), B.build([], [T(5), T(6)])])              # the nested disjunction as THIRD clause: foldable only
    nested_or_2nd = B.build([], [T(4), B.build([], [T(5), T(6)]), T(7)])          # ... as first or second clause: exact as [5, 6, 4, 7]
    mixed = B.build([B.build([], [T(2), T(3)]), B.build([], [T(4), T(5)])], [])   # two disjunctions under MUST: not served
    with_msm = B.build([], [T(4), B.build([], [T(5), T(6)])], min_should_match=2)   # msm counts the OUTER clauses: not foldable
    deep = B.build([T(1), B.build([T(2), B.build([T(3), T(4)], [])], [])], [])      # two levels: not foldable
    assert not nested_and.is_flat() and B.build([T(1), T(2)], []).is_flat()
    s.flatten_nested, s.cpu_fallback = False, None
    for q in (nested_or, mixed):
        with pytest.raises(ra.RgpuError) as e:
            s.pack([q], leaf)
        assert e.value.status == -5   # ErrorKind::UnsupportedOperation
    # MUST [t1, MUST [t2, t3]] with ONE outer MUST clause is served as it is, bit-exact: the nested conjunction's sum is formed
    # first (RGPU_OP_NESTED_MUST, test_a_conjunction_under_must) — the same clauses as the flat query, another op
    q, t = s.pack([nested_and], leaf)
    flat_q, flat_t = s.pack([B.build([T(1), T(2), T(3)], [])], leaf)
    assert q[0]["op"] == (ra.OP_AND | (2 << 16) | ra._lib.OP_NESTED_MUST | (1 << 26)) and q[0]["n_terms"] == 1 and t.tobytes() == flat_t.tobytes()
    # SHOULD [t4, SHOULD [t5, t6], t7]: DisjunctionSumScorer adds its children in clause order from 0.0, (t4 + (t5 + t6)) + t7 — the
    # flat disjunction with the nested clauses in FRONT forms ((t5 + t6) + t4) + t7: the one add that differs commutes. No flag.
    for fl in (False, True):
        s.flatten_nested = fl
        q, t = s.pack([nested_or_2nd, B.build([], [B.build([], [T(5), T(6)]), T(4), T(7)], must_nots=[T(9)])], leaf)
        flat_q, flat_t = s.pack([B.build([], [T(5), T(6), T(4), T(7)]), B.build([], [T(5), T(6), T(4), T(7)], must_nots=[T(9)])], leaf)
        assert (q == flat_q).all() and (t == flat_t).all()
    s.flatten_nested = False
    with pytest.raises(ra.RgpuError):   # ten clauses in all: the heap-order kernels would take the flat query
        s.pack([B.build([], [T(1), B.build([], [T(i) for i in range(2, 9)]), T(10), T(11)])], leaf)
    s.flatten_nested = True
    q, t = s.pack([nested_or], leaf)
    flat_q, flat_t = s.pack([B.build([], [T(4), T(7), T(5), T(6)])], leaf)
    assert (q == flat_q).all() and (t == flat_t).all()               # the folded tree IS the flat query
    two_nested = B.build([T(1), B.build([T(2), T(3)], []), B.build([T(4), T(5)], [])], [])   # MUST of two MUSTs: only the fold serves it
    q, t = s.pack([two_nested], leaf)
    flat_q, flat_t = s.pack([B.build([T(i) for i in range(1, 6)], [])], leaf)
    assert (q == flat_q).all() and (t == flat_t).all()
    for q in (mixed, with_msm, deep):
        with pytest.raises(ra.RgpuError) as e:
            s.pack([q], leaf)
        assert e.value.status == -5
    # the seam: search() -> cpu_fallback for what the GPU path declines (and only for that)
    seen = []
    s.cpu_fallback = lambda query, collector: seen.append((query, collector)) or "cpu"
    col = ra.TopDocsCollector(10)
    assert s.search(mixed, col) == "cpu" and seen == [(mixed, col)]
    assert s.search(T(1), object()) == "cpu" and len(seen) == 2      # a collector the GPU path does not serve
    s.flatten_nested, s.cpu_fallback = False, None
