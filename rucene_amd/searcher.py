"""Host-side mirror of the slice of Rucene's search API that the GPU path serves, with the reference's names,
argument meaning and error behaviour so tests read like the reference's own (paths relative to
/root/reference/src/core):

    search/searcher.rs:205-249, 306-363, 487-525, 732-767   IndexSearcher::search, statistics quirk
    search/query/term_query.rs:45-95                         TermQuery::new(term, boost) / create_weight
    search/query/boolean_query.rs:40-86                      BooleanQuery::build(musts, shoulds, ...)
    search/similarity/bm25_similarity.rs:45-177              BM25Similarity::new(k1, b) / compute_weight
    search/collector/top_docs.rs:97-183                      TopDocsCollector::new(k) / top_docs()
    search/statistics.rs                                     CollectionStatistics / TermStatistics

    index/reader/directory_reader.rs, segment_reader.rs      open_directory: commit point -> segments -> per-format readers

Terms are named by bytes and resolved through each leaf's block-tree dictionary (rgpu_terms_*), or — synthetic indexes —
by an id into a flat table of BlockTermState records. All scoring work happens in librucene_gpu.so; this module only
resolves terms, computes BM25 weights (via the C ABI's host helper) and packs query structs.
"""
import numpy as np

from . import _lib
from ._lib import OP_AND, OP_NESTED_MUST, OP_OR, OP_SHOULD_REQUIRED, OP_TERM, QUERY_DTYPE, QUERY_TERM_DTYPE, TERM_STATE_DTYPE, RgpuError


class CollectionStatistics:
    """search/statistics.rs CollectionStatistics (field-level)."""

    def __init__(self, field, doc_base, max_doc, doc_count, sum_total_term_freq, sum_doc_freq=-1):
        self.field, self.doc_base, self.max_doc = field, doc_base, max_doc
        self.doc_count, self.sum_total_term_freq, self.sum_doc_freq = doc_count, sum_total_term_freq, sum_doc_freq


class BM25Similarity:
    """bm25_similarity.rs:45-63. compute_weight returns (weight, cache[256]) == BM25SimWeight{weight, cache}."""
    DEFAULT_BM25_K1 = 1.2
    DEFAULT_BM25_B = 0.75

    def __init__(self, k1=DEFAULT_BM25_K1, b=DEFAULT_BM25_B):
        self.k1, self.b = float(np.float32(k1)), float(np.float32(b))

    def compute_weight(self, collection_stats, doc_freqs, boost=1.0):
        w, _idf, cache = _lib.bm25_compute_weight(self.k1, self.b, collection_stats.max_doc, collection_stats.doc_count,
                                                  collection_stats.sum_total_term_freq, doc_freqs, boost)
        return w, cache

    @staticmethod
    def encode_norm_value(boost, field_length):
        return _lib.bm25_encode_norm(boost, field_length)

    def __str__(self):
        return "BM25Similarity(k1: %s, b: %s)" % (self.k1, self.b)


class LeafReader:
    """One segment as the searcher sees it (index/reader/leaf_reader.rs:62-182, reduced to what BM25 term and
    boolean queries read): postings file, norms, live docs, FieldReader statistics, and the term dictionary — either a
    flat table of term states indexed by term id (synthetic indexes) or a block-tree dictionary (`.tim`/`.tip`) keyed
    by term bytes."""

    RESOLVED_CACHE_TERMS = 1 << 20   # bound of the per-leaf memo of resolved byte terms

    def __init__(self, doc_bytes, norms, max_doc, terms, doc_base=0, live_docs=None, doc_count=None,
                 sum_total_term_freq=0, sum_doc_freq=-1, field="body", term_dictionary=None, field_number=0,
                 index_options=_lib.INDEX_OPTIONS_DOCS_AND_FREQS, has_payloads=False):
        self.doc_bytes, self.norms, self.max_doc, self.doc_base = doc_bytes, norms, int(max_doc), int(doc_base)
        self.terms = np.ascontiguousarray(terms if terms is not None else [], dtype=TERM_STATE_DTYPE)
        self.live_docs = live_docs
        self.doc_count = int(max_doc if doc_count is None else doc_count)
        self.sum_total_term_freq, self.sum_doc_freq, self.field = int(sum_total_term_freq), int(sum_doc_freq), field
        self.term_dictionary, self.field_number = term_dictionary, int(field_number)
        self.index_options = int(index_options)  # doc::IndexOptions ordinal: 1 Docs, 2 DocsAndFreqs, 3 DocsAndFreqsAndPositions, 4 ...AndOffsets
        self.has_payloads = bool(has_payloads)   # FieldInfo::has_store_payloads
        self.pos_bytes = None      # the ".pos" file of a positions field
        self.pay_bytes = None      # the ".pay" file of a field that stores payloads or offsets
        self.term_positions = None  # per flat term id: TERM_POSITIONS_DTYPE records (synthetic segments)
        self._resolved = {}  # term bytes -> state or None
        self.segment = None  # rgpu_segment, created by the searcher

    @classmethod
    def from_synthetic(cls, seg, doc_base=None):
        return cls(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, seg.doc_base if doc_base is None else doc_base,
                   seg.live_docs, seg.doc_count, seg.sum_total_term_freq, seg.sum_doc_freq)

    @classmethod
    def from_synthetic_positions(cls, seg, doc_base=None, offsets=False, payloads=False):
        """A synthetic DocsAndFreqsAndPositions field (indexgen.build_explicit_positions): .doc + .pos + per-term pointers;
        offsets / payloads: as the segment was built (+ .pay)."""
        leaf = cls(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, seg.doc_base if doc_base is None else doc_base,
                   seg.live_docs, seg.doc_count, seg.sum_total_term_freq, seg.sum_doc_freq,
                   index_options=_lib.INDEX_OPTIONS_OFFSETS if offsets else _lib.INDEX_OPTIONS_POSITIONS, has_payloads=payloads)
        leaf.pos_bytes = seg.pos_bytes
        leaf.pay_bytes = seg.pay_bytes
        tp = np.zeros(seg.terms.size, dtype=_lib.TERM_POSITIONS_DTYPE)
        tp["pos_start_fp"], tp["last_pos_block_offset"] = seg.pos_start_fp, seg.last_pos_block_offset
        if seg.pay_start_fp is not None:
            tp["pay_start_fp"] = seg.pay_start_fp
        leaf.term_positions = tp
        return leaf

    def positions_state(self, term):
        """The position-stream pointers of BlockTermState for `term` (None when the term is absent)."""
        if isinstance(term, bytes):
            if self.term_dictionary is None:
                raise RgpuError(-2, "this leaf has no term dictionary: query it by term id")
            states, pos, found = self.term_dictionary.lookup_positions(self.field_number, [term])
            return (states[0], pos[0]) if found[0] else None
        st = self.term_state(term)
        if st is None or self.term_positions is None:
            return None
        return st, self.term_positions[term]

    @classmethod
    def from_index_files(cls, doc, tim, tip, nvm, nvd, max_doc, field_number=0, index_options=2, liv=None, del_count=-1,
                         doc_base=0, field="body", other_fields=(), fnm=None, has_payloads=False):
        """A segment as Rucene wrote it (SegmentReader::open -> the per-format producers): `.doc` postings, `.tim`/`.tip`
        block-tree term dictionary, `.nvm`/`.nvd` norms, optional `.liv` live docs. With `fnm` (the segment's field infos
        file) the searched field is found by its name `field` and every other indexed field is declared from the file;
        otherwise pass `field_number`/`index_options` and `other_fields`: (number, index_options[, has_payloads]) of the
        segment's other indexed fields (the `.tim` summary lists them all).
        Any indexed field can be searched; for phrases set `pos_bytes` (and, for a field with payloads / offsets, `pay_bytes`)."""
        if fnm is not None:
            infos = _lib.field_infos_from_lucene60(fnm)
            mine = [fi for fi in infos if fi["name"] == field]
            if not mine:
                raise RgpuError(-2, "no field named %r in this segment" % field)
            field_number, index_options, has_payloads = mine[0]["number"], mine[0]["index_options"], bool(mine[0]["has_payloads"])
            other_fields = [(fi["number"], fi["index_options"], int(fi["has_payloads"])) for fi in infos
                            if fi["index_options"] != 0 and fi["name"] != field]
        if index_options not in (_lib.INDEX_OPTIONS_DOCS, _lib.INDEX_OPTIONS_DOCS_AND_FREQS, _lib.INDEX_OPTIONS_POSITIONS, _lib.INDEX_OPTIONS_OFFSETS):
            raise RgpuError(-5, "the searched field must be indexed (IndexOptions::Docs ... ::DocsAndFreqsAndPositionsAndOffsets)")
        td = _lib.TermDictionary(tim, tip, [(field_number, index_options, int(has_payloads))] + list(other_fields), max_doc)
        stats = td.field_stats(field_number)
        if stats is None:
            raise RgpuError(-2, "field %d has no postings in this segment" % field_number)
        norms = _lib.norms_from_lucene53(nvm, nvd, field_number, max_doc)
        live = _lib.live_docs_from_lucene50(liv, max_doc, del_count) if liv is not None else None
        return cls(doc, norms, max_doc, None, doc_base, live, stats["doc_count"], stats["sum_total_term_freq"],
                   stats["sum_doc_freq"], field, td, field_number, index_options, has_payloads)

    def resolve(self, terms):
        """One batched dictionary lookup for the byte terms not seen before (seek_exact + term_state each)."""
        new = [t for t in dict.fromkeys(terms) if isinstance(t, bytes) and t not in self._resolved]
        if not new:
            return
        if self.term_dictionary is None:
            raise RgpuError(-2, "this leaf has no term dictionary: query it by term id")
        states, found = self.term_dictionary.lookup(self.field_number, new)
        if len(self._resolved) + len(new) > self.RESOLVED_CACHE_TERMS:   # a memo, not an index: start over rather than grow forever
            self._resolved.clear()
        for t, st, ok in zip(new, states, found):
            self._resolved[t] = st if ok else None

    def term_state(self, term):
        """TermIterator::seek_exact + term_state(); None when the term is absent from this leaf."""
        if isinstance(term, bytes):
            if term not in self._resolved:
                self.resolve([term])
            return self._resolved[term]
        if term < 0 or term >= self.terms.size or self.terms[term]["doc_freq"] <= 0:
            return None
        return self.terms[term]


def _base36(v):
    digits, out = "0123456789abcdefghijklmnopqrstuvwxyz", ""
    while True:
        out = digits[v % 36] + out
        v //= 36
        if v == 0:
            return out


def open_directory(path, field="body"):
    """StandardDirectoryReader::open for the slice this path needs: the newest commit point `segments_N` names the
    segments; per segment `.si` gives max_doc, `.fnm` the field's number and index options, `_Lucene50_0.{doc,tim,tip}` the
    postings and term dictionary, `.nvm/.nvd` the norms, `_<delgen>.liv` the live docs (index/reader/directory_reader.rs:
    90-140; segment_reader.rs open; file names per codec/segment_infos/mod.rs:60-114). Returns the LeafReaders with
    cumulative doc bases, ready for GpuIndexSearcher. A compound segment's files are taken out of its `.cfs` through the
    `.cfe` entry table (codec/compound.rs); its `.si` and `.liv` stay outside, as Rucene writes them."""
    import os
    gens = [int(f[len("segments_"):], 36) for f in os.listdir(path) if f.startswith("segments_")]
    if not gens:
        raise RgpuError(-6, "no segments_N file found in %s" % path)   # IndexNotFound
    gen = max(gens)

    def read(name):
        with open(os.path.join(path, name), "rb") as fh:
            return fh.read()
    leaves, doc_base = [], 0
    for seg in _lib.commit_from_segments_file(read("segments_" + _base36(gen)), gen):
        name = seg["name"]
        info = _lib.segment_info_from_lucene62(read(name + ".si"), expected_id=seg["id"])
        inner = None
        if info["is_compound_file"]:
            inner = _lib.compound_files_from_lucene50(read(name + ".cfe"), read(name + ".cfs"), expected_id=seg["id"])

        def part(suffix, _inner=inner, _name=name):   # a file of this segment, from the directory or from inside its .cfs
            if _inner is None:
                return read(_name + suffix)
            if suffix not in _inner:
                raise RgpuError(-6, "%s%s is not in the compound file" % (_name, suffix))
            return _inner[suffix]
        # field infos rewritten by a doc-values update live outside the compound file under a generation suffix
        # (SegmentCommitInfo::field_infos_gen; file_name_from_generation, codec/segment_infos/mod.rs:100-114)
        fnm = read("%s_%s.fnm" % (name, _base36(seg["field_infos_gen"]))) if seg["field_infos_gen"] > 0 else part(".fnm")
        if seg["del_count"] > info["max_doc"]:
            raise RgpuError(-4, "invalid deletion count: %d vs maxDoc=%d" % (seg["del_count"], info["max_doc"]))
        # postings files carry PerFieldPostingsFormat's suffix: format "Lucene50", suffix "0" (field_infos/mod.rs:441-447)
        liv = read("%s_%s.liv" % (name, _base36(seg["del_gen"]))) if seg["del_gen"] >= 0 and seg["del_count"] > 0 else None
        leaf = LeafReader.from_index_files(np.frombuffer(part("_Lucene50_0.doc"), dtype=np.uint8), part("_Lucene50_0.tim"),
                                           part("_Lucene50_0.tip"), part(".nvm"), part(".nvd"), info["max_doc"],
                                           liv=liv, del_count=seg["del_count"] if liv is not None else -1, doc_base=doc_base,
                                           field=field, fnm=fnm)
        leaves.append(leaf)
        doc_base += info["max_doc"]
    return leaves


def _i32(op):
    """rgpu_query.op is an int32: RGPU_OP_NESTED_AT(i >= 32) sets its sign bit"""
    return ((int(op) + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


class TermQuery:
    """term: bytes (resolved through each leaf's term dictionary) or an int term id (synthetic flat term table)."""

    def __init__(self, term, boost=1.0):
        self.term = bytes(term) if isinstance(term, (bytes, bytearray, memoryview)) else int(term)
        self.boost = float(boost)

    def extract_terms(self):
        return [self]

    def __str__(self):
        return "TermQuery(field: body, term: %r, boost: %s)" % (self.term, self.boost)


class PhraseQuery:
    """PhraseQuery::build / ::new (query/phrase_query.rs:60-130): terms at positions 0, 1, 2, ... (or `positions`); slop 0 = an
    exact phrase, slop > 0 = the reference's SloppyPhraseScorer."""

    def __init__(self, terms, positions=None, boost=1.0, slop=0):
        self.terms = [bytes(t) if isinstance(t, (bytes, bytearray, memoryview)) else int(t) for t in terms]
        self.positions = list(range(len(self.terms))) if positions is None else [int(p) for p in positions]
        self.boost = float(boost)
        self.slop = int(slop)
        if self.slop < 0:
            raise RgpuError(-2, "Slop must be >= 0, got %d" % self.slop)
        if len(self.terms) < 2:
            raise RgpuError(-2, "PhraseWeight does not support less than 2 terms, call rewrite first")
        if len(self.positions) != len(self.terms):
            raise RgpuError(-2, "Must have as many terms as positions")


class BooleanQuery:
    """Only the trees the GPU path serves: all-SHOULD with any min_should_match (OR), or MUST clauses (AND) with optional
    SHOULD clauses beside them (ReqOptScorer, boolean_query.rs:253-262 — its sequential skipping rule included unless the
    context was opened with req_opt_rule=-1, see RGPU_OP_WITH_SHOULD in include/rucene_gpu.h); each optionally with MUST_NOT TermQuery clauses
    (ReqNotScorer, boolean_query.rs:235-273). FILTER clauses are required clauses that score 0 (create_weight with
    needs_scores = false -> NonScoringSimilarity, boolean_query.rs:106-108, searcher.rs:158-202): they ride as MUST clauses
    of weight 0, which leaves every f32 sum unchanged."""

    def __init__(self, must_queries, should_queries, min_should_match, must_not_queries=(), filter_queries=()):
        self.must_queries, self.should_queries, self.min_should_match = must_queries, should_queries, min_should_match
        self.must_not_queries = list(must_not_queries)
        self.filter_queries = list(filter_queries)

    @staticmethod
    def build(musts, shoulds, filters=(), must_nots=(), min_should_match=0):
        # boolean_query.rs:40-86
        msm = min_should_match if min_should_match > 0 else (1 if len(musts) == 0 else 0)
        if len(musts) + len(shoulds) + len(filters) + len(must_nots) == 0:
            raise RgpuError(-2, "boolean query should at least contain one inner query!")
        if len(must_nots) == 0 and len(musts) + len(shoulds) + len(filters) == 1:
            if filters:   # ConstantScoreQuery::with_boost(filter, 0.0) (boolean_query.rs:70-73): every match scores 0
                return TermQuery(filters[0].term, 0.0)
            return (list(musts) + list(shoulds))[0]
        if msm > 255:
            raise RgpuError(-5, "min_should_match above 255")
        if (musts or filters) and not shoulds:
            msm = 0   # nothing for it to count (beside MUST clauses it has no effect anyway: ReqOptScorer only advances the optional scorer)
        # (a clause that is itself a BooleanQuery builds, as it does in the reference: whether the GPU path serves the tree is
        # decided when it is searched — GpuIndexSearcher.flatten_nested / cpu_fallback)
        if any(not isinstance(q, (TermQuery, BooleanQuery)) for q in list(musts) + list(shoulds) + list(must_nots) + list(filters)):
            raise RgpuError(-5, "only term and boolean clauses are known to this mirror")
        return BooleanQuery(list(musts), list(shoulds), msm, list(must_nots), list(filters))

    def normalized(self):
        """Nested clauses that do NOT score, in their exact flat forms (same docs, counts and f32 sums as the reference's scorer tree),
        or self when there is none of that shape:
          * a MUST_NOT clause that is a should-only BooleanQuery of terms, "-(b c)": ReqNotScorer excludes what the nested
            DisjunctionSumScorer matches — the MUST_NOT clauses b, c. (boolean_query.rs:236-252 puts ONE disjunction over the MUST_NOT
            scorers with the OUTER min_should_match: the same exclusion only while that is <= 1; above it the clause stays nested.)
          * a FILTER clause that is a must- / filter-only BooleanQuery of terms, "#(+b +c)": its weights are created with
            needs_scores = false (boolean_query.rs:106-108), each clause scores 0.0 and the nested conjunction's 0.0 + 0.0 joins the outer
            sum as one 0.0 — the FILTER clauses b, c (x + 0.0 == x wherever the cost order puts them).
          * a FILTER clause that is a should-only BooleanQuery of terms, "+a #(b c)": the required disjunction of zero-boost clauses."""
        must_nots, filters, changed = [], [], False
        for q in self.must_not_queries:
            if (isinstance(q, BooleanQuery) and self.min_should_match <= 1 and q.is_flat() and q.should_queries and q.min_should_match <= 1
                    and not (q.must_queries or q.must_not_queries or q.filter_queries)):
                must_nots.extend(q.should_queries)
                changed = True
            else:
                must_nots.append(q)
        musts = list(self.must_queries)
        for q in self.filter_queries:
            if (isinstance(q, BooleanQuery) and q.is_flat() and (q.must_queries or q.filter_queries)
                    and not (q.should_queries or q.must_not_queries)):
                filters.extend(q.must_queries + q.filter_queries)
                changed = True
            elif (isinstance(q, BooleanQuery) and q.is_flat() and 1 <= len(q.should_queries) <= 9 and q.min_should_match <= 1
                    and not (q.must_queries or q.filter_queries or q.must_not_queries)
                    and not self.should_queries and all(isinstance(m, TermQuery) for m in self.must_queries) and not any(isinstance(m, BooleanQuery) for m in musts)):
                # "+a #(b c)" — a filter by a disjunction ("category is b or c"): the required disjunction "+a +(b c)" whose clauses score
                # 0.0 (needs_scores = false): ConjunctionScorer([a ..., DisjunctionSumScorer(b, c)]) adds (0.0 + 0.0) somewhere in
                # the cost order, which leaves the f32 sum of the scoring clauses as it is. Behind the MUST clauses, where
                # BooleanQuery::create_weight puts the FILTER weights (boolean_query.rs:101-108). One such clause per query.
                musts.append(BooleanQuery([], [TermQuery(t.term, 0.0) for t in q.should_queries], q.min_should_match))
                changed = True
            else:
                filters.append(q)
        return BooleanQuery(musts, list(self.should_queries), self.min_should_match, must_nots, filters) if changed else self

    def is_flat(self):
        return all(isinstance(q, TermQuery) for q in self.must_queries + self.should_queries + self.must_not_queries + self.filter_queries)

    def flattened(self):
        """ONE level of nesting folded into this query, or None when the tree is not of that shape: a MUST clause that is itself a
        must-only BooleanQuery of terms, a SHOULD clause that is a should-only one (min_should_match <= 1) — what query builders that
        AND / OR sub-expressions together produce. The reference does not rewrite such trees (boolean_query.rs:195-279 builds a
        ConjunctionScorer over [a, ConjunctionScorer(b, c)]): doc ids and hit counts are those of the flat query, but its f32 sums
        are formed as a + (b + c) where the flat query forms (a + b) + c — equal within 1e-5 relative (north_star's tolerance for
        floats), NOT bit for bit; a top-k boundary inside such a rounding band may fall differently. Opt-in for that reason."""
        def fold(clauses, want_must):
            out = []
            for q in clauses:
                if isinstance(q, TermQuery):
                    out.append(q)
                    continue
                if not q.is_flat() or q.must_not_queries or q.filter_queries:
                    return None
                if want_must and q.must_queries and not q.should_queries:
                    out.extend(q.must_queries)
                elif not want_must and q.should_queries and not q.must_queries and q.min_should_match <= 1:
                    out.extend(q.should_queries)
                else:
                    return None
            return out
        if not all(isinstance(q, TermQuery) for q in self.must_not_queries + self.filter_queries):
            return None
        musts = fold(self.must_queries, True)
        if musts is None:
            return None
        if musts or self.filter_queries:   # SHOULD clauses beside MUST ones stay as they are (ReqOptScorer's optional side)
            if not all(isinstance(q, TermQuery) for q in self.should_queries):
                return None
            return BooleanQuery(musts, list(self.should_queries), self.min_should_match, self.must_not_queries, self.filter_queries)
        shoulds = fold(self.should_queries, False)
        if shoulds is None or (self.min_should_match > 1 and len(shoulds) != len(self.should_queries)):
            return None   # (min_should_match counts the OUTER clauses: folding would change what it counts)
        return BooleanQuery([], shoulds, self.min_should_match, self.must_not_queries, [])

    def required_disjunction(self):
        """"+a +(b c)": MUST TermQuery clauses and exactly ONE MUST clause that is a should-only BooleanQuery of 1..9 terms
        (min_should_match <= 1), no SHOULD clause of its own -> (that nested query) else None. BooleanWeight::create_scorer builds
        ConjunctionScorer([TermScorer ..., DisjunctionSumScorer]) for it (boolean_query.rs:200-215); the C ABI takes it as
        RGPU_OP_WITH_SHOULD(AND, n) | RGPU_OP_SHOULD_REQUIRED."""
        if self.should_queries or not all(isinstance(q, TermQuery) for q in self.must_not_queries + self.filter_queries):
            return None
        nested = [q for q in self.must_queries if not isinstance(q, TermQuery)]
        if len(nested) != 1:
            return None
        d = nested[0]
        if (not d.is_flat() or d.must_queries or d.must_not_queries or d.filter_queries or d.min_should_match > 1
                or not 1 <= len(d.should_queries) <= 9):
            return None
        return d

    def nested_disjunction_first(self):
        """"(b c) a d" / "a (b c) d": a should-only query whose FIRST or SECOND clause is itself a should-only BooleanQuery of terms ->
        the flat disjunction with the nested clauses moved to the front, else None. DisjunctionSumScorer sums its children in
        clause order from 0.0 (SimpleQueue, below ten children: disjunction_scorer.rs:41-45, 211-225), the nested scorer's own sum
        formed first: (a + (b + c)) + d. The flat query [b, c, a, d] forms ((b + c) + a) + d — the same f32, because the one add
        that differs has two operands and commutes (a doc without a, or without b / c, drops the absent terms from both sums alike).
        Same docs, same hit count, same score bits: no tolerance, no flag. Fewer than ten clauses in all (the clause-order kernel)."""
        if self.must_queries or self.filter_queries or self.min_should_match > 1:
            return None
        if not all(isinstance(q, TermQuery) for q in self.must_not_queries):
            return None
        nested = [i for i, q in enumerate(self.should_queries) if not isinstance(q, TermQuery)]
        if len(nested) != 1 or nested[0] > 1:
            return None
        d = self.should_queries[nested[0]]
        if not d.is_flat() or d.must_queries or d.must_not_queries or d.filter_queries or d.min_should_match > 1 or not d.should_queries:
            return None
        rest = [q for q in self.should_queries if q is not d]
        if len(d.should_queries) + len(rest) >= 10:
            return None
        return BooleanQuery([], list(d.should_queries) + rest, self.min_should_match, self.must_not_queries, [])

    def nested_conjunction(self):
        """"+a +(+b +c)": MUST TermQuery clauses and exactly ONE MUST clause that is a must-only BooleanQuery of >= 2 terms, no SHOULD
        clause of its own -> (that nested query) else None. The reference builds ConjunctionScorer([TermScorer ...,
        ConjunctionScorer(b, c)]) (boolean_query.rs:200-215): the nested sum is formed first. The C ABI takes it as
        RGPU_OP_WITH_SHOULD(AND, n) | RGPU_OP_NESTED_MUST."""
        if self.should_queries or not all(isinstance(q, TermQuery) for q in self.must_not_queries + self.filter_queries):
            return None
        nested = [q for q in self.must_queries if not isinstance(q, TermQuery)]
        if len(nested) != 1:
            return None
        c = nested[0]
        if not c.is_flat() or c.should_queries or c.must_not_queries or c.filter_queries or len(c.must_queries) < 2:
            return None
        return c

    def required_clauses(self):
        """MUST clauses followed by the FILTER clauses as zero-weight MUST clauses (BooleanWeight puts both into must_weights)."""
        return list(self.must_queries) + [TermQuery(f.term, 0.0) for f in self.filter_queries]

    def extract_terms(self):  # boolean_query.rs:124-145: MUST, SHOULD and FILTER clauses only
        return list(self.must_queries) + list(self.should_queries) + list(self.filter_queries)


class TopDocs:
    def __init__(self, total_hits, score_docs):
        self._total, self._docs = int(total_hits), score_docs

    def total_hits(self):
        return self._total

    def score_docs(self):
        """[(doc, score)] best first: score desc, then doc asc (the canonical tie rule)."""
        return self._docs


class TopDocsCollector:
    def __init__(self, estimated_hits):
        self.estimated_hits = int(estimated_hits)
        self._result = TopDocs(0, [])

    def needs_scores(self):
        return True

    def top_docs(self):
        return self._result


class GpuIndexSearcher:
    """IndexSearcher over GPU-resident leaves. `search(query, collector)` mirrors searcher.rs:487-525;
    `search_batch` is the batched form the hardware wants (one launch set per leaf for many queries)."""

    def __init__(self, leaves, ctx=None, similarity=None, next_limit=None, flatten_nested=False, cpu_fallback=None):
        """flatten_nested: fold one level of nested BooleanQuery clauses (BooleanQuery.flattened: same docs and counts, scores
        within 1e-5 of the reference's — off by default because it is not bit-exact). cpu_fallback(query, collector): what
        SURVEY 8(f)1 calls "everything else to the CPU path" — search() hands every tree the GPU path does not serve
        (UnsupportedOperation) to it, as the Rust shim hands them to DefaultIndexSearcher (rust/gpu/searcher.rs); None: raise."""
        self.flatten_nested = bool(flatten_nested)
        self.cpu_fallback = cpu_fallback
        self.leaves = list(leaves)
        # DefaultIndexSearcher::new(reader, next_limit: Option<usize>) (searcher.rs:291-296, :361): how many approximations of a
        # two-phase scorer (here: sloppy phrases) may go by on a leaf without a collected doc. None = the default, 500 000
        if next_limit is not None and int(next_limit) < 0:
            raise RgpuError(-2, "next_limit must be >= 0 (None: the reference's default of 500 000)")
        # (the C ABI spells Some(0) RGPU_NEXT_LIMIT_ZERO = -2: a zero there keeps meaning "the default")
        self.next_limit = 0 if next_limit is None else (-2 if int(next_limit) == 0 else int(next_limit))
        self.ctx = ctx or _lib.Context()
        self.similarity = similarity or BM25Similarity()
        for leaf in self.leaves:
            if leaf.segment is None:
                leaf.segment = _lib.Segment(self.ctx, leaf.doc_bytes, leaf.norms, leaf.max_doc, leaf.doc_base, leaf.live_docs,
                                            leaf.index_options | (_lib.FIELD_STORES_PAYLOADS if getattr(leaf, "has_payloads", False) else 0))
                if getattr(leaf, "pay_bytes", None) is not None:
                    leaf.segment.attach_payloads(leaf.pay_bytes)   # Lucene50PostingsReader::open checks the third file too
        # searcher.rs:306-363: statistics of the first leaf with the largest max_doc stand in for the index
        self._stats_leaf = 0
        for i, leaf in enumerate(self.leaves):
            if leaf.max_doc > self.leaves[self._stats_leaf].max_doc:
                self._stats_leaf = i
        sl = self.leaves[self._stats_leaf]
        self.collection_statistics = CollectionStatistics(sl.field, sl.doc_base, self.max_doc(), sl.doc_count,
                                                          sl.sum_total_term_freq, sl.sum_doc_freq)
        self._weights = {}
        self._planners = {}       # per leaf: the native batch planner
        self._stats_terms = None  # override_statistics: another leaf's term table / dictionary

    def max_doc(self):
        return sum(leaf.max_doc for leaf in self.leaves)

    def term_statistics(self, term_id):
        """searcher.rs:732-767: df of the term in the statistics leaf only (0 when absent there)."""
        stats = getattr(self, "_stats_terms", None)
        if stats is None:
            st = self.leaves[self._stats_leaf].term_state(term_id)
            return 0 if st is None else int(st["doc_freq"])
        if isinstance(stats, _lib.TermDictionary):
            states, found = stats.lookup(self.leaves[self._stats_leaf].field_number, [term_id])
            return int(states[0]["doc_freq"]) if found[0] else 0
        return int(stats[term_id]["doc_freq"]) if 0 <= term_id < len(stats) else 0

    def _weight(self, term_id, boost):
        key = (term_id, boost)
        if key not in self._weights and len(self._weights) >= (1 << 20):
            self._weights.clear()
        if key not in self._weights:
            w, cache = self.similarity.compute_weight(self.collection_statistics, [self.term_statistics(term_id)], boost)
            self._weights[key] = (w, self.ctx.sim_table(cache, self.similarity.k1))
        return self._weights[key]

    def _flatten(self, query):
        """-> (op, required / scored clauses, optional SHOULD clauses beside MUST ones, MUST_NOT clauses)"""
        if isinstance(query, TermQuery):
            return OP_TERM, [query], [], []
        if isinstance(query, BooleanQuery):
            query = query.normalized()
            if not query.is_flat():
                first = query.nested_disjunction_first()
                if first is not None:
                    return self._flatten(first)
                # "+a +(b c)" / "+a +(+b +c)": ONE nested should-only / must-only query among the MUST clauses. The library sorts the
                # conjunction's children by cost per leaf as ConjunctionScorer::new does and adds the nested sum where the reference
                # adds it (RGPU_OP_NESTED_AT breaks ties like the stable sort): the reference's f32 sums, whatever the costs.
                for nested, flag in ((query.required_disjunction(), OP_SHOULD_REQUIRED), (query.nested_conjunction(), OP_NESTED_MUST)):
                    if nested is None:
                        continue
                    at = [q is nested for q in query.must_queries].index(True)
                    musts = [q for q in query.must_queries if q is not nested]
                    if not (musts or query.filter_queries):
                        # (a lone nested MUST clause: BooleanQuery::build has already rewritten such a tree to the clause itself)
                        raise RgpuError(-5, "a nested query with no clause beside it is that query")
                    inner = list(nested.should_queries if flag == OP_SHOULD_REQUIRED else nested.must_queries)
                    required = musts + [TermQuery(f.term, 0.0) for f in query.filter_queries]
                    return _i32(OP_AND | (len(inner) << 16) | flag | (at << 26)), required, inner, query.must_not_queries
                folded = query.flattened() if getattr(self, "flatten_nested", False) else None
                if folded is None:
                    raise RgpuError(-5, "nested boolean clauses are not served by the GPU path (flatten_nested folds one level of MUST-of-MUSTs / SHOULD-of-SHOULDs)")
                query = folded
            required = query.required_clauses()
            if required:
                opts = query.should_queries
                return OP_AND | (len(opts) << 16), required, opts, query.must_not_queries   # RGPU_OP_WITH_SHOULD
            msm = query.min_should_match
            return (OP_OR | (msm << 8) if msm > 1 else OP_OR), query.should_queries, [], query.must_not_queries
        raise RgpuError(-5, "query type not served by the GPU path: %r" % (query,))

    def override_statistics(self, collection_statistics, stats_terms=None):
        """Score with the statistics of a leaf that lives elsewhere (segment-sharded search: every shard takes them from
        shard 0, the index-wide first largest leaf — SURVEY.md §8(e)): `stats_terms` = that leaf's flat term table
        (TERM_STATE_DTYPE, indexed by term id) or its TermDictionary; None keeps this searcher's own statistics leaf."""
        self.collection_statistics = collection_statistics
        self._stats_terms = stats_terms
        self._weights, self._planners = {}, {}

    def _planner(self, leaf):
        """The native batch planner (rgpu_planner, csrc/host/batch_planner.hpp) of one leaf: term resolution in the leaf
        and in the statistics leaf, BM25 weights, one sim table."""
        key = id(leaf)
        p = self._planners.get(key)
        if p is None:
            cs = self.collection_statistics
            stats = self._stats_terms
            if stats is None:
                sl = self.leaves[self._stats_leaf]
                stats = sl.term_dictionary if sl.term_dictionary is not None else sl.terms
            mine = leaf.term_dictionary if leaf.term_dictionary is not None else leaf.terms
            if isinstance(mine, _lib.TermDictionary) != isinstance(stats, _lib.TermDictionary):
                raise RgpuError(-2, "the searched leaf and the statistics leaf must name terms the same way (bytes or ids)")
            _w, cache = self.similarity.compute_weight(cs, [1], 1.0)   # the norm cache depends on the collection alone
            table = self.ctx.sim_table(cache, self.similarity.k1)
            p = _lib.Planner(None, cs.max_doc, cs.doc_count, cs.sum_total_term_freq, mine, None if stats is mine else stats,
                             self.similarity.k1, self.similarity.b, leaf.field_number, sim_table=table)
            self._planners[key] = p
        return p

    def pack_uniform(self, op, term_ids, leaf, min_should_match=0):
        """The planner for a batch handed over as an ARRAY: `term_ids[q, c]` = flat-table id of clause c of query q, every
        query the same shape — `op` OP_TERM (one column), OP_AND (all MUST) or OP_OR (all SHOULD, optionally with
        min_should_match), boost 1. Equals pack([TermQuery / BooleanQuery.build(...) ...]) on the same ids; the whole of it
        runs behind the C ABI (rgpu_plan_uniform_ids)."""
        term_ids = np.asarray(term_ids, dtype=np.int64)
        if term_ids.ndim == 1:
            term_ids = term_ids.reshape(-1, 1)
        nq, nc = term_ids.shape
        if op not in (OP_TERM, OP_AND, OP_OR) or (op == OP_TERM and nc != 1) or nc < 1:
            raise RgpuError(-1, "pack_uniform: op TERM takes one column, AND / OR at least one")
        if nc > _lib.MAX_QUERY_TERMS:
            raise RgpuError(-5, "more than %d clauses" % _lib.MAX_QUERY_TERMS)
        if nc == 1:
            op, min_should_match = OP_TERM, 0   # BooleanQuery::build with a single clause rewrites to that clause (boolean_query.rs:56-68)
        return self._planner(leaf).plan_uniform((op | (min_should_match << 8)) if (op == OP_OR and min_should_match > 1) else op, term_ids)

    def search_uniform_device(self, op, term_ids, leaf, k, hits_ptr, totals_ptr, stream=0, min_should_match=0, comm=None):
        """pack_uniform + rgpu_search_batch_device (comm: rgpu_search_batch_sharded) as ONE call behind the C ABI
        (rgpu_planner_search_uniform_ids_device): the serving path of a batch named by term ids. Enqueue-only; same rows as
        the two calls."""
        term_ids = np.asarray(term_ids, dtype=np.int64)
        if term_ids.ndim == 1:
            term_ids = term_ids.reshape(-1, 1)
        nc = term_ids.shape[1]
        if op not in (OP_TERM, OP_AND, OP_OR) or (op == OP_TERM and nc != 1) or nc < 1:
            raise RgpuError(-1, "search_uniform_device: op TERM takes one column, AND / OR at least one")
        if nc > _lib.MAX_QUERY_TERMS:
            raise RgpuError(-5, "more than %d clauses" % _lib.MAX_QUERY_TERMS)
        if nc == 1:
            op, min_should_match = OP_TERM, 0
        full_op = (op | (min_should_match << 8)) if (op == OP_OR and min_should_match > 1) else op
        self._planner(leaf).search_uniform_device(leaf.segment, full_op, term_ids, k, hits_ptr, totals_ptr, stream, comm=comm)

    def pack(self, queries, leaf):
        """queries -> (rgpu_query[], rgpu_query_term[]) for one leaf."""
        flat = [self._flatten(q) for q in queries]
        clauses = [c for _, t, o, n in flat for group in (t, o, n) for c in group]
        by_id = [type(c.term) is int for c in clauses]
        if clauses and (all(by_id) or not any(by_id)) and all(by_id) == (leaf.term_dictionary is None):
            # one naming scheme throughout (the usual case): the per-clause work — resolution, weights — happens natively
            if max(len(t) + len(o) + len(n) for _, t, o, n in flat) > _lib.MAX_QUERY_TERMS:
                raise RgpuError(-5, "more than %d clauses" % _lib.MAX_QUERY_TERMS)
            boosts = None if all(c.boost == 1.0 for c in clauses) else [c.boost for c in clauses]
            return self._planner(leaf).plan_batch([f[0] for f in flat], [len(f[1]) for f in flat], [c.term for c in clauses],
                                                  [len(f[3]) for f in flat], boosts)
        return self._pack_clause_by_clause(queries, leaf, flat)

    def _pack_clause_by_clause(self, queries, leaf, flat=None):
        """pack() one clause at a time in Python: mixed naming schemes, numpy-scalar ids; also what tests hold the native
        planner against."""
        flat = flat or [self._flatten(q) for q in queries]
        byte_terms = [c.term for _, t, o, n in flat for c in list(t) + list(o) + list(n) if isinstance(c.term, bytes)]
        if byte_terms:
            leaf.resolve(byte_terms)
            self.leaves[self._stats_leaf].resolve(byte_terms)
        n_terms = sum(len(t) + len(o) + len(n) for _, t, o, n in flat)
        qs = np.zeros(len(flat), dtype=QUERY_DTYPE)
        ts = np.zeros(max(n_terms, 1), dtype=QUERY_TERM_DTYPE)
        pos = 0
        for i, (op, clauses, opts, nots) in enumerate(flat):
            if len(clauses) + len(opts) + len(nots) > _lib.MAX_QUERY_TERMS:
                raise RgpuError(-5, "more than %d clauses" % _lib.MAX_QUERY_TERMS)
            qs[i] = (op, len(clauses), pos, len(nots))   # clause order: MUST / scored, optional SHOULD, MUST_NOT
            for c in list(clauses) + list(opts) + list(nots):
                w, table = self._weight(c.term, c.boost)
                st = leaf.term_state(c.term)
                if st is not None:
                    ts[pos]["state"] = st
                else:
                    ts[pos]["state"]["doc_freq"] = 0
                    ts[pos]["state"]["skip_offset"] = -1
                    ts[pos]["state"]["singleton_doc_id"] = -1
                ts[pos]["weight"] = w
                ts[pos]["sim_table"] = table
                pos += 1
        return qs, ts

    def search_phrase_batch(self, queries, k):
        """IndexSearcher::search(PhraseQuery, TopDocsCollector(k)) for a batch of exact phrases -> (hits, total_hits).
        PhraseQuery::create_weight (phrase_query.rs:136-186): one BM25 weight from the statistics of ALL the phrase's terms."""
        per_leaf = []
        for leaf in self.leaves:
            qs, ts = self.pack_phrases(queries, leaf)
            per_leaf.append(leaf.segment.search_phrase_batch(qs, ts, k))
        if len(per_leaf) == 1:
            return per_leaf[0]
        return self._merge_leaves(per_leaf, len(queries), k)

    def pack_phrases(self, queries, leaf):
        """PhraseQuery objects -> the rgpu_phrase_query[] / rgpu_phrase_term[] of rgpu_search_phrase_batch for one leaf (attaches the
        leaf's .pos file on first use)."""
        if leaf.pos_bytes is None:
            raise RgpuError(-1, "phrase search needs a positions field (LeafReader.pos_bytes)")
        if not getattr(leaf, "_pos_attached", False):
            leaf.segment.attach_positions(leaf.pos_bytes)
            leaf._pos_attached = True
        qs = np.zeros(len(queries), dtype=_lib.PHRASE_QUERY_DTYPE)
        ts = np.zeros(sum(len(q.terms) for q in queries), dtype=_lib.PHRASE_TERM_DTYPE)
        at = 0
        for i, q in enumerate(queries):
            w, cache = self.similarity.compute_weight(self.collection_statistics, [self.term_statistics(t) for t in q.terms], q.boost)
            qs[i] = (len(q.terms), at, w, self.ctx.sim_table(cache, self.similarity.k1), q.slop, self.next_limit)
            for t, p in zip(q.terms, q.positions):
                sp = leaf.positions_state(t)
                if sp is not None:
                    ts[at]["state"], ts[at]["positions"] = sp
                else:
                    ts[at]["state"]["doc_freq"] = 0
                ts[at]["position"] = p
                at += 1
        return qs, ts

    def rescore_batch(self, hits, rescore_queries, query_weight=1.0, rescore_weight=1.0, mode=_lib.RESCORE_TOTAL, window_size=None):
        """QueryRescorer::rescore (search/scorer/rescorer.rs:376-390) for a batch: row i of `hits` (a first pass's output) is
        re-ranked by rescore_queries[i] — TermQuery or an all-MUST / all-SHOULD BooleanQuery. One call per leaf, the last
        one sorts the windows and re-weights the tails."""
        k = hits.shape[1]
        req = np.zeros(len(rescore_queries), dtype=_lib.RESCORE_REQUEST_DTYPE)
        req["query_weight"], req["rescore_weight"], req["mode"] = query_weight, rescore_weight, mode
        req["window_size"] = k if window_size is None else window_size
        out = hits
        for i, leaf in enumerate(self.leaves):
            qs, ts = self.pack(rescore_queries, leaf)
            out = leaf.segment.rescore_batch(qs, ts, req, out, finish=(i == len(self.leaves) - 1))
        return out

    def search_batch(self, queries, k):
        """-> (hits[n][k] structured {doc, score}, total_hits[n]) merged over all leaves."""
        per_leaf = []
        for leaf in self.leaves:
            qs, ts = self.pack(queries, leaf)
            per_leaf.append(leaf.segment.search_batch(qs, ts, k))
        if len(per_leaf) == 1:
            return per_leaf[0]
        return self._merge_leaves(per_leaf, len(queries), k)

    def _merge_leaves(self, per_leaf, n_queries, k):
        # TopDocsCollector::finish_parallel (top_docs.rs:157-172) on the device: [leaf][query][k] -> [query][k]
        import torch
        hits = torch.from_numpy(np.stack([h.view(np.int64).reshape(n_queries, k) for h, _ in per_leaf])).cuda()
        totals = torch.from_numpy(np.stack([t for _, t in per_leaf])).cuda()
        out_h = torch.empty((n_queries, k), dtype=torch.int64, device="cuda")
        out_t = torch.empty((n_queries,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        self.ctx.merge_topk_device(hits.data_ptr(), totals.data_ptr(), len(per_leaf), n_queries, k, out_h.data_ptr(), out_t.data_ptr())
        self.ctx.synchronize()  # the merge is only enqueued (on the ctx's stream)
        return out_h.cpu().numpy().view(_lib.HIT_DTYPE).reshape(n_queries, k), out_t.cpu().numpy()

    def search(self, query, collector):
        """IndexSearcher::search(query, collector) for a TopDocsCollector."""
        try:
            if not isinstance(collector, TopDocsCollector):
                raise RgpuError(-5, "only TopDocsCollector is served by the GPU path")
            hits, totals = self.search_batch([query], collector.estimated_hits)
        except RgpuError as e:
            # ErrorKind::UnsupportedOperation -> the CPU searcher, exactly where rust/gpu/searcher.rs falls back to DefaultIndexSearcher
            if e.status == -5 and self.cpu_fallback is not None:
                return self.cpu_fallback(query, collector)
            raise
        row = hits[0]
        docs = [(int(d), float(s)) for d, s in zip(row["doc"], row["score"]) if d >= 0]
        collector._result = TopDocs(int(totals[0]), docs)
