// C++ host-side mirror of the slice of Rucene's search API served by the GPU path, header-only over the C ABI
// (include/rucene_gpu.h). The reference is compiled Rust and this image has no Rust toolchain, so the layer a
// Rucene maintainer would write as a Rust shim is provided in C++ with the reference's names,
// argument meaning and error behaviour (paths relative to /root/reference/src/core):
//
//   search/searcher.rs:205-249, 306-363, 487-525, 732-767   IndexSearcher::search + the largest-leaf statistics
//   search/query/term_query.rs:45-95                         TermQuery::new(term, boost), create_weight
//   search/query/boolean_query.rs:40-86                      BooleanQuery::build
//   search/collector/top_docs.rs:97-183                      TopDocsCollector::new(k), top_docs()
//   search/sort_field/collapse_top_docs.rs:22-36             ScoreDoc
//   error.rs:24-91                                           ErrorKind -> rucene::Error{kind}
//
// Terms are named either by bytes — resolved per leaf through its block-tree dictionary (rgpu_terms_*, the
// TermIterator::seek_exact + term_state step of TermWeight::create_scorer, term_query.rs:150-180) — or, for synthetic
// indexes, by an id into a flat table of BlockTermState records.
#pragma once
#include <dirent.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/rucene_gpu.h"
#include "bm25_similarity.hpp"

namespace rucene {

struct Error : std::runtime_error {
  int kind;  // rgpu_status == error.rs ErrorKind
  Error(int k, const std::string& m) : std::runtime_error(m), kind(k) {}
};
inline void check(int32_t rc) {
  if (rc < 0) throw Error(rc, rgpu_last_error(nullptr));
}

struct ScoreDoc {
  int32_t doc;
  float score;
};

class TopDocs {
 public:
  TopDocs() = default;
  TopDocs(int64_t total, std::vector<ScoreDoc> docs) : total_(total), docs_(std::move(docs)) {}
  int64_t total_hits() const { return total_; }
  const std::vector<ScoreDoc>& score_docs() const { return docs_; }  // score desc, then doc asc

 private:
  int64_t total_ = 0;
  std::vector<ScoreDoc> docs_;
};

class TopDocsCollector {
 public:
  explicit TopDocsCollector(size_t estimated_hits) : estimated_hits_(estimated_hits) {}
  bool needs_scores() const { return true; }
  size_t estimated_hits() const { return estimated_hits_; }
  TopDocs top_docs() const { return result_; }
  void set_result(TopDocs r) { result_ = std::move(r); }

 private:
  size_t estimated_hits_;
  TopDocs result_;
};

struct Query {
  virtual ~Query() {}
};
struct TermQuery : Query {
  int64_t term = -1;   // id into LeafReader::terms, or -1 when the term is named by bytes
  std::string text;    // Term::bytes
  float boost;
  explicit TermQuery(int64_t t, float b = 1.0f) : term(t), boost(b) {}
  explicit TermQuery(std::string bytes, float b = 1.0f) : text(std::move(bytes)), boost(b) {}
  bool by_text() const { return term < 0; }
};
struct BooleanQuery : Query {
  std::vector<TermQuery> must_queries, should_queries, must_not_queries;
  int32_t min_should_match = 0;
  // RGPU_OP_SHOULD_REQUIRED: the SHOULD clauses are a should-only BooleanQuery nested under MUST ("+a +(b c)") — set by
  // NestedBooleanQuery::required_disjunction, never by build()
  bool should_required = false;
  // RGPU_OP_NESTED_MUST: the SHOULD slots hold a must-only BooleanQuery nested under MUST ("+a +(+b +c)") — set by
  // NestedBooleanQuery::nested_conjunction
  bool nested_must = false;
  int32_t nested_at = 0;  // RGPU_OP_NESTED_AT: how many of the MUST term clauses stood before the nested clause in the caller's query
  // boolean_query.rs:40-86 restricted to what the GPU path serves: SHOULD-only term trees, or MUST clauses with optional
  // SHOULD clauses beside them (ReqOptScorer, its sequential skipping rule included: see RGPU_OP_WITH_SHOULD), each
  // optionally with MUST_NOT term clauses (ReqNotScorer); a single clause without MUST_NOTs
  // collapses to that clause
  // FILTER clauses are required clauses that score 0 (create_weight with needs_scores = false -> NonScoringSimilarity,
  // boolean_query.rs:106-108, searcher.rs:158-202): they join the MUST clauses with boost 0, which leaves every sum unchanged
  static std::unique_ptr<Query> build(std::vector<TermQuery> musts, std::vector<TermQuery> shoulds, int32_t min_should_match = 0,
                                      std::vector<TermQuery> must_nots = {}, std::vector<TermQuery> filters = {}) {
    const int32_t msm = min_should_match > 0 ? min_should_match : (musts.empty() ? 1 : 0);
    if (musts.empty() && shoulds.empty() && must_nots.empty() && filters.empty())
      throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "boolean query should at least contain one inner query!");
    for (TermQuery& f : filters) f.boost = 0.0f;
    if (must_nots.empty() && musts.size() + shoulds.size() + filters.size() == 1)  // a lone FILTER: ConstantScoreQuery, boost 0
      return std::unique_ptr<Query>(new TermQuery(!filters.empty() ? filters[0] : (musts.empty() ? shoulds[0] : musts[0])));
    musts.insert(musts.end(), filters.begin(), filters.end());
    if (msm > 255) throw Error(RGPU_ERR_UNSUPPORTED, "min_should_match above 255");
    // (beside MUST clauses min_should_match has no effect — ReqOptScorer only advances the optional scorer — and a tree of
    // MUST_NOT clauses only matches nothing: BooleanWeight::create_scorer -> None)
    auto q = std::unique_ptr<BooleanQuery>(new BooleanQuery());
    q->must_queries = std::move(musts);
    q->should_queries = std::move(shoulds);
    q->must_not_queries = std::move(must_nots);
    q->min_should_match = q->must_queries.empty() ? msm : 0;
    return std::unique_ptr<Query>(q.release());
  }
};

// A BooleanQuery whose MUST / SHOULD clauses may themselves be queries (BooleanQuery::build takes Vec<Box<dyn Query>>,
// boolean_query.rs:40-86). The GPU path serves flat trees of term clauses; `flattened()` folds ONE level — a MUST clause that is
// a must-only BooleanQuery, a SHOULD clause that is a should-only one (min_should_match <= 1) — into a flat BooleanQuery, or
// returns null when the tree is not of that shape. The reference does not rewrite such trees: it sums a + (b + c) where the
// flat query sums (a + b) + c — the same docs and hit counts, scores equal within 1e-5 relative (north_star's float
// tolerance), not bit for bit. GpuIndexSearcher folds only when `flatten_nested` is set; everything else goes to `cpu_fallback`
// ("everything else to the CPU path"; the Rust shim: rust/gpu/searcher.rs).
struct NestedBooleanQuery : Query {
  std::vector<std::unique_ptr<Query>> must_queries, should_queries;
  std::vector<TermQuery> must_not_queries;
  int32_t min_should_match = 0;
  // Nested clauses that do NOT score have exact flat forms (same docs, counts and f32 sums as the reference's scorer tree):
  //   * a MUST_NOT clause that is a should-only BooleanQuery of terms, "-(b c)": ReqNotScorer excludes what the nested
  //     DisjunctionSumScorer matches, b or c — the MUST_NOT clauses b, c (boolean_query.rs:236-252 builds one disjunction over the
  //     MUST_NOT scorers with the OUTER min_should_match: equal only while that is <= 1, so msm > 1 is not expanded);
  //   * a FILTER clause that is a must-only BooleanQuery of terms, "#(+b +c)": its weights are created with needs_scores = false
  //     (boolean_query.rs:106-108), every clause scores 0.0 and the conjunction's 0.0 + 0.0 joins the outer sum as one 0.0 — the
  //     FILTER clauses b, c (MUST clauses of boost 0; x + 0.0 == x wherever the cost order puts them).
  std::vector<std::unique_ptr<Query>> must_not_nested, filter_nested;
  // -> MUST_NOT terms (the query's own + the expanded ones) and the zero-boost MUST terms the nested FILTER clauses stand for; false:
  // a nested non-scoring clause of another shape (the tree goes to cpu_fallback)
  bool scoring_clauses_are_terms() const {
    for (const auto* v : {&must_queries, &should_queries})
      for (const auto& q : *v)
        if (!dynamic_cast<const TermQuery*>(q.get())) return false;
    return true;
  }
  // `filter_disjunction` (optional): a FILTER clause that is a should-only BooleanQuery of 1..9 terms, "+a #(b c)" — served by
  // required_disjunction() as the required disjunction of zero-boost clauses; at most one, and only where the caller asks for it
  bool expand_non_scoring(std::vector<TermQuery>* nots, std::vector<TermQuery>* zeros, const BooleanQuery** filter_disjunction = nullptr) const {
    *nots = must_not_queries;
    zeros->clear();
    for (const auto& q : must_not_nested) {
      if (auto* t = dynamic_cast<const TermQuery*>(q.get())) { nots->push_back(*t); continue; }
      auto* b = dynamic_cast<const BooleanQuery*>(q.get());
      if (!b || min_should_match > 1 || !b->must_queries.empty() || !b->must_not_queries.empty() || b->should_queries.empty() || b->min_should_match > 1 ||
          b->should_required || b->nested_must)
        return false;
      nots->insert(nots->end(), b->should_queries.begin(), b->should_queries.end());
    }
    for (const auto& q : filter_nested) {
      std::vector<TermQuery> terms;
      if (auto* t = dynamic_cast<const TermQuery*>(q.get())) terms.push_back(*t);
      else {
        auto* b = dynamic_cast<const BooleanQuery*>(q.get());
        if (!b || !b->must_not_queries.empty() || b->should_required || b->nested_must) return false;
        if (b->must_queries.empty()) {  // should-only: a filter by a disjunction
          if (!filter_disjunction || *filter_disjunction || b->should_queries.empty() || b->should_queries.size() > 9 || b->min_should_match > 1) return false;
          *filter_disjunction = b;
          continue;
        }
        if (!b->should_queries.empty()) return false;
        terms = b->must_queries;
      }
      for (TermQuery& t : terms) { t.boost = 0.0f; zeros->push_back(t); }
    }
    return true;
  }
  std::unique_ptr<Query> flattened() const {
    auto fold = [](const std::vector<std::unique_ptr<Query>>& clauses, bool want_must, std::vector<TermQuery>* out) -> bool {
      for (const auto& q : clauses) {
        if (auto* t = dynamic_cast<const TermQuery*>(q.get())) { out->push_back(*t); continue; }
        auto* b = dynamic_cast<const BooleanQuery*>(q.get());
        if (!b || !b->must_not_queries.empty()) return false;
        if (want_must && !b->must_queries.empty() && b->should_queries.empty()) out->insert(out->end(), b->must_queries.begin(), b->must_queries.end());
        else if (!want_must && b->must_queries.empty() && !b->should_queries.empty() && b->min_should_match <= 1)
          out->insert(out->end(), b->should_queries.begin(), b->should_queries.end());
        else return false;
      }
      return true;
    };
    std::vector<TermQuery> musts, shoulds, nots, zeros;
    if (!expand_non_scoring(&nots, &zeros)) return nullptr;
    if (!fold(must_queries, true, &musts)) return nullptr;
    musts.insert(musts.end(), zeros.begin(), zeros.end());
    if (!musts.empty()) {  // SHOULD clauses beside MUST ones stay as they are: ReqOptScorer's optional side takes term clauses only
      for (const auto& q : should_queries) {
        auto* t = dynamic_cast<const TermQuery*>(q.get());
        if (!t) return nullptr;
        shoulds.push_back(*t);
      }
    } else {
      if (!fold(should_queries, false, &shoulds)) return nullptr;
      if (min_should_match > 1 && shoulds.size() != should_queries.size()) return nullptr;  // it counts the OUTER clauses
    }
    return BooleanQuery::build(std::move(musts), std::move(shoulds), min_should_match, std::move(nots));
  }
  // "(b c) a d" / "a (b c) d": a should-only query whose FIRST or SECOND clause is itself a should-only BooleanQuery of terms -> the
  // flat disjunction with the nested clauses moved to the front, else null. DisjunctionSumScorer sums its children in clause order
  // from 0.0 (SimpleQueue, below ten children: disjunction_scorer.rs:41-45, 211-225), the nested scorer's own sum formed first:
  // (a + (b + c)) + d; the flat [b, c, a, d] forms ((b + c) + a) + d — the same f32: the one add that differs has two operands
  // and commutes. Same docs, counts and score bits: no tolerance, no flag. Fewer than ten clauses in all (the clause-order kernel).
  std::unique_ptr<Query> nested_disjunction_first() const {
    if (!must_queries.empty() || min_should_match > 1) return nullptr;
    std::vector<TermQuery> inner, rest, nots, zeros;
    if (!expand_non_scoring(&nots, &zeros) || !zeros.empty()) return nullptr;  // (a FILTER clause makes it a conjunction)
    for (size_t i = 0; i < should_queries.size(); ++i) {
      if (auto* t = dynamic_cast<const TermQuery*>(should_queries[i].get())) { rest.push_back(*t); continue; }
      auto* b = dynamic_cast<const BooleanQuery*>(should_queries[i].get());
      if (!b || !inner.empty() || i > 1 || !b->must_queries.empty() || !b->must_not_queries.empty() || b->min_should_match > 1 ||
          b->should_queries.empty())
        return nullptr;
      inner = b->should_queries;
    }
    if (inner.empty() || inner.size() + rest.size() >= 10) return nullptr;
    inner.insert(inner.end(), rest.begin(), rest.end());
    return BooleanQuery::build({}, std::move(inner), min_should_match, std::move(nots));
  }
  // "+a +(b c)": MUST term clauses and exactly ONE MUST clause that is a should-only BooleanQuery of 1..9 terms
  // (min_should_match <= 1), no SHOULD clause of its own -> the tree as MUST clauses + required SHOULD clauses
  // (RGPU_OP_WITH_SHOULD(AND, n) | RGPU_OP_SHOULD_REQUIRED), else null. BooleanWeight::create_scorer builds
  // ConjunctionScorer([TermScorer ..., DisjunctionSumScorer]) for it (boolean_query.rs:200-215). Whether the kernel's sum — MUST
  // sum + disjunction sum — is the reference's, bit for bit, depends on the children's costs: the library's per-leaf sort (RGPU_OP_NESTED_AT).
  std::unique_ptr<BooleanQuery> required_disjunction() const {
    if (!should_queries.empty()) return nullptr;
    std::unique_ptr<BooleanQuery> out(new BooleanQuery());
    const BooleanQuery* nested = nullptr;
    for (const auto& q : must_queries) {
      if (auto* t = dynamic_cast<const TermQuery*>(q.get())) { out->must_queries.push_back(*t); continue; }
      auto* b = dynamic_cast<const BooleanQuery*>(q.get());
      if (!b || nested) return nullptr;
      nested = b;
      out->nested_at = static_cast<int32_t>(out->must_queries.size());
    }
    std::vector<TermQuery> zeros;
    const BooleanQuery* by_filter = nullptr;
    if (!expand_non_scoring(&out->must_not_queries, &zeros, &by_filter)) return nullptr;
    if (by_filter) {  // "+a #(b c)": the required disjunction of zero-boost clauses (they score 0.0: needs_scores = false), behind the MUST
      if (nested) return nullptr;  // clauses, where BooleanQuery::create_weight puts the FILTER weights (boolean_query.rs:101-108)
      nested = by_filter;
      out->nested_at = static_cast<int32_t>(out->must_queries.size());
    }
    out->must_queries.insert(out->must_queries.end(), zeros.begin(), zeros.end());  // behind every MUST term: nested_at stands
    if (!nested || out->must_queries.empty() || !nested->must_queries.empty() || !nested->must_not_queries.empty() || nested->should_required ||
        nested->min_should_match > 1 || nested->should_queries.empty() || nested->should_queries.size() > 9)
      return nullptr;
    out->should_queries = nested->should_queries;
    if (by_filter) for (TermQuery& t : out->should_queries) t.boost = 0.0f;
    out->should_required = true;
    return out;
  }
  // "+a +(+b +c)": MUST term clauses and exactly ONE MUST clause that is a must-only BooleanQuery of >= 2 terms, no SHOULD clause of
  // its own -> MUST clauses + the nested clauses in the SHOULD slots with nested_must set (RGPU_OP_WITH_SHOULD(AND, n) |
  // RGPU_OP_NESTED_MUST), else null. The reference sums the nested conjunction first (conjunction_scorer.rs:87-95): bit for bit
  // (the library sorts the children per leaf), where the flat fold (flattened()) is within 1e-5.
  std::unique_ptr<BooleanQuery> nested_conjunction() const {
    if (!should_queries.empty()) return nullptr;
    std::unique_ptr<BooleanQuery> out(new BooleanQuery());
    const BooleanQuery* nested = nullptr;
    for (const auto& q : must_queries) {
      if (auto* t = dynamic_cast<const TermQuery*>(q.get())) { out->must_queries.push_back(*t); continue; }
      auto* b = dynamic_cast<const BooleanQuery*>(q.get());
      if (!b || nested) return nullptr;
      nested = b;
      out->nested_at = static_cast<int32_t>(out->must_queries.size());
    }
    std::vector<TermQuery> zeros;
    if (!expand_non_scoring(&out->must_not_queries, &zeros)) return nullptr;
    out->must_queries.insert(out->must_queries.end(), zeros.begin(), zeros.end());  // behind every MUST term: nested_at stands
    if (!nested || out->must_queries.empty() || !nested->should_queries.empty() || !nested->must_not_queries.empty() || nested->should_required ||
        nested->nested_must || nested->must_queries.size() < 2)
      return nullptr;
    out->should_queries = nested->must_queries;
    out->nested_must = true;
    return out;
  }
};

// PhraseQuery (query/phrase_query.rs:60-130: PhraseQuery::build numbers the terms' positions 0, 1, 2, ...;
// explicit positions leave gaps). Needs a positions field (LeafReader::index_options == 3 with pos_bytes).
struct PhraseQuery : Query {
  std::vector<TermQuery> terms;
  std::vector<int32_t> positions;
  float boost;
  int32_t slop;  // 0: ExactPhraseScorer; > 0: SloppyPhraseScorer (phrase_query.rs:312-331)
  explicit PhraseQuery(std::vector<TermQuery> t, std::vector<int32_t> pos = {}, float b = 1.0f, int32_t slop_ = 0)
      : terms(std::move(t)), positions(std::move(pos)), boost(b), slop(slop_) {
    if (slop < 0) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "Slop must be >= 0");
    if (positions.empty()) for (size_t i = 0; i < terms.size(); ++i) positions.push_back(static_cast<int32_t>(i));
    if (terms.size() < 2 || terms.size() != positions.size())
      throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "a phrase needs two or more terms, one position each (a one-term phrase is a TermQuery)");
  }
};

// RescoreRequest (search/scorer/rescorer.rs:67-116): how a second query's score is folded into the first pass's
struct RescoreRequest {
  const Query* query = nullptr;  // TermQuery, or an all-MUST / all-SHOULD BooleanQuery
  float query_weight = 1.0f, rescore_weight = 1.0f;
  rgpu_rescore_mode mode = RGPU_RESCORE_TOTAL;
  int32_t window_size = 0;  // 0: the whole row
};

// One segment: postings file, norms, live docs, FieldReader statistics, and the terms: a block-tree dictionary
// (rgpu_terms_open over the segment's .tim/.tip) and/or a flat term table.
struct LeafReader {
  int32_t index_options = 2;  // doc::IndexOptions ordinal of the searched field: 1 Docs, 2 DocsAndFreqs
  const uint8_t* doc_bytes = nullptr;
  size_t doc_len = 0;
  const uint8_t* norms = nullptr;
  const uint64_t* live_docs = nullptr;
  int32_t max_doc = 0, doc_base = 0;
  int64_t doc_count = 0, sum_total_term_freq = 0, sum_doc_freq = -1;
  const rgpu_term_state* terms = nullptr;
  int64_t n_terms = 0;
  const rgpu_terms* dictionary = nullptr;  // not owned
  int32_t field_number = 0;
  // a positions field (index_options == 3): the .pos file and, for a flat term table, each term's position pointers
  const uint8_t* pos_bytes = nullptr;
  size_t pos_len = 0;
  const rgpu_term_positions* term_positions = nullptr;
  rgpu_segment* segment = nullptr;  // filled by the searcher
  // TermIterator::seek_exact + term_state(); false when the term is absent from this leaf
  bool term_state(const TermQuery& q, rgpu_term_state* out) const {
    if (q.by_text()) {
      if (!dictionary) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "this leaf has no term dictionary: query it by term id");
      const int64_t offs[2] = {0, static_cast<int64_t>(q.text.size())};
      uint8_t found = 0;
      check(rgpu_terms_lookup(dictionary, field_number, reinterpret_cast<const uint8_t*>(q.text.data()), offs, 1, out, &found));
      return found != 0;
    }
    if (q.term >= n_terms || terms[q.term].doc_freq <= 0) return false;
    *out = terms[q.term];
    return true;
  }
  // the same plus the position-stream pointers of BlockTermState (lucene50_decode_term, posting_reader.rs:264-306)
  bool positions_state(const TermQuery& q, rgpu_term_state* out, rgpu_term_positions* pos) const {
    if (q.by_text()) {
      if (!dictionary) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "this leaf has no term dictionary: query it by term id");
      const int64_t offs[2] = {0, static_cast<int64_t>(q.text.size())};
      uint8_t found = 0;
      check(rgpu_terms_lookup_positions(dictionary, field_number, reinterpret_cast<const uint8_t*>(q.text.data()), offs, 1, out, pos, &found));
      return found != 0;
    }
    if (!term_state(q, out) || !term_positions) return false;
    *pos = term_positions[q.term];
    return true;
  }
};

// StandardDirectoryReader::open for the slice this path needs (index/reader/directory_reader.rs:90-140, segment_reader.rs):
// the newest commit point segments_N names the segments; per segment .si gives max_doc, .fnm the field's number, then
// _Lucene50_0.{doc,tim,tip}, .nvm/.nvd and _<delgen>.liv — taken out of .cfs for a compound segment. Owns every buffer the
// LeafReaders point into; hand `leaves()` to a GpuIndexSearcher and keep this object alive as long as that searcher.
class IndexDirectory {
 public:
  static std::unique_ptr<IndexDirectory> open(const std::string& path, const std::string& field) {
    std::unique_ptr<IndexDirectory> dir(new IndexDirectory());
    int64_t gen = -1;
    {  // find_segment_file: the highest generation present
      DIR* d = opendir(path.c_str());
      if (!d) throw Error(RGPU_ERR_IO, "cannot list " + path);
      while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.compare(0, 9, "segments_") == 0 && name.size() > 9) gen = std::max<int64_t>(gen, (int64_t)std::stoll(name.substr(9), nullptr, 36));
      }
      closedir(d);
    }
    if (gen < 0) throw Error(RGPU_ERR_IO, "no segments_N file found in " + path);
    const std::vector<uint8_t> commit_bytes = slurp(path + "/segments_" + base36((uint64_t)gen));
    int32_t n = rgpu_commit_from_segments_file(commit_bytes.data(), commit_bytes.size(), gen, nullptr, 0);
    check(n);
    std::vector<rgpu_commit_segment> commit((size_t)n);
    check(rgpu_commit_from_segments_file(commit_bytes.data(), commit_bytes.size(), gen, commit.data(), n));
    int32_t doc_base = 0;
    for (const rgpu_commit_segment& seg : commit) {
      dir->segments_.emplace_back(new Segment());
      Segment& s = *dir->segments_.back();
      const std::string name = seg.name, stem = path + "/" + name;
      const std::vector<uint8_t> si = slurp(stem + ".si");
      rgpu_segment_info info;
      check(rgpu_segment_info_from_lucene62(si.data(), si.size(), seg.id, &info));
      if (seg.del_count > info.max_doc) throw Error(RGPU_ERR_CORRUPT_INDEX, "invalid deletion count");
      std::vector<uint8_t> cfs;
      std::vector<rgpu_compound_entry> entries;
      if (info.is_compound_file) {
        cfs = slurp(stem + ".cfs");
        const std::vector<uint8_t> cfe = slurp(stem + ".cfe");
        int32_t ne = rgpu_compound_entries_from_lucene50(cfe.data(), cfe.size(), cfs.data(), cfs.size(), seg.id, nullptr, 0);
        check(ne);
        entries.resize((size_t)ne);
        check(rgpu_compound_entries_from_lucene50(cfe.data(), cfe.size(), cfs.data(), cfs.size(), seg.id, entries.data(), ne));
      }
      auto part = [&](const std::string& suffix) -> std::vector<uint8_t> {
        if (!info.is_compound_file) return slurp(stem + suffix);
        for (const rgpu_compound_entry& e : entries)
          if (suffix == e.id) return std::vector<uint8_t>(cfs.begin() + e.offset, cfs.begin() + e.offset + e.length);
        throw Error(RGPU_ERR_IO, name + suffix + " is not in the compound file");
      };
      const std::vector<uint8_t> fnm = seg.field_infos_gen > 0 ? slurp(stem + "_" + base36((uint64_t)seg.field_infos_gen) + ".fnm") : part(".fnm");
      size_t names_len = 0;
      int32_t nf = rgpu_field_infos_from_lucene60(fnm.data(), fnm.size(), nullptr, 0, nullptr, 0, &names_len);
      check(nf);
      std::vector<rgpu_field_info> infos((size_t)nf);
      std::vector<char> names(names_len + 1);
      check(rgpu_field_infos_from_lucene60(fnm.data(), fnm.size(), infos.data(), nf, names.data(), names_len, &names_len));
      int32_t field_number = -1, index_options = 0;
      std::vector<rgpu_field_info> indexed;
      const char* p = names.data();
      for (int32_t i = 0; i < nf; ++i, p += std::strlen(p) + 1) {
        if (infos[(size_t)i].index_options != 0) indexed.push_back(infos[(size_t)i]);
        if (field == p) { field_number = infos[(size_t)i].number; index_options = infos[(size_t)i].index_options; }
      }
      if (field_number < 0) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "no field named " + field + " in segment " + name);
      if (index_options != 1 && index_options != 2)
        throw Error(RGPU_ERR_UNSUPPORTED, "the searched field must be indexed with IndexOptions::Docs or ::DocsAndFreqs");
      s.doc = part("_Lucene50_0.doc");
      const std::vector<uint8_t> tim = part("_Lucene50_0.tim"), tip = part("_Lucene50_0.tip"), nvm = part(".nvm"), nvd = part(".nvd");
      check(rgpu_terms_open(tim.data(), tim.size(), tip.data(), tip.size(), indexed.data(), (int32_t)indexed.size(), info.max_doc, &s.terms));
      s.norms.resize((size_t)info.max_doc);
      check(rgpu_norms_from_lucene53(nvm.data(), nvm.size(), nvd.data(), nvd.size(), field_number, info.max_doc, s.norms.data()));
      if (seg.del_gen >= 0 && seg.del_count > 0) {
        const std::vector<uint8_t> liv = slurp(stem + "_" + base36((uint64_t)seg.del_gen) + ".liv");
        s.live.resize((size_t)((info.max_doc + 63) / 64));
        check(rgpu_live_docs_from_lucene50(liv.data(), liv.size(), info.max_doc, seg.del_count, s.live.data()));
      }
      rgpu_field_stats stats;
      check(rgpu_terms_field_stats(s.terms, field_number, &stats));
      LeafReader leaf;
      leaf.doc_bytes = s.doc.data();
      leaf.doc_len = s.doc.size();
      leaf.norms = s.norms.data();
      leaf.live_docs = s.live.empty() ? nullptr : s.live.data();
      leaf.max_doc = info.max_doc;
      leaf.doc_base = doc_base;
      leaf.doc_count = stats.doc_count;
      leaf.sum_total_term_freq = stats.sum_total_term_freq;
      leaf.sum_doc_freq = stats.sum_doc_freq;
      leaf.dictionary = s.terms;
      leaf.field_number = field_number;
      leaf.index_options = index_options;
      dir->leaves_.push_back(leaf);
      doc_base += info.max_doc;
    }
    return dir;
  }
  const std::vector<LeafReader>& leaves() const { return leaves_; }

 private:
  struct Segment {
    std::vector<uint8_t> doc, norms;
    std::vector<uint64_t> live;
    rgpu_terms* terms = nullptr;
    ~Segment() { rgpu_terms_close(terms); }
  };
  std::vector<std::unique_ptr<Segment>> segments_;
  std::vector<LeafReader> leaves_;

  static std::vector<uint8_t> slurp(const std::string& file) {
    std::ifstream f(file, std::ios::binary);
    if (!f) throw Error(RGPU_ERR_IO, "cannot open " + file);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  static std::string base36(uint64_t v) {  // util/numeric.rs:148-160
    std::string r;
    do { r.insert(r.begin(), "0123456789abcdefghijklmnopqrstuvwxyz"[v % 36]); v /= 36; } while (v);
    return r;
  }
};

class GpuIndexSearcher {
 public:
  // DefaultIndexSearcher::new(reader, next_limit) (searcher.rs:291-296): approximations a two-phase scorer (a sloppy phrase)
  // may spend on a leaf without a collected doc before the leaf is abandoned; 0 = the reference's default of 500 000, -1 = none
  int32_t next_limit = 0;
  // `config`: the library's knobs (rgpu_config: byte budgets for the doc bitmaps and the prepared-term store, enqueue-only
  // disjunction batches, work partitioning ...); null = the defaults. abi_version is filled in here.
  GpuIndexSearcher(std::vector<LeafReader> leaves, const BM25Similarity& sim = BM25Similarity(), int device = 0,
                   const rgpu_config* config = nullptr)
      : leaves_(std::move(leaves)), sim_(sim) {
    rgpu_config cfg{};
    if (config) cfg = *config;
    cfg.abi_version = RGPU_ABI_VERSION;
    check(rgpu_init(device, &cfg, &ctx_));
    for (auto& l : leaves_) {
      check(rgpu_segment_upload_field(ctx_, l.doc_bytes, l.doc_len, l.norms, l.max_doc, l.doc_base, l.live_docs, l.index_options, &l.segment));
      if (l.pos_bytes) check(rgpu_segment_attach_positions(l.segment, l.pos_bytes, l.pos_len));
    }
    // searcher.rs:306-363: the first leaf with the largest max_doc provides the collection statistics
    for (size_t i = 1; i < leaves_.size(); ++i)
      if (leaves_[i].max_doc > leaves_[stats_leaf_].max_doc) stats_leaf_ = i;
    const LeafReader& s = leaves_[stats_leaf_];
    stats_.doc_base = s.doc_base;
    stats_.max_doc = max_doc();
    stats_.doc_count = s.doc_count;
    stats_.sum_total_term_freq = s.sum_total_term_freq;
    stats_.sum_doc_freq = s.sum_doc_freq;
  }
  ~GpuIndexSearcher() {
    for (rgpu_planner* p : planners_) rgpu_planner_destroy(p);
    for (auto& l : leaves_) rgpu_segment_free(l.segment);
    rgpu_shutdown(ctx_);
  }
  GpuIndexSearcher(const GpuIndexSearcher&) = delete;
  GpuIndexSearcher& operator=(const GpuIndexSearcher&) = delete;

  int64_t max_doc() const {
    int64_t n = 0;
    for (auto& l : leaves_) n += l.max_doc;
    return n;
  }
  // searcher.rs:732-767: doc_freq of the term in the statistics leaf only
  TermStatistics term_statistics(const TermQuery& term) const {
    TermStatistics ts;
    rgpu_term_state st;
    const bool have = leaves_[stats_leaf_].term_state(term, &st);
    ts.doc_freq = have ? st.doc_freq : 0;
    ts.total_term_freq = have ? st.total_term_freq : 0;
    return ts;
  }
  const CollectionStatistics& collection_statistics() const { return stats_; }

  // Trees the GPU path does not serve: fold one level of nesting (same docs, scores within 1e-5 — NestedBooleanQuery), and / or
  // hand the query to the host's CPU searcher — where rust/gpu/searcher.rs calls DefaultIndexSearcher::search
  bool flatten_nested = false;
  std::function<void(const Query&, TopDocsCollector&)> cpu_fallback;

  // IndexSearcher::search(query, collector) for a TopDocsCollector
  void search(const Query& query, TopDocsCollector& collector) {
    try {
      std::unique_ptr<Query> folded;
      const Query* q = &query;
      if (auto* nested = dynamic_cast<const NestedBooleanQuery*>(&query)) {
        std::unique_ptr<BooleanQuery> req = nested->required_disjunction();
        // (the library sorts a conjunction's children by cost per leaf and adds a nested child's sum where ConjunctionScorer::score
        // adds it — RGPU_OP_NESTED_AT breaks ties like the stable sort: the reference's f32 sums whatever the costs)
        if ((folded = nested->nested_disjunction_first())) {}  // exact as a flat disjunction in another clause order
        else if (req) folded = std::move(req);
        else if ((req = nested->nested_conjunction())) folded = std::move(req);
        else if (flatten_nested || nested->scoring_clauses_are_terms()) folded = nested->flattened();  // (only non-scoring clauses nested: exact)  // (what is left — two nested clauses, a disjunction from the third SHOULD clause on — within 1e-5)
        if (!folded) throw Error(RGPU_ERR_UNSUPPORTED, "nested boolean clauses are not served by the GPU path");
        q = folded.get();
      }
      std::vector<const Query*> one{q};
      std::vector<TopDocs> r = search_many(one, collector.estimated_hits());
      collector.set_result(std::move(r[0]));
    } catch (const Error& e) {
      if (e.kind != RGPU_ERR_UNSUPPORTED || !cpu_fallback) throw;  // ErrorKind::UnsupportedOperation -> the CPU path
      cpu_fallback(query, collector);
    }
  }

  // the batched form the hardware wants: one launch set per leaf for many queries
  std::vector<TopDocs> search_many(const std::vector<const Query*>& queries, size_t k) {
    const int32_t nq = static_cast<int32_t>(queries.size());
    std::vector<std::vector<rgpu_hit>> leaf_hits(leaves_.size());
    std::vector<std::vector<int64_t>> leaf_totals(leaves_.size());
    for (size_t li = 0; li < leaves_.size(); ++li) {
      std::vector<rgpu_query> qs;
      std::vector<rgpu_query_term> ts;
      plan(queries, li, &qs, &ts);
      leaf_hits[li].assign(static_cast<size_t>(nq) * k, rgpu_hit{-1, 0.f});
      leaf_totals[li].assign(static_cast<size_t>(nq), 0);
      check(rgpu_search_batch(leaves_[li].segment, qs.data(), nq, ts.data(), static_cast<int32_t>(ts.size()), static_cast<int32_t>(k),
                              leaf_hits[li].data(), leaf_totals[li].data()));
    }
    return merge_leaves(leaf_hits, leaf_totals, nq, k);
  }

  // IndexSearcher::search(PhraseQuery, TopDocsCollector(k)) for a batch of exact phrases. PhraseQuery::create_weight
  // (phrase_query.rs:136-186): ONE BM25 weight from the statistics of all the phrase's terms.
  std::vector<TopDocs> search_phrases(const std::vector<const PhraseQuery*>& queries, size_t k) {
    const int32_t nq = static_cast<int32_t>(queries.size());
    std::vector<std::vector<rgpu_hit>> leaf_hits(leaves_.size());
    std::vector<std::vector<int64_t>> leaf_totals(leaves_.size());
    for (size_t li = 0; li < leaves_.size(); ++li) {
      const LeafReader& leaf = leaves_[li];
      if (!leaf.pos_bytes) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "phrase search needs a positions field (LeafReader::pos_bytes)");
      std::vector<rgpu_phrase_query> qs;
      std::vector<rgpu_phrase_term> ts;
      for (const PhraseQuery* q : queries) {
        std::vector<TermStatistics> stats;
        for (const TermQuery& t : q->terms) stats.push_back(term_statistics(t));
        const BM25SimWeight w = sim_.compute_weight(stats_, stats.data(), static_cast<int32_t>(stats.size()), q->boost);
        if (sim_table_ < 0) check(sim_table_ = rgpu_sim_table_upload(ctx_, w.cache.data(), w.k1));
        qs.push_back(rgpu_phrase_query{static_cast<int32_t>(q->terms.size()), static_cast<int32_t>(ts.size()), w.weight, sim_table_, q->slop, next_limit});
        for (size_t i = 0; i < q->terms.size(); ++i) {
          rgpu_phrase_term pt{};
          if (!leaf.positions_state(q->terms[i], &pt.state, &pt.positions)) { pt.state = rgpu_term_state{}; pt.state.skip_offset = -1; pt.state.singleton_doc_id = -1; }
          pt.position = q->positions[i];
          ts.push_back(pt);
        }
      }
      leaf_hits[li].assign(static_cast<size_t>(nq) * k, rgpu_hit{-1, 0.f});
      leaf_totals[li].assign(static_cast<size_t>(nq), 0);
      check(rgpu_search_phrase_batch(leaf.segment, qs.data(), nq, ts.data(), static_cast<int32_t>(ts.size()), static_cast<int32_t>(k),
                                     leaf_hits[li].data(), leaf_totals[li].data()));
    }
    return merge_leaves(leaf_hits, leaf_totals, nq, k);
  }

  // QueryRescorer::rescore (search/scorer/rescorer.rs:118-226) for a batch: row i of `first_pass` is re-ranked by
  // requests[i]. One device call per leaf; the last one sorts the windows and re-weights the tails.
  std::vector<TopDocs> rescore(const std::vector<TopDocs>& first_pass, const std::vector<RescoreRequest>& requests, size_t k) {
    const int32_t nq = static_cast<int32_t>(first_pass.size());
    if (requests.size() != first_pass.size()) throw Error(RGPU_ERR_ILLEGAL_ARGUMENT, "one rescore request per row");
    std::vector<rgpu_hit> rows(static_cast<size_t>(nq) * k, rgpu_hit{-1, 0.f});
    std::vector<rgpu_rescore_request> reqs;
    for (int32_t q = 0; q < nq; ++q) {
      const std::vector<ScoreDoc>& docs = first_pass[static_cast<size_t>(q)].score_docs();
      for (size_t i = 0; i < docs.size() && i < k; ++i) rows[static_cast<size_t>(q) * k + i] = rgpu_hit{docs[i].doc, docs[i].score};
      const RescoreRequest& r = requests[static_cast<size_t>(q)];
      reqs.push_back(rgpu_rescore_request{r.query_weight, r.rescore_weight, static_cast<int32_t>(r.mode),
                                          r.window_size > 0 ? r.window_size : static_cast<int32_t>(k)});
    }
    for (size_t li = 0; li < leaves_.size(); ++li) {
      std::vector<rgpu_query> qs;
      std::vector<rgpu_query_term> ts;
      for (const RescoreRequest& r : requests) pack(*r.query, leaves_[li], &qs, &ts);
      check(rgpu_rescore_batch(leaves_[li].segment, qs.data(), nq, ts.data(), static_cast<int32_t>(ts.size()), reqs.data(),
                               static_cast<int32_t>(k), rows.data(), li + 1 == leaves_.size() ? 1 : 0));
    }
    std::vector<TopDocs> out;
    for (int32_t q = 0; q < nq; ++q) {
      std::vector<ScoreDoc> docs;
      for (size_t i = 0; i < k; ++i) {
        const rgpu_hit& h = rows[static_cast<size_t>(q) * k + i];
        if (h.doc >= 0) docs.push_back(ScoreDoc{h.doc, h.score});
      }
      out.emplace_back(first_pass[static_cast<size_t>(q)].total_hits(), std::move(docs));
    }
    return out;
  }

 private:
  // TopDocsCollector::finish_parallel (top_docs.rs:157-172) over a handful of leaves: canonical order
  std::vector<TopDocs> merge_leaves(const std::vector<std::vector<rgpu_hit>>& leaf_hits, const std::vector<std::vector<int64_t>>& leaf_totals,
                                    int32_t nq, size_t k) const {
    std::vector<TopDocs> out;
    for (int32_t q = 0; q < nq; ++q) {
      std::vector<ScoreDoc> all;
      int64_t total = 0;
      for (size_t li = 0; li < leaves_.size(); ++li) {
        total += leaf_totals[li][static_cast<size_t>(q)];
        for (size_t i = 0; i < k; ++i) {
          const rgpu_hit& h = leaf_hits[li][static_cast<size_t>(q) * k + i];
          if (h.doc >= 0) all.push_back(ScoreDoc{h.doc, h.score});
        }
      }
      std::stable_sort(all.begin(), all.end(), [](const ScoreDoc& a, const ScoreDoc& b) {
        return a.score > b.score || (a.score == b.score && a.doc < b.doc);
      });
      if (all.size() > k) all.resize(k);
      out.emplace_back(total, std::move(all));
    }
    return out;
  }

  std::pair<float, int32_t> weight_of(const TermQuery& tq) {
    // TermQuery::create_weight (term_query.rs:58-95) -> BM25Similarity::compute_weight
    const TermStatistics ts = term_statistics(tq);
    const BM25SimWeight w = sim_.compute_weight(stats_, &ts, 1, tq.boost);
    if (sim_table_ < 0) check(sim_table_ = rgpu_sim_table_upload(ctx_, w.cache.data(), w.k1));  // one field -> one cache
    return {w.weight, sim_table_};
  }
  // The whole batch at once: the query objects are flattened to op / clause-count / term arrays and handed to the native
  // batch planner (rgpu_plan_batch_*: term resolution in this leaf and in the statistics leaf, BM25 weights, the sim table)
  // — one call per leaf instead of one dictionary lookup and one f64 log per clause. A batch that mixes id-named and
  // byte-named terms (or names terms the leaf cannot resolve that way) goes clause by clause through pack().
  void plan(const std::vector<const Query*>& queries, size_t li, std::vector<rgpu_query>* qs, std::vector<rgpu_query_term>* ts) {
    const LeafReader& leaf = leaves_[li];
    std::vector<int32_t> ops, n_terms, n_not;
    std::vector<int64_t> ids, offs{0};
    std::vector<uint8_t> bytes;
    std::vector<float> boosts;
    bool any_id = false, any_text = false, any_boost = false;
    auto clause = [&](const TermQuery& c) {
      if (c.by_text()) { any_text = true; bytes.insert(bytes.end(), c.text.begin(), c.text.end()); }
      else { any_id = true; }
      ids.push_back(c.term);
      offs.push_back(static_cast<int64_t>(bytes.size()));
      boosts.push_back(c.boost);
      any_boost = any_boost || c.boost != 1.0f;
    };
    for (const Query* q : queries) {
      if (auto* t = dynamic_cast<const TermQuery*>(q)) {
        ops.push_back(RGPU_OP_TERM); n_terms.push_back(1); n_not.push_back(0);
        clause(*t);
      } else if (auto* b = dynamic_cast<const BooleanQuery*>(q)) {
        const bool conj = !b->must_queries.empty();
        ops.push_back(conj ? (RGPU_OP_WITH_SHOULD(RGPU_OP_AND, b->should_queries.size()) | (b->should_required ? RGPU_OP_SHOULD_REQUIRED : 0) | (b->nested_must ? RGPU_OP_NESTED_MUST : 0) | RGPU_OP_NESTED_AT(b->nested_at))
                           : (b->min_should_match > 1 ? RGPU_OP_OR_MSM(b->min_should_match) : (int32_t)RGPU_OP_OR));
        n_terms.push_back(static_cast<int32_t>(conj ? b->must_queries.size() : b->should_queries.size()));
        n_not.push_back(static_cast<int32_t>(b->must_not_queries.size()));
        for (const TermQuery& c : (conj ? b->must_queries : b->should_queries)) clause(c);
        if (conj) for (const TermQuery& c : b->should_queries) clause(c);
        for (const TermQuery& c : b->must_not_queries) clause(c);
      } else {
        throw Error(RGPU_ERR_UNSUPPORTED, "query type not served by the GPU path");
      }
    }
    const LeafReader& sl = leaves_[stats_leaf_];
    const bool native = (any_id != any_text) && (any_text ? (leaf.dictionary && sl.dictionary) : (leaf.terms && sl.terms));
    if (!native) {
      for (const Query* q : queries) pack(*q, leaf, qs, ts);
      return;
    }
    if (planners_.size() < 2 * leaves_.size()) planners_.resize(2 * leaves_.size(), nullptr);
    const size_t pi = 2 * li + (any_text ? 1 : 0);  // a leaf may be queried by id and by bytes: one planner each
    if (!planners_[pi]) {
      rgpu_plan_stats ps{stats_.max_doc, stats_.doc_count, stats_.sum_total_term_freq, sim_.k1(), sim_.b()};
      if (sim_table_ < 0) {  // one field -> one norm cache, shared with the clause-by-clause path
        const TermStatistics none;
        check(sim_table_ = rgpu_sim_table_upload(ctx_, sim_.compute_weight(stats_, &none, 1, 1.0f).cache.data(), sim_.k1()));
      }
      if (any_text) check(rgpu_planner_create(nullptr, &ps, leaf.dictionary, &sl == &leaf ? nullptr : sl.dictionary, leaf.field_number, &planners_[pi]));
      else check(rgpu_planner_create_flat(nullptr, &ps, leaf.terms, leaf.n_terms, &sl == &leaf ? nullptr : sl.terms, &sl == &leaf ? 0 : sl.n_terms, &planners_[pi]));
      check(rgpu_planner_set_sim_table(planners_[pi], sim_table_));
    }
    qs->resize(queries.size());
    ts->resize(std::max<size_t>(1, ids.size()));
    const int64_t cap = static_cast<int64_t>(ids.size());
    if (any_text)
      check(rgpu_plan_batch_bytes(planners_[pi], static_cast<int32_t>(queries.size()), ops.data(), n_terms.data(), n_not.data(), bytes.data(), offs.data(),
                                  any_boost ? boosts.data() : nullptr, qs->data(), ts->data(), cap));
    else
      check(rgpu_plan_batch_ids(planners_[pi], static_cast<int32_t>(queries.size()), ops.data(), n_terms.data(), n_not.data(), ids.data(),
                                any_boost ? boosts.data() : nullptr, qs->data(), ts->data(), cap));
    ts->resize(ids.size());
  }
  void pack(const Query& q, const LeafReader& leaf, std::vector<rgpu_query>* qs, std::vector<rgpu_query_term>* ts) {
    const std::vector<TermQuery>* clauses = nullptr;
    const std::vector<TermQuery>* opts = nullptr;  // SHOULD clauses beside MUST ones
    const std::vector<TermQuery>* nots = nullptr;
    std::vector<TermQuery> single;
    int32_t op = RGPU_OP_TERM;
    if (auto* t = dynamic_cast<const TermQuery*>(&q)) {
      single.push_back(*t);
      clauses = &single;
    } else if (auto* b = dynamic_cast<const BooleanQuery*>(&q)) {
      op = b->must_queries.empty() ? (b->min_should_match > 1 ? RGPU_OP_OR_MSM(b->min_should_match) : (int32_t)RGPU_OP_OR)
                                   : (RGPU_OP_WITH_SHOULD(RGPU_OP_AND, b->should_queries.size()) | (b->should_required ? RGPU_OP_SHOULD_REQUIRED : 0) | (b->nested_must ? RGPU_OP_NESTED_MUST : 0) | RGPU_OP_NESTED_AT(b->nested_at));
      clauses = b->must_queries.empty() ? &b->should_queries : &b->must_queries;
      if (!b->must_queries.empty()) opts = &b->should_queries;
      nots = &b->must_not_queries;
    } else {
      throw Error(RGPU_ERR_UNSUPPORTED, "query type not served by the GPU path");
    }
    std::vector<TermQuery> all(*clauses);  // clause order: MUST / scored, optional SHOULD, MUST_NOT
    if (opts) all.insert(all.end(), opts->begin(), opts->end());
    if (nots) all.insert(all.end(), nots->begin(), nots->end());
    rgpu_query rq{op, static_cast<int32_t>(clauses->size()), static_cast<int32_t>(ts->size()),
                  static_cast<int32_t>(nots ? nots->size() : 0)};
    for (const TermQuery& c : all) {
      rgpu_query_term qt{};
      if (!leaf.term_state(c, &qt.state)) { qt.state = rgpu_term_state{}; qt.state.skip_offset = -1; qt.state.singleton_doc_id = -1; }
      auto w = weight_of(c);
      qt.weight = w.first;
      qt.sim_table = w.second;
      ts->push_back(qt);
    }
    qs->push_back(rq);
  }

  std::vector<LeafReader> leaves_;
  BM25Similarity sim_;
  rgpu_ctx* ctx_ = nullptr;
  size_t stats_leaf_ = 0;
  CollectionStatistics stats_;
  int32_t sim_table_ = -1;
  std::vector<rgpu_planner*> planners_;  // per leaf and naming scheme (2 * leaf + by-bytes), created on first use
};

}  // namespace rucene
