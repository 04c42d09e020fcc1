// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's query-evaluation path: SmallFloat norm codec, BM25Similarity, TermScorer,
// ConjunctionScorer, DisjunctionSumScorer (SimpleQueue + DisiPriorityQueue), BulkScorer, TopDocsCollector
// (Rust std BinaryHeap emulation + canonical mode) and DefaultIndexSearcher::search with its
// largest-segment statistics quirk.
//
// PARITY UNPINNED for: TermScorer, DisjunctionSumScorer / DisiPriorityQueue order, top-k tie behaviour of
// std::collections::BinaryHeap (only distinct-score cases are tested by the reference) — see SURVEY.md §8(c).
// Pinned by reference tests (ported in tests/test_oracle_kat.py): small_float.rs:76-115,
// bm25_similarity.rs:400-465, conjunction_scorer.rs:162-222, top_docs.rs:235-264, bulk_scorer.rs:167-200,
// searcher.rs:916-952.
//
// Follows (paths relative to /root/reference/src/core):
//   util/small_float.rs:16-36                    float_to_byte315 / byte315_to_float
//   search/similarity/bm25_similarity.rs:33-43   NORM_TABLE
//   search/similarity/bm25_similarity.rs:72-114  avg_field_length / encode_norm_value / idf
//   search/similarity/bm25_similarity.rs:151-212 compute_weight (cache[256]) / compute_score
//   search/similarity/bm25_similarity.rs:363-366 do_normalize (weight = idf * boost)
//   search/scorer/term_scorer.rs:43-67           TermScorer
//   search/scorer/conjunction_scorer.rs:26-128   ConjunctionScorer
//   search/scorer/disjunction_scorer.rs:24-104, 187-377  DisjunctionSumScorer, SimpleQueue, SubScorers
//   util/disi.rs:135-341                         DisiPriorityQueue
//   search/scorer/bulk_scorer.rs:57-154          BulkScorer::score / score_range_*
//   search/collector/top_docs.rs:28-95           TopDocsBaseCollector
//   search/sort_field/collapse_top_docs.rs:22-68 ScoreDoc ordering (PartialOrd reversed)
//   util/external/binary_heap.rs:121-210         push/sift_up, PeekMut drop -> sift_down, pop -> sift_down_to_bottom
//   search/searcher.rs:306-363, 487-525, 732-767 statistics / search / term_statistics
//   search/query/term_query.rs:57-163, boolean_query.rs:96-125,195-279  weights and scorer construction
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "postings.hpp"

namespace orc {

// ---- SmallFloat -------------------------------------------------------------------------------------------------

inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// small_float.rs:17-26
inline uint8_t float_to_byte315(float f) {
  int32_t bits = (int32_t)f32_bits(f);
  int32_t small_float = bits >> (24 - 3);
  if (small_float <= ((63 - 15) << 3)) return bits <= 0 ? 0 : 1;
  if (small_float >= ((63 - 15) << 3) + 0x100) return 255;
  return (uint8_t)(small_float - ((63 - 15) << 3));
}
// small_float.rs:28-36
inline float byte315_to_float(uint8_t b) {
  if (b == 0) return 0.0f;
  uint32_t bits = (uint32_t)b << (24 - 3);
  bits += (63 - 15) << 24;
  return bits_f32(bits);
}
// small_float.rs:82-107 — the test's own "origin" formulas (used to cross-check, as the reference test does)
inline float origin_byte_to_float(uint8_t b) {
  if (b == 0) return 0.0f;
  uint32_t mantissa = b & 7, exponent = (b >> 3) & 31;
  return bits_f32(((exponent + (63 - 15)) << 24) | (mantissa << 21));
}
inline uint8_t origin_float_to_byte(float f) {
  if (f < 0.0f) return 0;
  int32_t bits = (int32_t)f32_bits(f);
  int32_t mantissa = (bits & 0xffffff) >> 21;
  int32_t exponent = (((bits >> 24) & 0x7f) - 63) + 15;
  if (exponent > 31) { exponent = 31; mantissa = 7; }
  if (exponent < 0 || (exponent == 0 && mantissa == 0)) { exponent = 0; mantissa = 1; }
  return (uint8_t)((exponent << 3) | mantissa);
}

// ---- BM25 --------------------------------------------------------------------------------------------------------

// statistics.rs — CollectionStatistics / TermStatistics (fields BM25 reads)
struct CollectionStatistics {
  int32_t doc_base = 0;
  int64_t max_doc = 0;
  int64_t doc_count = -1;
  int64_t sum_total_term_freq = -1;
  int64_t sum_doc_freq = -1;
};
struct TermStatistics {
  int64_t doc_freq = 0;
  int64_t total_term_freq = -1;
};

// bm25_similarity.rs:33-43
inline const float* norm_table() {
  static float t[256];
  static bool init = false;
  if (!init) {
    for (int i = 1; i < 256; i++) {
      float f = byte315_to_float((uint8_t)i);
      t[i] = 1.0f / (f * f);
    }
    t[0] = 1.0f / t[255];
    init = true;
  }
  return t;
}
// bm25_similarity.rs:72-83
inline float bm25_avg_field_length(const CollectionStatistics& cs) {
  if (cs.sum_total_term_freq <= 0) return 1.0f;
  int64_t doc_count = cs.doc_count == -1 ? cs.max_doc : cs.doc_count;
  return (float)((double)cs.sum_total_term_freq / (double)doc_count);
}
// bm25_similarity.rs:90-92
inline uint8_t bm25_encode_norm_value(float boost, int32_t field_length) {
  return float_to_byte315(boost / std::sqrt((float)field_length));
}
// bm25_similarity.rs:99-114
inline float bm25_idf(const TermStatistics* ts, int n, const CollectionStatistics& cs) {
  float idf = 0.0f;
  int64_t doc_count = cs.doc_count == -1 ? cs.max_doc : cs.doc_count;
  for (int i = 0; i < n; i++) {
    int64_t df = ts[i].doc_freq;
    idf += (float)std::log(1.0 + ((double)doc_count - (double)df + 0.5) / ((double)df + 0.5));
  }
  return idf;
}
// bm25_similarity.rs:151-177 + 240-263 + 363-366
struct BM25Weight {
  float k1, b, idf, boost, weight, avg_dl;
  float cache[256];
};
inline BM25Weight bm25_compute_weight(float k1, float b, const CollectionStatistics& cs, const TermStatistics* ts,
                                      int n, float boost) {
  BM25Weight w;
  w.k1 = k1; w.b = b;
  float avgdl = bm25_avg_field_length(cs);
  w.avg_dl = avgdl;
  w.idf = bm25_idf(ts, n, cs);
  const float* nt = norm_table();
  for (int i = 0; i < 256; i++) w.cache[i] = k1 * ((1.0f - b) + b * (nt[i] / avgdl));
  w.boost = boost;
  w.weight = w.idf * boost;  // do_normalize
  return w;
}
// bm25_similarity.rs:203-212 — f32, left to right; norms absent -> k1
inline float bm25_compute_score(float weight, float k1, float freq, bool has_norms, float norm_cache) {
  float norm = has_norms ? norm_cache : k1;
  return weight * (k1 + 1.0f) * freq / (freq + norm);
}

// ---- Scorer trait + implementations -----------------------------------------------------------------------------

struct Scorer {  // search/mod.rs:66-156 (DocIterator) + scorer/mod.rs:85-99 (Scorer); Box<dyn Scorer>
  virtual ~Scorer() {}
  virtual int32_t doc_id() const = 0;
  virtual int32_t next() = 0;
  virtual int32_t advance(int32_t target) = 0;
  virtual size_t cost() const = 0;
  virtual float score() = 0;
  virtual int32_t approximate_next() { return next(); }
  virtual int32_t approximate_advance(int32_t t) { return advance(t); }
  virtual bool support_two_phase() const { return false; }  // search/mod.rs:141-143
  virtual bool matches() { return true; }                   // search/mod.rs:128-131
  virtual uint64_t postings_visited() const { return 0; }  // instrumentation only
};
typedef std::unique_ptr<Scorer> ScorerBox;

// search/mod.rs:208-300 — MockDocIterator + MockSimpleScorer (score == doc id)
struct MockScorer : Scorer {
  std::vector<int32_t> doc_ids;
  int32_t current = -1;
  int32_t offset = -1;
  explicit MockScorer(std::vector<int32_t> ids) : doc_ids(std::move(ids)) {}
  int32_t doc_id() const override { return current; }
  int32_t next() override {
    offset++;
    current = ((size_t)offset >= doc_ids.size()) ? NO_MORE_DOCS : doc_ids[(size_t)offset];
    return current;
  }
  int32_t advance(int32_t target) override {
    while (true) { int32_t d = next(); if (d >= target) return d; }
  }
  size_t cost() const override { return doc_ids.size(); }
  float score() override { return (float)current; }
};

// term_scorer.rs:43-67 + BM25SimScorer (bm25_similarity.rs:185-212)
struct TermScorer : Scorer {
  BlockDocIterator it;
  float weight, k1;
  const float* cache;   // BM25Weight::cache (Arc<[f32;256]>)
  const uint8_t* norms;  // 1 byte per doc (norms_producer.rs:146-154) or null
  uint64_t visited = 0;
  // index_has_freq: FieldInfo::index_options >= DocsAndFreqs (posting_reader.rs:189-212); a Docs field's freq() is 1
  TermScorer(const PostingsReader* r, const BlockTermState& st, const BM25Weight* w, const uint8_t* norms_, bool index_has_freq = true)
      : it(r, index_has_freq, st, FLAG_FREQS), weight(w->weight), k1(w->k1), cache(w->cache), norms(norms_) {}
  int32_t doc_id() const override { return it.doc_id(); }
  int32_t next() override { visited++; return it.next(); }
  int32_t advance(int32_t t) override { visited++; return it.advance(t); }
  size_t cost() const override { return it.cost(); }
  float score() override {
    int32_t d = it.doc_id();
    float freq = (float)it.freq();
    return bm25_compute_score(weight, k1, freq, norms != nullptr, norms ? cache[norms[d] & 0xFF] : 0.0f);
  }
  uint64_t postings_visited() const override { return visited; }
};

// conjunction_scorer.rs:26-128
struct ConjunctionScorer : Scorer {
  ScorerBox lead1, lead2;
  std::vector<ScorerBox> others;
  explicit ConjunctionScorer(std::vector<ScorerBox> children) {
    if (children.size() < 2) throw OracleError(E_ILLEGAL_ARGUMENT, "conjunction needs >= 2 children");
    std::stable_sort(children.begin(), children.end(),
                     [](const ScorerBox& a, const ScorerBox& b) { return a->cost() < b->cost(); });
    lead1 = std::move(children[0]);
    lead2 = std::move(children[1]);
    for (size_t i = 2; i < children.size(); i++) others.push_back(std::move(children[i]));
  }
  int32_t skip_to_approx(int32_t target) {
    int32_t doc = target;
    while (true) {  // 'advanceHead
      int32_t next2 = lead2->approximate_advance(doc);
      if (next2 != doc) {
        doc = lead1->approximate_advance(next2);
        if (next2 != doc) continue;
      }
      if (doc == NO_MORE_DOCS) return doc;
      bool restart = false;
      for (auto& other : others) {
        if (other->doc_id() < doc) {
          int32_t next = other->approximate_advance(doc);
          if (next > doc) {
            doc = lead1->approximate_advance(next);
            restart = true;
            break;
          }
        }
      }
      if (restart) continue;
      return doc;
    }
  }
  float score() override {
    float s = lead1->score();
    s += lead2->score();
    for (auto& o : others) s += o->score();
    return s;
  }
  int32_t doc_id() const override { return lead1->doc_id(); }
  int32_t next() override { return approximate_next(); }
  int32_t advance(int32_t t) override { return approximate_advance(t); }
  size_t cost() const override { return lead1->cost(); }
  int32_t approximate_next() override { return skip_to_approx(lead1->approximate_next()); }
  int32_t approximate_advance(int32_t t) override { return skip_to_approx(lead1->approximate_advance(t)); }
  uint64_t postings_visited() const override {
    uint64_t v = lead1->postings_visited() + lead2->postings_visited();
    for (auto& o : others) v += o->postings_visited();
    return v;
  }
};

// util/disi.rs:135-341 — hand-rolled min-heap on doc id with top_list()
struct DisiPriorityQueue {
  struct Wrapper { Scorer* scorer; Wrapper* next; int32_t doc() const { return scorer->doc_id(); } };
  std::vector<Wrapper*> heap;
  size_t size = 0;
  std::vector<Wrapper> buffer;
  static size_t left_node(size_t n) { return ((n + 1) << 1) - 1; }
  static size_t right_node(size_t l) { return l + 1; }
  explicit DisiPriorityQueue(std::vector<ScorerBox>& children) {
    buffer.reserve(children.size());
    for (auto& c : children) buffer.push_back(Wrapper{c.get(), nullptr});
    heap.assign(children.size(), nullptr);
    for (auto& w : buffer) {  // do_push
      heap[size] = &w;
      up_heap(size);
      size++;
    }
  }
  Wrapper* peek() const { return heap[0]; }
  void up_heap(size_t i) {
    Wrapper* node = heap[i];
    int32_t node_doc = node->doc();
    while (i > 0) {
      size_t j = ((i + 1) >> 1) - 1;
      if (node_doc >= heap[j]->doc()) break;
      heap[i] = heap[j];
      i = j;
    }
    heap[i] = node;
  }
  void down_heap(size_t sz) {
    size_t i = 0;
    Wrapper* node = heap[0];
    size_t j = left_node(i);
    if (j < size) {
      size_t k = right_node(j);
      if (k < sz && heap[k]->doc() < heap[j]->doc()) j = k;
      if (heap[j]->doc() < node->doc()) {
        do {
          heap[i] = heap[j];
          i = j;
          j = left_node(i);
          k = right_node(j);
          if (k < sz && heap[k]->doc() < heap[j]->doc()) j = k;
        } while (j < sz && heap[j]->doc() < node->doc());
        heap[i] = node;
      }
    }
  }
  void update_top() { down_heap(size); }
  Wrapper* top_list() {
    Wrapper* list = heap[0];
    list->next = nullptr;
    if (size >= 3) {
      list = top_list_to(list, 1);
      list = top_list_to(list, 2);
    } else if (size == 2 && heap[1]->doc() == list->doc()) {
      heap[1]->next = list;
      list = heap[1];
    }
    return list;
  }
  Wrapper* top_list_to(Wrapper* list, size_t i) {
    Wrapper* w = heap[i];
    if (w->doc() == list->doc()) {
      w->next = list;
      list = w;
      size_t left = left_node(i), right = left + 1;
      if (right < size) {
        list = top_list_to(list, left);
        list = top_list_to(list, right);
      } else if (left < size && heap[left]->doc() == list->doc()) {
        heap[left]->next = list;
        list = heap[left];
      }
    }
    return list;
  }
};

// disjunction_scorer.rs:24-104, 187-377 (min_should_match > 1 forces the SimpleQueue, :41)
struct DisjunctionSumScorer : Scorer {
  std::vector<ScorerBox> children;
  std::unique_ptr<DisiPriorityQueue> dpq;  // null -> SimpleQueue
  int32_t curr_doc = NO_MORE_DOCS;         // SimpleQueue::curr_doc
  bool needs_scores;
  size_t cost_;
  int32_t min_should_match;
  DisjunctionSumScorer(std::vector<ScorerBox> ch, bool needs_scores_, int32_t msm)
      : children(std::move(ch)), needs_scores(needs_scores_), min_should_match(msm) {
    cost_ = 0;
    for (auto& c : children) cost_ += c->cost();
    if (children.size() < 10 || min_should_match > 1) {
      for (auto& s : children) curr_doc = std::min(curr_doc, s->doc_id());  // SimpleQueue::new
    } else {
      dpq.reset(new DisiPriorityQueue(children));
    }
  }
  float score() override {
    if (!needs_scores) return 0.0f;
    float s = 0.0f;
    if (!dpq) {
      for (auto& c : children) if (c->doc_id() == curr_doc) s += c->score();
    } else {
      for (auto* d = dpq->top_list(); d != nullptr; d = d->next) s += d->scorer->score();
    }
    return s;
  }
  int32_t doc_id() const override { return dpq ? dpq->peek()->doc() : curr_doc; }
  int32_t next() override { return approximate_next(); }
  int32_t advance(int32_t t) override { return approximate_advance(t); }
  size_t cost() const override { return cost_; }
  int32_t approximate_next() override {
    if (!dpq) {
      while (true) {
        if (curr_doc == NO_MORE_DOCS) return curr_doc;
        int32_t cur = curr_doc, min_doc = NO_MORE_DOCS;
        for (auto& s : children) {
          if (s->doc_id() == cur) s->approximate_next();
          min_doc = std::min(min_doc, s->doc_id());
        }
        curr_doc = min_doc;
        if (min_should_match > 1) {
          int should_count = 0;
          for (auto& s : children) if (s->doc_id() == min_doc) should_count++;
          if (should_count < min_should_match) continue;
        }
        return curr_doc;
      }
    }
    int32_t doc = dpq->peek()->doc();
    do {
      dpq->peek()->scorer->approximate_next();
      dpq->update_top();  // PeekMut drop
    } while (dpq->peek()->doc() == doc);
    return dpq->peek()->doc();
  }
  int32_t approximate_advance(int32_t target) override {
    if (!dpq) {
      int32_t min_doc = NO_MORE_DOCS;
      for (auto& s : children) {
        if (s->doc_id() < target) s->approximate_advance(target);
        min_doc = std::min(min_doc, s->doc_id());
      }
      curr_doc = min_doc;
      return curr_doc;
    }
    do {
      dpq->peek()->scorer->approximate_advance(target);
      dpq->update_top();
    } while (dpq->peek()->doc() < target);
    return dpq->peek()->doc();
  }
  uint64_t postings_visited() const override {
    uint64_t v = 0;
    for (auto& c : children) v += c->postings_visited();
    return v;
  }
};

// scorer/req_not_scorer.rs:20-120 — required scorer minus the docs of a prohibited one; scores are the required
// scorer's. PINNED by the reference's own tests (req_not_scorer.rs:126-165, tests/test_oracle_kat.py).
struct ReqNotScorer : Scorer {
  ScorerBox req_scorer, not_scorer;
  ReqNotScorer(ScorerBox req, ScorerBox nots) : req_scorer(std::move(req)), not_scorer(std::move(nots)) {}
  float score() override { return req_scorer->score(); }  // :36-40
  int32_t doc_id() const override { return req_scorer->doc_id(); }
  int32_t next() override {  // :47-63
    while (true) {
      const int32_t doc = req_scorer->next();
      if (doc == NO_MORE_DOCS) break;
      if (doc == not_scorer->doc_id()) continue;
      else if (doc < not_scorer->doc_id()) return doc;
      const int32_t not_doc = not_scorer->advance(doc);
      if (doc < not_doc) return doc;
    }
    return NO_MORE_DOCS;
  }
  int32_t advance(int32_t target) override {  // :65-78
    const int32_t doc = req_scorer->advance(target);
    if (doc < NO_MORE_DOCS) {
      while (true) {
        if (doc == not_scorer->doc_id()) return next();
        else if (doc < not_scorer->doc_id()) return doc;
        not_scorer->advance(doc);
      }
    }
    return NO_MORE_DOCS;
  }
  size_t cost() const override { return req_scorer->cost(); }
  int32_t approximate_next() override {  // :88-104
    while (true) {
      const int32_t doc = req_scorer->approximate_next();
      if (doc == NO_MORE_DOCS) break;
      if (doc == not_scorer->doc_id()) continue;
      else if (doc < not_scorer->doc_id()) return doc;
      const int32_t not_doc = not_scorer->approximate_advance(doc);
      if (doc < not_doc) return doc;
    }
    return NO_MORE_DOCS;
  }
  int32_t approximate_advance(int32_t target) override {  // :106-119
    const int32_t doc = req_scorer->approximate_advance(target);
    if (doc < NO_MORE_DOCS) {
      while (true) {
        if (doc == not_scorer->doc_id()) return approximate_next();
        else if (doc < not_scorer->doc_id()) return doc;
        not_scorer->approximate_advance(doc);
      }
    }
    return NO_MORE_DOCS;
  }
  uint64_t postings_visited() const override { return req_scorer->postings_visited() + not_scorer->postings_visited(); }
};

// scorer/req_opt_scorer.rs:19-100 — a required scorer whose score is topped up by an optional one positioned on the
// same doc. score() carries SEQUENTIAL state: once more than OPT_SCORE_THRESHOLD docs took the optional path, a doc
// whose required score is under half the running mean of those docs' required scores skips the optional clause
// (:46-50). Iteration is the required scorer's. PINNED by the reference's own test (req_opt_scorer.rs:104-134: scores
// 6, 9, 20 — tests/test_oracle_kat.py); the skipping rule itself is exercised by no reference test (parity unpinned).
constexpr size_t OPT_SCORE_THRESHOLD = 100;
struct ReqOptScorer : Scorer {
  ScorerBox req_scorer, opt_scorer;
  float scores_sum = 0.0f;
  size_t scores_num = 0;
  // threshold: OPT_SCORE_THRESHOLD is the reference; SIZE_MAX turns the skipping rule off ("exact sums": what the GPU
  // path computes, see include/rucene_gpu.h RGPU_OP_WITH_SHOULD) so that tests can pin that path bit for bit
  size_t threshold;
  ReqOptScorer(ScorerBox req, ScorerBox opt, size_t threshold_ = OPT_SCORE_THRESHOLD)
      : req_scorer(std::move(req)), opt_scorer(std::move(opt)), threshold(threshold_) {}
  float score() override {  // :41-66
    const int32_t current_doc = req_scorer->doc_id();
    float score = req_scorer->score();
    if (scores_num > threshold) {
      if (2.0f * score < scores_sum / (float)scores_num) return score;
    }
    scores_sum += score;
    scores_num += 1;
    int32_t opt_doc = opt_scorer->doc_id();
    if (opt_doc < current_doc) opt_doc = opt_scorer->advance(current_doc);
    if (opt_doc == current_doc) score += opt_scorer->score();
    return score;
  }
  int32_t doc_id() const override { return req_scorer->doc_id(); }
  int32_t next() override { return req_scorer->next(); }
  int32_t advance(int32_t target) override { return req_scorer->advance(target); }
  size_t cost() const override { return req_scorer->cost(); }
  int32_t approximate_next() override { return req_scorer->approximate_next(); }
  int32_t approximate_advance(int32_t target) override { return req_scorer->approximate_advance(target); }
  uint64_t postings_visited() const override { return req_scorer->postings_visited() + opt_scorer->postings_visited(); }
};

// ---- TopDocsCollector --------------------------------------------------------------------------------------------

struct ScoreDoc { int32_t doc; float score; };

enum TieMode { TIE_RUST_HEAP = 0, TIE_CANONICAL = 1 };

// top_docs.rs:28-95. rust_heap mode emulates std BinaryHeap<ScoreDoc> (binary_heap.rs:121-210) where the
// heap's `<=`/`>=` are ScoreDoc's PartialOrd, i.e. the REVERSE of score order (collapse_top_docs.rs:54-60):
// a <= b  <=>  a.score >= b.score. canonical mode keeps the k best under (score desc, doc asc).
struct TopDocsCollector {
  std::vector<ScoreDoc> pq;
  size_t estimated_hits;
  size_t total_hits = 0;
  int32_t cur_doc_base = 0;
  int mode;
  TopDocsCollector(size_t k, int mode_) : estimated_hits(k), mode(mode_) { pq.reserve(k); }

  static bool le(const ScoreDoc& a, const ScoreDoc& b) { return a.score >= b.score; }  // PartialOrd <=
  static bool ge(const ScoreDoc& a, const ScoreDoc& b) { return a.score <= b.score; }  // PartialOrd >=
  size_t sift_up(size_t start, size_t pos) {
    ScoreDoc elt = pq[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (le(elt, pq[parent])) break;
      pq[pos] = pq[parent];
      pos = parent;
    }
    pq[pos] = elt;
    return pos;
  }
  void sift_down_range(size_t pos, size_t end) {
    ScoreDoc elt = pq[pos];
    size_t child = 2 * pos + 1;
    while (child < end) {
      size_t right = child + 1;
      if (right < end && le(pq[child], pq[right])) child = right;
      if (ge(elt, pq[child])) break;
      pq[pos] = pq[child];
      pos = child;
      child = 2 * pos + 1;
    }
    pq[pos] = elt;
  }
  void sift_down_to_bottom(size_t pos) {
    size_t end = pq.size(), start = pos;
    ScoreDoc elt = pq[pos];
    size_t child = 2 * pos + 1;
    while (child < end) {
      size_t right = child + 1;
      if (right < end && le(pq[child], pq[right])) child = right;
      pq[pos] = pq[child];
      pos = child;
      child = 2 * pos + 1;
    }
    pq[pos] = elt;
    sift_up(start, pos);
  }
  void heap_push(ScoreDoc d) { pq.push_back(d); sift_up(0, pq.size() - 1); }
  ScoreDoc heap_pop() {
    ScoreDoc item = pq.back();
    pq.pop_back();
    if (!pq.empty()) { std::swap(item, pq[0]); sift_down_to_bottom(0); }
    return item;
  }
  // canonical: "better" = higher score, then lower doc. pq kept as a min-heap on that order via std heap.
  static bool canon_better(const ScoreDoc& a, const ScoreDoc& b) {
    return a.score > b.score || (a.score == b.score && a.doc < b.doc);
  }

  // top_docs.rs:67-76
  void add_doc(int32_t doc_id, float score) {
    if (mode == TIE_RUST_HEAP) {
      if (pq.size() < estimated_hits) heap_push(ScoreDoc{doc_id, score});
      else if (!pq.empty()) {
        if (pq[0].score < score) { pq[0] = ScoreDoc{doc_id, score}; sift_down_range(0, pq.size()); }
      }
    } else {
      ScoreDoc d{doc_id, score};
      if (pq.size() < estimated_hits) {
        pq.push_back(d);
        std::push_heap(pq.begin(), pq.end(), canon_better);  // top = worst
      } else if (!pq.empty() && canon_better(d, pq[0])) {
        std::pop_heap(pq.begin(), pq.end(), canon_better);
        pq.back() = d;
        std::push_heap(pq.begin(), pq.end(), canon_better);
      }
    }
  }
  // top_docs.rs:84-94
  void collect(int32_t doc, Scorer* scorer) {
    float score = scorer->score();
    add_doc(doc + cur_doc_base, score);
    total_hits++;
  }
  // top_docs.rs:43-55 — pop all, reverse
  std::vector<ScoreDoc> top_docs() {
    size_t size = std::min(total_hits, pq.size());
    std::vector<ScoreDoc> out;
    if (mode == TIE_RUST_HEAP) {
      for (size_t i = 0; i < size; i++) out.push_back(heap_pop());
      std::reverse(out.begin(), out.end());
    } else {
      out = pq;
      std::sort(out.begin(), out.end(), canon_better);
      out.resize(size);
    }
    return out;
  }
};

// bulk_scorer.rs:57-154; live_docs = FixedBitSet i64 words (bit_set.rs:453-460) or null
// (MatchAllBits). `max_collect_per_leaf` > 0 emulates EarlyTerminatingSortingCollector raising
// LeafCollectionTerminated after N docs (collector/early_terminating.rs) — used only by the searcher KAT.
// A scorer with support_two_phase() (of the scorers restated here: SloppyPhraseScorer, phrase_scorer.rs:1060-1062) takes
// the two-phase arms (:97-113, :128-146): the live-docs test comes BEFORE matches(), every approximation — matching or not,
// live or not — counts towards `next`, and a leaf on which `next_limit` + 1 approximations went by without a single collected
// doc is abandoned (searcher.rs:47 DEFAULT_DISMATCH_NEXT_LIMIT = 500 000; DefaultIndexSearcher::new(reader, next_limit)).
constexpr size_t DEFAULT_DISMATCH_NEXT_LIMIT = 500000;
inline int32_t bulk_score(Scorer* scorer, TopDocsCollector* collector, const uint64_t* live_docs, int32_t min,
                          int32_t max, int max_collect_per_leaf = 0, size_t next_limit = DEFAULT_DISMATCH_NEXT_LIMIT) {
  int32_t current_doc = (min == 0 && max == NO_MORE_DOCS) ? scorer->approximate_next() : scorer->approximate_advance(min);
  int collected = 0;
  if (scorer->support_two_phase()) {
    size_t next = 0, collect = 0;
    while (current_doc < max) {
      const bool live = live_docs == nullptr || ((live_docs[current_doc >> 6] >> (current_doc & 63)) & 1);
      if (live && scorer->matches()) {
        collector->collect(current_doc, scorer);
        collect += 1;
      }
      current_doc = scorer->approximate_next();
      next += 1;
      if (collect == 0 && next > next_limit) break;
    }
    return current_doc;
  }
  while (current_doc < max) {
    bool live = live_docs == nullptr || ((live_docs[current_doc >> 6] >> (current_doc & 63)) & 1);
    if (live) {
      collector->collect(current_doc, scorer);
      if (max_collect_per_leaf > 0 && ++collected >= max_collect_per_leaf) return current_doc;  // leaf terminated
    }
    current_doc = scorer->next();
  }
  return current_doc;
}

// ---- index + searcher --------------------------------------------------------------------------------------------

// One segment (leaf): opened .doc, 1-byte norms, optional live docs, a flat term table standing in for the
// block-tree dictionary (out of scope: SURVEY.md §2 row 11), and the FieldReader statistics.
struct Segment {
  std::unique_ptr<PostingsReader> reader;
  const uint8_t* norms = nullptr;
  const uint64_t* live_docs = nullptr;
  int32_t max_doc = 0;
  int32_t doc_base = 0;
  int64_t doc_count = 0;             // Terms::doc_count
  int64_t sum_total_term_freq = 0;   // Terms::sum_total_term_freq
  int64_t sum_doc_freq = 0;
  const BlockTermState* terms = nullptr;  // indexed by term id; doc_freq == 0 -> term absent in this segment
  int64_t n_terms = 0;
  bool index_has_freq = true;        // the field's IndexOptions >= DocsAndFreqs
};

enum QueryOp { OP_TERM = 0, OP_AND = 1, OP_OR = 2 };
struct Query {
  int op;
  std::vector<int64_t> term_ids;
  std::vector<float> boosts;
  int32_t min_should_match = 0;
  std::vector<int64_t> must_not_ids;  // MUST_NOT TermQuery clauses (boolean_query.rs:33), scored with needs_scores = false
  std::vector<int64_t> opt_ids;       // SHOULD TermQuery clauses next to MUST ones (op == OP_AND / OP_TERM): ReqOptScorer
  bool opt_exact = false;             // true: never skip the optional clauses (not the reference's behaviour)
};

struct SearchResult {
  std::vector<ScoreDoc> hits;
  int64_t total_hits = 0;
  uint64_t postings_visited = 0;
};

struct IndexSearcher {
  std::vector<Segment*> leaves;  // reader order (doc_base ascending)
  float k1 = 1.2f, b = 0.75f;
  CollectionStatistics field_stats;  // the single field "body"
  int stats_leaf = 0;

  // searcher.rs:306-363 — statistics come from the first leaf with the largest max_doc (stable sort desc)
  explicit IndexSearcher(std::vector<Segment*> l) : leaves(std::move(l)) {
    int64_t total_max_doc = 0;
    for (auto* s : leaves) total_max_doc += s->max_doc;
    stats_leaf = 0;
    for (size_t i = 1; i < leaves.size(); i++)
      if (leaves[i]->max_doc > leaves[(size_t)stats_leaf]->max_doc) stats_leaf = (int)i;
    const Segment* s = leaves[(size_t)stats_leaf];
    field_stats.doc_base = s->doc_base;
    field_stats.max_doc = total_max_doc;
    field_stats.doc_count = s->doc_count;
    field_stats.sum_total_term_freq = s->sum_total_term_freq;
    field_stats.sum_doc_freq = s->sum_doc_freq;
  }
  // Sharded deployment: every shard scores with the statistics of the index-wide largest
  // leaf, which may live on another rank; the host ships that leaf's statistics instead of recomputing idf.
  const Segment* stats_override = nullptr;
  void override_statistics(const Segment* s, int64_t total_max_doc) {
    stats_override = s;
    field_stats.doc_base = s->doc_base;
    field_stats.max_doc = total_max_doc;
    field_stats.doc_count = s->doc_count;
    field_stats.sum_total_term_freq = s->sum_total_term_freq;
    field_stats.sum_doc_freq = s->sum_doc_freq;
  }
  // searcher.rs:732-767 — df/ttf of the term in the statistics leaf only
  TermStatistics term_statistics(int64_t term_id) const {
    TermStatistics ts;
    const Segment* s = stats_override ? stats_override : leaves[(size_t)stats_leaf];
    ts.doc_freq = 0;
    ts.total_term_freq = 0;
    if (term_id >= 0 && term_id < s->n_terms && s->terms[term_id].doc_freq > 0) {
      ts.doc_freq = s->terms[term_id].doc_freq;
      ts.total_term_freq = s->terms[term_id].total_term_freq;
    }
    return ts;
  }
  // term_query.rs:58-95 -> BM25Similarity::compute_weight
  BM25Weight term_weight(int64_t term_id, float boost) const {
    TermStatistics ts = term_statistics(term_id);
    return bm25_compute_weight(k1, b, field_stats, &ts, 1, boost);
  }
  // term_query.rs:145-163 — None when the term is absent from the leaf
  ScorerBox term_scorer(const Segment* seg, int64_t term_id, const BM25Weight* w) const {
    if (term_id < 0 || term_id >= seg->n_terms || seg->terms[term_id].doc_freq <= 0) return nullptr;
    return ScorerBox(new TermScorer(seg->reader.get(), seg->terms[term_id], w, seg->norms, seg->index_has_freq));
  }
  // boolean_query.rs:195-279 restricted to trees of TermQuery clauses: MUST only, SHOULD only, each optionally
  // with MUST_NOT clauses; MUST + SHOULD -> ReqOptScorer is added by create_scorer below
  ScorerBox positive_scorer(const Segment* seg, const Query& q, const std::vector<BM25Weight>& weights) const {
    if (q.op == OP_TERM) return term_scorer(seg, q.term_ids[0], &weights[0]);
    std::vector<ScorerBox> scorers;
    for (size_t i = 0; i < q.term_ids.size(); i++) {
      ScorerBox s = term_scorer(seg, q.term_ids[i], &weights[i]);
      if (s) scorers.push_back(std::move(s));
      else if (q.op == OP_AND) return nullptr;
    }
    if (q.op == OP_AND) {
      // BooleanQuery::build collapses a single-clause query to the clause itself (boolean_query.rs:66-75);
      // with MUST_NOT clauses the single MUST scorer is used as is (boolean_query.rs:209-213)
      if (scorers.size() == 1) return std::move(scorers[0]);
      return ScorerBox(new ConjunctionScorer(std::move(scorers)));
    }
    if (scorers.empty()) return nullptr;
    int32_t msm = q.min_should_match > 0 ? q.min_should_match : 1;  // boolean_query.rs:47-55
    return ScorerBox(new DisjunctionSumScorer(std::move(scorers), true, msm));
  }
  ScorerBox create_scorer(const Segment* seg, const Query& q, const std::vector<BM25Weight>& weights) const {
    ScorerBox positive = positive_scorer(seg, q, weights);
    if (positive && !q.opt_ids.empty()) {
      // boolean_query.rs:217-233, 253-262: the SHOULD clauses that exist in this leaf always go through a
      // DisjunctionSumScorer (even a single one), with the query's min_should_match (0 next to MUST clauses unless set)
      std::vector<ScorerBox> opts;
      const size_t base = q.term_ids.size() + q.must_not_ids.size();
      for (size_t i = 0; i < q.opt_ids.size(); i++) {
        ScorerBox s = term_scorer(seg, q.opt_ids[i], &weights[base + i]);
        if (s) opts.push_back(std::move(s));
      }
      if (!opts.empty())
        positive.reset(new ReqOptScorer(std::move(positive), ScorerBox(new DisjunctionSumScorer(std::move(opts), true, q.min_should_match)),
                                        q.opt_exact ? SIZE_MAX : OPT_SCORE_THRESHOLD));
    }
    if (!positive || q.must_not_ids.empty()) return positive;
    // boolean_query.rs:235-252: absent terms drop out; one scorer is used directly, several are united by a
    // DisjunctionSumScorer(needs_scores = false, the query's min_should_match: 0 with MUST clauses, else 1)
    std::vector<ScorerBox> nots;
    for (size_t i = 0; i < q.must_not_ids.size(); i++) {
      ScorerBox s = term_scorer(seg, q.must_not_ids[i], &weights[q.term_ids.size() + i]);
      if (s) nots.push_back(std::move(s));
    }
    if (nots.empty()) return positive;
    ScorerBox prohibited;
    if (nots.size() == 1) {
      prohibited = std::move(nots[0]);
    } else {
      const int32_t msm = q.min_should_match > 0 ? q.min_should_match : (q.op == OP_OR ? 1 : 0);
      prohibited.reset(new DisjunctionSumScorer(std::move(nots), false, msm));
    }
    return ScorerBox(new ReqNotScorer(std::move(positive), std::move(prohibited)));  // boolean_query.rs:264-273
  }
  // searcher.rs:487-525
  SearchResult search(const Query& q, size_t k, int tie_mode, int max_collect_per_leaf = 0) const {
    std::vector<BM25Weight> weights;
    for (size_t i = 0; i < q.term_ids.size(); i++)
      weights.push_back(term_weight(q.term_ids[i], q.boosts.empty() ? 1.0f : q.boosts[i]));
    for (size_t i = 0; i < q.must_not_ids.size(); i++) weights.push_back(term_weight(q.must_not_ids[i], 1.0f));
    for (size_t i = 0; i < q.opt_ids.size(); i++) weights.push_back(term_weight(q.opt_ids[i], 1.0f));
    TopDocsCollector collector(k, tie_mode);
    SearchResult r;
    for (auto* seg : leaves) {
      ScorerBox scorer = create_scorer(seg, q, weights);
      if (!scorer) continue;
      collector.cur_doc_base = seg->doc_base;
      bulk_score(scorer.get(), &collector, seg->live_docs, 0, NO_MORE_DOCS, max_collect_per_leaf);
      r.postings_visited += scorer->postings_visited();
    }
    r.total_hits = (int64_t)collector.total_hits;
    r.hits = collector.top_docs();
    return r;
  }

  // The query's own score of GIVEN docs (global ids, ascending): per leaf one scorer, advanced from doc to doc — the walk of
  // QueryRescorer::iterative_rescore (rescorer.rs:231-277) without the combining step. What a parity check needs where the
  // reference pins a score only up to summation order (DisjunctionSumScorer over a DisiPriorityQueue,
  // disjunction_scorer.rs:41-45): "is this doc a match, and what does the reference score it" for docs another
  // implementation returned. matched[i] = 0 (score 0) for a doc the query does not match or a deleted one.
  void score_docs(const Query& q, const int32_t* docs, size_t n_docs, float* scores_out, uint8_t* matched_out) const {
    std::vector<BM25Weight> weights;
    for (size_t i = 0; i < q.term_ids.size(); i++) weights.push_back(term_weight(q.term_ids[i], q.boosts.empty() ? 1.0f : q.boosts[i]));
    for (size_t i = 0; i < q.must_not_ids.size(); i++) weights.push_back(term_weight(q.must_not_ids[i], 1.0f));
    for (size_t i = 0; i < q.opt_ids.size(); i++) weights.push_back(term_weight(q.opt_ids[i], 1.0f));
    size_t at = 0;
    for (auto* seg : leaves) {
      const int32_t end_doc = seg->doc_base + seg->max_doc;
      if (at >= n_docs) break;
      if (docs[at] >= end_doc) continue;
      ScorerBox scorer = create_scorer(seg, q, weights);
      for (; at < n_docs && docs[at] < end_doc; ++at) {
        if (at > 0 && docs[at] <= docs[at - 1]) throw OracleError(E_ILLEGAL_ARGUMENT, "score_docs: docs must be strictly ascending");
        scores_out[at] = 0.0f;
        matched_out[at] = 0;
        if (!scorer || docs[at] < seg->doc_base) continue;
        const int32_t target = docs[at] - seg->doc_base;
        if (seg->live_docs && !((seg->live_docs[target >> 6] >> (target & 63)) & 1)) continue;
        int32_t actual = scorer->doc_id();
        if (actual < target) actual = scorer->advance(target);
        if (actual == target) { scores_out[at] = scorer->score(); matched_out[at] = 1; }
      }
    }
    for (; at < n_docs; ++at) { scores_out[at] = 0.0f; matched_out[at] = 0; }
  }

  // ---- QueryRescorer (search/scorer/rescorer.rs:129-374) ---------------------------------------------------------------
  // RescoreMode::combine (:97-116): 0 Avg, 1 Max, 2 Min, 3 Total, 4 Multiply
  static float rescore_mode_combine(int mode, float primary, float secondary) {
    switch (mode) {
      case 0: return (primary + secondary) / 2.0f;
      case 1: return primary >= secondary ? primary : secondary;  // f32::max (no NaNs here)
      case 2: return primary <= secondary ? primary : secondary;
      case 3: return primary + secondary;
      default: return primary * secondary;
    }
  }
  // combine_score (:337-352)
  static float rescore_combine_score(int mode, float query_weight, float rescore_weight, float last_score, bool is_match, float new_score) {
    if (is_match) return rescore_mode_combine(mode, last_score * query_weight, new_score * rescore_weight);
    return last_score * query_weight;
  }
  // rescore (:376-390) = query_rescore (:279-335: the window's hits sorted by doc, iterative_rescore :231-277 — one scorer
  // per leaf advanced from hit to hit — then hits.sort(): score desc, doc asc) + combine_docs (:354-374: the rescored
  // window goes back on top, hits past it take the query weight). `hits` is the first pass's list, best first.
  void rescore(const Query& q, std::vector<ScoreDoc>& hits, size_t window_size, float query_weight, float rescore_weight, int mode) const {
    std::vector<BM25Weight> weights;
    for (size_t i = 0; i < q.term_ids.size(); i++) weights.push_back(term_weight(q.term_ids[i], q.boosts.empty() ? 1.0f : q.boosts[i]));
    for (size_t i = 0; i < q.must_not_ids.size(); i++) weights.push_back(term_weight(q.must_not_ids[i], 1.0f));
    for (size_t i = 0; i < q.opt_ids.size(); i++) weights.push_back(term_weight(q.opt_ids[i], 1.0f));
    std::vector<ScoreDoc> window(hits.begin(), hits.begin() + (ptrdiff_t)std::min(window_size, hits.size()));
    std::sort(window.begin(), window.end(), [](const ScoreDoc& a, const ScoreDoc& b) { return a.doc < b.doc; });
    size_t hit_upto = 0;
    int32_t end_doc = 0, doc_base = 0;
    int reader_idx = -1, current_reader_idx = -1;
    ScorerBox scorer;
    while (hit_upto < window.size()) {
      const int32_t doc_id = window[hit_upto].doc;
      const float current_score = window[hit_upto].score;
      while (doc_id >= end_doc && reader_idx < (int)leaves.size() - 1) {
        reader_idx += 1;
        end_doc = leaves[(size_t)reader_idx]->doc_base + leaves[(size_t)reader_idx]->max_doc;
      }
      if (reader_idx != current_reader_idx) {
        doc_base = leaves[(size_t)reader_idx]->doc_base;
        scorer = create_scorer(leaves[(size_t)reader_idx], q, weights);
        current_reader_idx = reader_idx;
      }
      if (scorer) {
        const int32_t target_doc = doc_id - doc_base;
        int32_t actual_doc = scorer->doc_id();
        if (actual_doc < target_doc) actual_doc = scorer->advance(target_doc);
        if (actual_doc == target_doc) window[hit_upto].score = rescore_combine_score(mode, query_weight, rescore_weight, current_score, true, scorer->score());
        else window[hit_upto].score = rescore_combine_score(mode, query_weight, rescore_weight, current_score, false, 0.0f);
      } else {
        window[hit_upto].score = rescore_combine_score(mode, query_weight, rescore_weight, current_score, false, 0.0f);
      }
      hit_upto += 1;
    }
    std::sort(window.begin(), window.end(), [](const ScoreDoc& a, const ScoreDoc& b) { return a.score != b.score ? a.score > b.score : a.doc < b.doc; });
    for (size_t i = 0; i < window.size(); i++) hits[i] = window[i];
    for (size_t i = window.size(); i < hits.size(); i++) hits[i].score = hits[i].score * query_weight;
  }
};

}  // namespace orc
