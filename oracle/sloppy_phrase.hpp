// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's SloppyPhraseScorer (PhraseQuery with slop > 0), single-term phrase positions only (what
// PhraseQuery builds; the multi-term branches belong to MultiPhraseQuery and are not restated).
//
// PINNED by the reference's own phrase test (query/phrase_query.rs:511-637: "quick fox" with slop 0 / 1 / 2 / 3 over three
// documents -> total_hits 1 / 2 / 2 / 3; tests/test_sloppy_phrase.py). Everything finer — sloppy frequencies, the repeats
// machinery — is PARITY UNPINNED: the source text is the only authority.
//
// Follows (paths relative to /root/reference/src/core/search):
//   query/phrase_query.rs:272-330        PhraseWeight::create_scorer: slop != 0 -> SloppyPhraseScorer over the postings in QUERY order
//   scorer/phrase_scorer.rs:318-376      PhrasePositions::{new, first_position, next_position}
//   scorer/phrase_scorer.rs:393-430      PPElement ordering (reversed: the heap's top is the smallest (position, offset, ord))
//   scorer/phrase_scorer.rs:432-515      SloppyPhraseScorer::new
//   scorer/phrase_scorer.rs:537-577      phrase_freq
//   scorer/phrase_scorer.rs:590-642      init_phrase_positions / init_simple / init_complex / place_first_positions / advance_pp
//   scorer/phrase_scorer.rs:647-726      advance_rpts / lesser / collide
//   scorer/phrase_scorer.rs:733-790      fill_queue / advance_repeat_groups (single-term arm)
//   scorer/phrase_scorer.rs:805-871, 909-946  init_first_time / sort_rpt_groups / gather_rpt_groups (single-term arm) /
//                                        repeating_terms / repeating_pps
//   scorer/phrase_scorer.rs:1008-1071    score = sim.score(doc, sloppy_freq); matches(): sloppy_freq > f32::EPSILON; two-phase next
//   scorer/mod.rs:158-168                two_phase_next
//   similarity/bm25_similarity.rs:65-67  sloppy_freq(distance) = 1.0 / (distance as f32 + 1.0)
//
// Third-party arithmetic on this path: Rust's std::collections::BinaryHeap (toolchain nightly-2020-03-12) holds the
// PPElements. Its order of pops is NOT simply "smallest first" here: advance_rpts moves PhrasePositions that sit INSIDE the
// heap (their keys change in place, the heap is not told), so which element a pop returns depends on the array the heap
// keeps. The heap is therefore emulated operation for operation (push = sift_up; pop = swap the last element into the root,
// sift_down_to_bottom, sift_up), following the copy of that std version the reference vendors (util/external/
// binary_heap.rs:121-210), as oracle/search.hpp does for TopDocsCollector.
#pragma once
#include <cfloat>
#include <climits>
#include <map>
#include <memory>
#include <vector>

#include "phrase.hpp"

namespace orc {

struct SloppyPhraseScorer : Scorer {
  struct PhrasePositions {  // phrase_scorer.rs:318-376
    int32_t position = 0, count = 0, offset = 0, ord = 0;
    BlockPostingIterator* postings = nullptr;
    int32_t rpt_group = -1, rpt_ind = 0;
    int64_t term = 0;  // Term identity (the flat term id stands in for field + bytes)
    void first_position() { count = postings->freq(); next_position(); }
    bool next_position() {
      if (count > 0) { count -= 1; position = postings->next_position() - offset; return true; }
      return false;
    }
  };
  std::vector<std::unique_ptr<BlockPostingIterator>> iterators;
  std::unique_ptr<ConjunctionScorer> conjunction;
  std::vector<PhrasePositions> pps;
  float sloppy_freq_ = 0.0f;
  int32_t slop;
  size_t num_postings;
  std::vector<size_t> pq;  // BinaryHeap<PPElement>::data: indices into pps
  int32_t end = 0;
  bool has_rpts = false, checked_rpts = false;
  std::vector<std::vector<size_t>> rpt_group;
  std::vector<size_t> rpt_stack;
  int32_t num_matches = 0;
  bool needs_scores;
  const BM25Weight* weight;
  const uint8_t* norms;

  // `its[i]` / `offsets[i]` / `terms[i]`: the phrase's terms in QUERY order (phrase_query.rs:324-331 does not sort them)
  SloppyPhraseScorer(std::vector<std::unique_ptr<BlockPostingIterator>> its, const std::vector<int32_t>& offsets, const std::vector<int64_t>& terms,
                     int32_t slop_, const BM25Weight* w, const uint8_t* norms_, bool needs_scores_)
      : iterators(std::move(its)), slop(slop_), num_postings(iterators.size()), needs_scores(needs_scores_), weight(w), norms(norms_) {
    std::vector<ScorerBox> as_scorers;
    for (size_t i = 0; i < iterators.size(); i++) {
      PhrasePositions pp;
      pp.offset = offsets[i];
      pp.ord = (int32_t)i;
      pp.postings = iterators[i].get();
      pp.term = terms[i];
      pps.push_back(pp);
      as_scorers.emplace_back(new PostingsIterAsScorer(iterators[i].get()));
    }
    conjunction.reset(new ConjunctionScorer(std::move(as_scorers)));
  }

  // ---- std BinaryHeap<PPElement> (util/external/binary_heap.rs:121-210) with PPElement's reversed ordering:
  // a <= b  <=>  (a.position, a.offset, a.ord) >= (b.position, b.offset, b.ord)   (phrase_scorer.rs:404-422)
  bool key_less(size_t a, size_t b) const {  // strict lexicographic (position, offset, ord)
    const PhrasePositions &x = pps[a], &y = pps[b];
    if (x.position != y.position) return x.position < y.position;
    if (x.offset != y.offset) return x.offset < y.offset;
    return x.ord < y.ord;
  }
  bool le(size_t a, size_t b) const { return !key_less(a, b); }   // PartialOrd <=
  bool gt(size_t a, size_t b) const { return key_less(a, b); }    // PartialOrd >
  size_t sift_up(size_t start, size_t pos) {
    const size_t elt = pq[pos];
    while (pos > start) {
      const size_t parent = (pos - 1) / 2;
      if (le(elt, pq[parent])) break;
      pq[pos] = pq[parent];
      pos = parent;
    }
    pq[pos] = elt;
    return pos;
  }
  void sift_down_to_bottom(size_t pos) {
    const size_t end_ = pq.size(), start = pos;
    const size_t elt = pq[pos];
    size_t child = 2 * pos + 1;
    while (child < end_) {
      const size_t right = child + 1;
      if (right < end_ && !gt(pq[child], pq[right])) child = right;  // the greater of the two children
      pq[pos] = pq[child];
      pos = child;
      child = 2 * pos + 1;
    }
    pq[pos] = elt;
    sift_up(start, pos);
  }
  void heap_push(size_t idx) { pq.push_back(idx); sift_up(0, pq.size() - 1); }
  size_t heap_pop() {
    if (pq.empty()) throw OracleError(E_ILLEGAL_STATE, "pop from an empty heap (unwrap on None)");
    size_t item = pq.back();
    pq.pop_back();
    if (!pq.empty()) { std::swap(item, pq[0]); sift_down_to_bottom(0); }
    return item;
  }
  size_t heap_peek() const {
    if (pq.empty()) throw OracleError(E_ILLEGAL_STATE, "peek into an empty heap (unwrap on None)");
    return pq[0];
  }

  static float slop_factor(int32_t distance) { return 1.0f / ((float)distance + 1.0f); }  // bm25_similarity.rs:65-67

  // phrase_scorer.rs:537-577
  float phrase_freq() {
    if (!init_phrase_positions()) return 0.0f;
    float freq = 0.0f;
    num_matches = 0;
    size_t pp_idx = heap_pop();
    int32_t match_length = end - pps[pp_idx].position;
    int32_t next = pps[heap_peek()].position;
    while (advance_pp(pp_idx)) {
      if (has_rpts && !advance_rpts(pp_idx)) break;  // pps exhausted
      if (pps[pp_idx].position > next) {  // done minimizing current match-length
        if (match_length <= slop) {
          freq += slop_factor(match_length);
          num_matches += 1;
          if (!needs_scores) return freq;
        }
        heap_push(pp_idx);
        pp_idx = heap_pop();
        next = pps[heap_peek()].position;
        match_length = end - pps[pp_idx].position;
      } else {
        const int32_t match_length2 = end - pps[pp_idx].position;
        match_length = std::min(match_length, match_length2);
      }
    }
    if (match_length <= slop) {
      freq += slop_factor(match_length);
      num_matches += 1;
    }
    return freq;
  }
  // phrase_scorer.rs:590-600
  bool init_phrase_positions() {
    end = INT32_MIN;
    if (!checked_rpts) return init_first_time();
    if (!has_rpts) { init_simple(); return true; }
    return init_complex();
  }
  void init_simple() {  // :604-617
    pq.clear();
    for (size_t idx = 0; idx < num_postings; idx++) {
      pps[idx].first_position();
      if (pps[idx].position > end) end = pps[idx].position;
      heap_push(idx);
    }
  }
  bool init_complex() {  // :620-627
    place_first_positions();
    if (!advance_repeat_groups()) return false;
    fill_queue();
    return true;
  }
  void place_first_positions() { for (PhrasePositions& pp : pps) pp.first_position(); }  // :630-635
  bool advance_pp(size_t idx) {  // :638-646
    if (!pps[idx].next_position()) return false;
    if (pps[idx].position > end) end = pps[idx].position;
    return true;
  }
  static int32_t tp_pos(const PhrasePositions& pp) { return pp.position + pp.offset; }  // :880-882
  size_t lesser(size_t a, size_t b) const {  // :704-713
    const PhrasePositions &x = pps[a], &y = pps[b];
    return (x.position < y.position || (x.position == y.position && x.offset < y.offset)) ? a : b;
  }
  int32_t collide(size_t idx) const {  // :716-726
    const PhrasePositions& pp = pps[idx];
    const int32_t tp = tp_pos(pp);
    for (size_t i : rpt_group[(size_t)pp.rpt_group]) {
      const PhrasePositions& pp2 = pps[i];
      if (idx != i && tp_pos(pp2) == tp) return pp2.rpt_ind;
    }
    return -1;
  }
  // phrase_scorer.rs:651-701
  bool advance_rpts(size_t pp_idx) {
    if (pps[pp_idx].rpt_group < 0) return true;  // not a repeater
    const std::vector<size_t>& rg = rpt_group[(size_t)pps[pp_idx].rpt_group];
    const size_t num_bits = rg.size();       // FixedBitSet::new(len): ensure_capacity(k) with k < len never grows it
    std::vector<char> bits(num_bits, 0);
    size_t cardinality = 0;
    const int32_t k0 = pps[pp_idx].rpt_ind;
    size_t cur = pp_idx;
    while (true) {
      const int32_t k = collide(cur);
      if (k < 0) break;
      cur = lesser(cur, rg[(size_t)k]);  // always advance the lesser of the (only) two colliding pps
      if (!advance_pp(cur)) return false;
      if (k != k0 && !bits[(size_t)k]) { bits[(size_t)k] = 1; cardinality++; }  // mark only those currently in the queue
    }
    // collisions resolved, now re-queue: empty (partially) the queue until seeing all pps advanced for resolving collisions
    size_t n = 0;
    while (cardinality > 0) {
      const size_t pp2 = heap_pop();
      rpt_stack[n++] = pp2;
      const PhrasePositions& p2 = pps[pp2];
      if (p2.rpt_group >= 0 && p2.rpt_ind < (int32_t)num_bits && bits[(size_t)p2.rpt_ind]) { bits[(size_t)p2.rpt_ind] = 0; cardinality--; }
    }
    for (size_t i = 0; i < n; i++) heap_push(rpt_stack[n - 1 - i]);  // add back to queue
    return true;
  }
  void fill_queue() {  // :733-745
    pq.clear();
    int32_t e = end;
    for (size_t idx = 0; idx < pps.size(); idx++) {
      if (pps[idx].position > e) e = pps[idx].position;
      heap_push(idx);
    }
    end = e;
  }
  bool advance_repeat_groups() {  // :755-790, the arm without multi-term repeats: the j-th pp of a group advances j times
    for (const std::vector<size_t>& rg : rpt_group)
      for (size_t j = 1; j < rg.size(); j++)
        for (size_t k = 0; k < j; k++)
          if (!pps[rg[j]].next_position()) return false;  // PPs exhausted
    return true;
  }
  // phrase_scorer.rs:805-822: done once, on the first candidate doc
  bool init_first_time() {
    checked_rpts = true;
    place_first_positions();
    // repeating_terms (:909-931): terms that occur in more than one pp, numbered as their second occurrence is met
    std::map<int64_t, size_t> tcnt, tord;
    for (const PhrasePositions& pp : pps) {
      const size_t cnt = ++tcnt[pp.term];
      if (cnt == 2) { const size_t ord = tord.size(); tord[pp.term] = ord; }
    }
    has_rpts = !tord.empty();
    if (has_rpts) {
      rpt_stack.assign(num_postings, 0);
      // repeating_pps (:934-946)
      std::vector<size_t> rpp;
      for (size_t idx = 0; idx < pps.size(); idx++) if (tord.count(pps[idx].term)) rpp.push_back(idx);
      // gather_rpt_groups (:841-871), no multi-terms: can base on positions in first doc
      std::vector<std::vector<size_t>> res;
      for (size_t i = 0; i < rpp.size(); i++) {
        const size_t idx1 = rpp[i];
        if (pps[idx1].rpt_group >= 0) continue;  // already marked as a repetition
        const int32_t tp = tp_pos(pps[idx1]);
        for (size_t jj = i + 1; jj < rpp.size(); jj++) {
          const size_t idx2 = rpp[jj];
          if (pps[idx2].rpt_group >= 0 || pps[idx2].offset == pps[idx1].offset || tp_pos(pps[idx2]) != tp) continue;
          int32_t g = pps[idx1].rpt_group;  // a repetition
          if (g < 0) {
            g = (int32_t)res.size();
            pps[idx1].rpt_group = g;
            res.push_back(std::vector<size_t>{idx1});
          }
          pps[idx2].rpt_group = g;
          res[(size_t)g].push_back(idx2);
        }
      }
      // sort_rpt_groups (:826-838): by (query) offset; the index in the group is kept for re-queuing
      for (std::vector<size_t>& rg : res) {
        std::stable_sort(rg.begin(), rg.end(), [&](size_t a, size_t b) { return pps[a].offset < pps[b].offset; });
        for (size_t j = 0; j < rg.size(); j++) pps[rg[j]].rpt_ind = (int32_t)j;
        rpt_group.push_back(rg);
      }
      if (!advance_repeat_groups()) return false;  // PPs exhausted
    }
    fill_queue();
    return true;
  }

  // ---- Scorer / DocIterator (phrase_scorer.rs:1008-1071)
  bool matches() override {
    sloppy_freq_ = phrase_freq();
    return sloppy_freq_ > FLT_EPSILON;
  }
  bool support_two_phase() const override { return true; }                       // :1060-1062
  int32_t approximate_next() override { return conjunction->next(); }             // :1064-1066
  int32_t approximate_advance(int32_t t) override { return conjunction->advance(t); }  // :1068-1070
  int32_t two_phase_next() {  // scorer/mod.rs:158-168
    int32_t doc = conjunction->doc_id();
    while (true) {
      if (doc == NO_MORE_DOCS) return NO_MORE_DOCS;
      if (matches()) return doc;
      doc = conjunction->next();
    }
  }
  float sloppy_freq() const { return sloppy_freq_; }
  float score() override {
    const int32_t d = conjunction->doc_id();
    return bm25_compute_score(weight->weight, weight->k1, sloppy_freq_, norms != nullptr, norms ? weight->cache[norms[d] & 0xFF] : 0.0f);
  }
  int32_t doc_id() const override { return conjunction->doc_id(); }
  int32_t next() override { conjunction->next(); return two_phase_next(); }
  int32_t advance(int32_t target) override { conjunction->advance(target); return two_phase_next(); }
  size_t cost() const override { return conjunction->cost(); }
};

}  // namespace orc
