// The batch planner: a whole batch of term / boolean queries — named by term bytes or by flat-table ids, as arrays —
// turned into the rgpu_query[] / rgpu_query_term[] that rgpu_search_batch* takes. Host-only, no GPU work besides one
// rgpu_sim_table_upload when the planner is created.
//
// What it replaces, per query per leaf, in the reference (paths relative to /root/reference/src/core):
//   search/query/term_query.rs:58-95        TermQuery::create_weight  -> term statistics, BM25Similarity::compute_weight
//   search/searcher.rs:732-767              IndexSearcher::term_statistics (doc_freq of the term in the STATISTICS leaf:
//                                           the first leaf with the largest max_doc, :306-363)
//   search/query/term_query.rs:145-163      TermWeight::create_scorer -> the leaf's TermIterator::seek_exact + term_state
//   search/similarity/bm25_similarity.rs:99-114, 151-177   idf (f64 log -> f32), weight = idf * boost, cache[256]
//   search/query/boolean_query.rs:40-86     BooleanQuery::build's single-clause rewrite
// The reference does this one query at a time behind trait objects; a serving host hands the GPU a thousand queries per
// call, so here it is three passes over flat arrays: resolve (one software-pipelined dictionary batch lookup, or a gather
// from the flat table), weigh (idf memoised by doc_freq: a Zipfian batch repeats few values, and the f64 log is the one
// expensive operation on the path), pack.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../../include/rucene_gpu.h"
#include "bm25_similarity.hpp"
#include "term_dict.hpp"

namespace rucene {

class BatchPlanner {
 public:
  // flat-table form: ids index `leaf` (states of the leaf being searched) and `stats` (states of the statistics leaf; the
  // same table when `stats` is null). The tables are copied.
  BatchPlanner(const rgpu_plan_stats& ps, int32_t sim_table, const rgpu_term_state* leaf, int64_t n_leaf, const rgpu_term_state* stats,
               int64_t n_stats)
      : ps_(ps), sim_table_(sim_table), flat_(true), leaf_states_(leaf, leaf + n_leaf) {
    if (stats) {
      stats_df_.resize((size_t)n_stats);
      for (int64_t i = 0; i < n_stats; ++i) stats_df_[(size_t)i] = stats[i].doc_freq;
      own_stats_ = true;
    }
    init();
  }
  // dictionary form: terms are bytes, resolved through the leaf's block-tree dictionary; doc_freq for the weight through
  // the statistics leaf's (null: the same leaf)
  BatchPlanner(const rgpu_plan_stats& ps, int32_t sim_table, const TermDictionary* leaf, const TermDictionary* stats, int32_t field_number)
      : ps_(ps), sim_table_(sim_table), flat_(false), leaf_dict_(leaf), stats_dict_(stats), field_(field_number) {
    init();
  }

  bool flat() const { return flat_; }
  int32_t sim_table() const { return sim_table_; }
  void set_sim_table(int32_t t) { std::lock_guard<std::mutex> g(mu_); sim_table_ = t; }

  // ops[q]: rgpu_query.op as the search takes it (RGPU_OP_* with the min_should_match / optional-SHOULD bytes);
  // n_terms[q] / n_must_not[q] as in rgpu_query; clause order per query: scored, optional SHOULD, MUST_NOT. Exactly one of
  // `ids` / (`bytes`, `offsets`) names the sum-of-clauses terms. Returns 0 or a negative rgpu_status (+ *why).
  int plan(int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not, const int64_t* ids,
           const uint8_t* bytes, const int64_t* offsets, const float* boosts, rgpu_query* queries_out, rgpu_query_term* terms_out,
           int64_t terms_cap, std::string* why) {
    if (flat_ != (ids != nullptr)) { *why = flat_ ? "this planner names terms by flat-table id" : "this planner names terms by their bytes"; return RGPU_ERR_ILLEGAL_ARGUMENT; }
    int64_t total = 0;
    for (int32_t q = 0; q < n_queries; ++q) {
      const int32_t op = ops[q] & 0xff, n_opt = (ops[q] >> 16) & 0xff;
      const int32_t nt = n_terms[q], nn = n_must_not ? n_must_not[q] : 0;
      if (op < RGPU_OP_TERM || op > RGPU_OP_OR || (ops[q] & ~(0xffffff | RGPU_OP_SHOULD_REQUIRED | RGPU_OP_NESTED_MUST | RGPU_OP_NESTED_AT(63))) != 0) { *why = "unknown query op"; return RGPU_ERR_ILLEGAL_ARGUMENT; }
      if (nt < 0 || nn < 0 || (op == RGPU_OP_TERM && nt != 1)) { *why = "bad clause count"; return RGPU_ERR_ILLEGAL_ARGUMENT; }
      if (nt + n_opt + nn > RGPU_MAX_QUERY_TERMS) { *why = "more than RGPU_MAX_QUERY_TERMS clauses in one query"; return RGPU_ERR_UNSUPPORTED; }
      int32_t out_op = ops[q];
      // BooleanQuery::build: a lone MUST or SHOULD clause (no MUST_NOT, no optional clause) IS that clause
      if (op != RGPU_OP_TERM && nt == 1 && n_opt == 0 && nn == 0) out_op = RGPU_OP_TERM;
      queries_out[q] = rgpu_query{out_op, nt, (int32_t)total, nn};
      total += nt + n_opt + nn;
    }
    if (total > terms_cap || total > 0x7fffffff) { *why = "terms_out is too small for the batch"; return RGPU_ERR_ILLEGAL_ARGUMENT; }
    std::lock_guard<std::mutex> g(mu_);
    df_scratch_.resize((size_t)total);
    // ---- resolve
    if (flat_) {
      const int64_t n_leaf = (int64_t)leaf_states_.size();
      const int64_t n_stats = own_stats_ ? (int64_t)stats_df_.size() : n_leaf;
      for (int64_t i = 0; i < total; ++i) {
        const int64_t id = ids[i];
        rgpu_term_state& st = terms_out[i].state;
        if (id >= 0 && id < n_leaf && leaf_states_[(size_t)id].doc_freq > 0) st = leaf_states_[(size_t)id];
        else st = absent();
        int32_t df = 0;  // searcher.rs:746-760: 0 when the term is absent from the statistics leaf
        if (id >= 0 && id < n_stats) df = own_stats_ ? stats_df_[(size_t)id] : leaf_states_[(size_t)id].doc_freq;
        df_scratch_[(size_t)i] = df > 0 ? df : 0;
      }
    } else {
      states_scratch_.resize((size_t)total);
      found_scratch_.resize((size_t)total);
      leaf_dict_->lookup_batch(field_, bytes, offsets, total, states_scratch_.data(), found_scratch_.data());
      for (int64_t i = 0; i < total; ++i) {
        static_assert(sizeof(TermState) == sizeof(rgpu_term_state), "layout");
        if (found_scratch_[(size_t)i]) std::memcpy(&terms_out[i].state, &states_scratch_[(size_t)i], sizeof(rgpu_term_state));
        else terms_out[i].state = absent();
        df_scratch_[(size_t)i] = found_scratch_[(size_t)i] ? states_scratch_[(size_t)i].doc_freq : 0;
      }
      if (stats_dict_ && stats_dict_ != leaf_dict_) {
        stats_dict_->lookup_batch(field_, bytes, offsets, total, states_scratch_.data(), found_scratch_.data());
        for (int64_t i = 0; i < total; ++i) df_scratch_[(size_t)i] = found_scratch_[(size_t)i] ? states_scratch_[(size_t)i].doc_freq : 0;
      }
    }
    // ---- weigh + pack
    for (int64_t i = 0; i < total; ++i) {
      const float idf = idf_of(df_scratch_[(size_t)i]);
      terms_out[i].weight = boosts ? idf * boosts[i] : idf;  // BM25SimWeight::weight = idf * boost (do_normalize); x * 1.0f == x
      terms_out[i].sim_table = sim_table_;
    }
    return RGPU_OK;
  }

  // for_each_flat with a memo of what the caller makes of a term (round 6: the fused TERM call was bound by its calling thread, and half
  // of that was term id -> state -> idf -> prepared-term look-up -> device descriptor, the same answer batch after batch). A direct-
  // mapped table of 65536 records keyed by term id; `key` names everything outside this planner that a record depends on (the caller's:
  // segment, prepared-store epoch, similarity table, flags) — another key empties the table (a generation number: O(1)). make(state, idf, &rec) -> 1: rec is the
  // term's record (kept), 0: the leaf does not hold the term (kept), < 0: give up (nothing kept, returns false at once);
  // use(i, rec or null) in id order. Under the planner's lock, like for_each_flat.
  struct MemoKey { uint64_t w[4]; bool operator==(const MemoKey& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; } };
  template <class Rec, class Make, class Use>
  bool for_each_flat_memo(const int64_t* ids, int64_t n, const MemoKey& key, Make&& make, Use&& use) {
    static_assert(std::is_trivially_copyable<Rec>::value && sizeof(Rec) <= FLAT_MEMO_REC, "a plain record of at most FLAT_MEMO_REC bytes");
    std::lock_guard<std::mutex> g(mu_);
    if (flat_memo_.empty() || !(flat_memo_key_ == key) || flat_memo_rec_ != sizeof(Rec)) {
      // another key: every record is stale. A generation number per slot makes that one increment, not a pass over 5 MB — a
      // workload whose other batches keep preparing new terms changes the key in front of every call
      if (flat_memo_.empty() || ++flat_memo_gen_ == 0) {
        flat_memo_.assign(FLAT_MEMO_SLOTS, FlatMemoSlot{});  // (gen 0 everywhere)
        flat_memo_gen_ = 1;
      }
      flat_memo_key_ = key;
      flat_memo_rec_ = sizeof(Rec);
    }
    const uint32_t gen = flat_memo_gen_;
    const int64_t n_leaf = (int64_t)leaf_states_.size();
    const int64_t n_stats = own_stats_ ? (int64_t)stats_df_.size() : n_leaf;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t id = ids[i];
      if (id < 0 || id >= n_leaf) { use(i, static_cast<const Rec*>(nullptr)); continue; }
      FlatMemoSlot& e = flat_memo_[(size_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> (64 - FLAT_MEMO_BITS))];
      Rec* rec = reinterpret_cast<Rec*>(e.rec);
      if (e.id != id || e.gen != gen) {
        const rgpu_term_state& st = leaf_states_[(size_t)id];
        int32_t held = 0;
        if (st.doc_freq > 0) {
          int32_t df = 0;
          if (id < n_stats) df = own_stats_ ? stats_df_[(size_t)id] : st.doc_freq;
          held = make(st, idf_of(df > 0 ? df : 0), rec);
          if (held < 0) { e.gen = 0; return false; }
        }
        e.id = id;
        e.gen = gen;
        e.held = held;
      }
      use(i, e.held ? rec : static_cast<const Rec*>(nullptr));
    }
    return true;
  }

  // The flat-table planner's per-clause work WITHOUT the arrays in between (the fused plan + search entry points write device
  // descriptors straight from it): f(i, state, weight) for every id, boost 1, under the planner's lock. An id outside the
  // table, or a term the leaf does not hold, arrives with doc_freq = 0 (TermWeight::create_scorer -> None).
  template <class F>
  void for_each_flat(const int64_t* ids, int64_t n, F&& f) {
    std::lock_guard<std::mutex> g(mu_);
    const int64_t n_leaf = (int64_t)leaf_states_.size();
    const int64_t n_stats = own_stats_ ? (int64_t)stats_df_.size() : n_leaf;
    const rgpu_term_state none = absent();
    for (int64_t i = 0; i < n; ++i) {
      const int64_t id = ids[i];
      const rgpu_term_state& st = (id >= 0 && id < n_leaf && leaf_states_[(size_t)id].doc_freq > 0) ? leaf_states_[(size_t)id] : none;
      int32_t df = 0;
      if (id >= 0 && id < n_stats) df = own_stats_ ? stats_df_[(size_t)id] : leaf_states_[(size_t)id].doc_freq;
      f(i, st, idf_of(df > 0 ? df : 0));
    }
  }

 private:
  static rgpu_term_state absent() { return rgpu_term_state{0, -1, 0, 0, -1}; }
  void init() {
    cs_.max_doc = ps_.max_doc;
    cs_.doc_count = ps_.doc_count;
    cs_.sum_total_term_freq = ps_.sum_total_term_freq;
    memo_df_.assign(MEMO, -1);
    memo_idf_.assign(MEMO, 0.f);
  }
  // idf of ONE term (bm25_similarity.rs:99-114 with a one-element slice: 0.0 + (log as f32)), memoised by doc_freq in a
  // direct-mapped table
  float idf_of(int32_t df) {
    const size_t slot = ((uint32_t)df * 2654435761u) >> (32 - MEMO_BITS);
    if (memo_df_[slot] == df) return memo_idf_[slot];
    TermStatistics ts;
    ts.doc_freq = df;
    const float v = BM25Similarity::idf(&ts, 1, cs_);
    memo_df_[slot] = df;
    memo_idf_[slot] = v;
    return v;
  }

  static constexpr int FLAT_MEMO_BITS = 16;
  static constexpr size_t FLAT_MEMO_SLOTS = (size_t)1 << FLAT_MEMO_BITS;
  static constexpr size_t FLAT_MEMO_REC = 64;
  struct FlatMemoSlot { int64_t id = -1; uint32_t gen = 0; int32_t held = 0; alignas(8) unsigned char rec[FLAT_MEMO_REC] = {}; };
  uint32_t flat_memo_gen_ = 0;  // the generation the current key's records carry (0: no slot is valid)
  std::vector<FlatMemoSlot> flat_memo_;  // (empty until the first for_each_flat_memo: 5 MB)
  MemoKey flat_memo_key_{{0, 0, 0, 0}};
  size_t flat_memo_rec_ = 0;
  static constexpr int MEMO_BITS = 16;
  static constexpr size_t MEMO = (size_t)1 << MEMO_BITS;
  rgpu_plan_stats ps_;
  CollectionStatistics cs_;
  int32_t sim_table_;
  bool flat_;
  bool own_stats_ = false;
  std::vector<rgpu_term_state> leaf_states_;
  std::vector<int32_t> stats_df_;
  const TermDictionary* leaf_dict_ = nullptr;
  const TermDictionary* stats_dict_ = nullptr;
  int32_t field_ = 0;
  std::mutex mu_;
  std::vector<int32_t> memo_df_;
  std::vector<float> memo_idf_;
  std::vector<int32_t> df_scratch_;
  std::vector<TermState> states_scratch_;
  std::vector<uint8_t> found_scratch_;
};

}  // namespace rucene
