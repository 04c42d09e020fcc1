// Disjunction (OR) on the GPU in two launches, nothing shared between wavefronts inside either:
//   1. k_score_terms   every clause's postings are decoded and BM25-scored exactly once (the TermScorer work of
//                      each sub-scorer, term_scorer.rs:43-67) into a {doc, score} run per clause in HBM;
//   2. k_or_windows    each wavefront owns a range of small doc-id windows. Per window it walks every clause's
//                      run from a cursor, adds the scores into an LDS accumulator *in clause order* — the
//                      summation order of SubScorers::score_sum over a SimpleQueue
//                      (search/scorer/disjunction_scorer.rs:213-225) — then scans the window: every touched doc
//                      is one collected hit (bulk_scorer.rs:114-120) offered to the wave's top-k.
// This is DisjunctionSumScorer's doc-at-a-time merge (disjunction_scorer.rs:24-104, util/disi.rs) turned
// term-at-a-time per window; for >= 10 clauses the reference sums in heap order, so only 1e-5 relative holds
// there. No block is decoded twice and no posting is scored twice, whatever the clause density.
#pragma once
#include "search.hpp"

namespace rgpu {

struct ScoredPosting {
  int32_t doc;
  float score;
};

// items = (flat clause index, chunk of blocks); out_prefix[j] = first slot of clause j's run
template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_score_terms(SegView seg, const DevTerm* __restrict__ terms,
                                                            const int64_t* __restrict__ item_prefix,
                                                            const int64_t* __restrict__ out_prefix, int n_terms,
                                                            int64_t n_items, int blocks_per_item,
                                                            ScoredPosting* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ float caches[WG_WAVES][WAVE_CACHE_FLOATS];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (item >= n_items) return;
  const int t = upper_slot(item_prefix, n_terms, item);
  const int chunk = (int)(item - item_prefix[t]);
  const DevTerm T = terms[t];
  uint8_t* slab = slabs[wave];
  float* cache = caches[wave];
  float k1;
  load_sim_table(seg, T.sim_table, cache, lane, k1);
  const float wk = T.weight * (k1 + 1.0f);
  const bool has_norms = seg.norms != nullptr;
  const bool tabled = has_norms && seg.n_norm_ranks > 0;
  if (tabled) build_score_table(cache, wk, lane);
  ScoredPosting* run = out + out_prefix[t];

  auto emit = [&](int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nb0, uint32_t nb1, bool v0, bool v1, int64_t slot) {
    float s0, s1;
    const uint32_t fmax = f0 > f1 ? f0 : f1;
    if (tabled && !__ballot((v0 || v1) && fmax > (uint32_t)SCORE_TABLE_FREQS)) {
      s0 = table_score(cache, nb0, v0 ? f0 : 1u);
      s1 = table_score(cache, nb1, v1 ? f1 : 1u);
    } else {
      s0 = bm25_score(wk, (float)(int32_t)f0, has_norms ? cache[nb0] : k1);
      s1 = bm25_score(wk, (float)(int32_t)f1, has_norms ? cache[nb1] : k1);
    }
    if (v0) run[slot] = ScoredPosting{d0, s0};  // (nontemporal stores were tried here: 8% slower — the runs are re-read next)
    if (v1) run[slot + 1] = ScoredPosting{d1, s1};
  };

  const int b0 = chunk * blocks_per_item;
  const int b1 = min(T.nblocks, b0 + blocks_per_item);
  int32_t base = b0 == 0 ? 0 : seg.dir_last[T.dir_base + b0 - 1];
  const uint8_t* term_rows = seg.bstore + T.bs_base;
  auto on_block = [&](int blk, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nb0, uint32_t nb1) {
    emit(d0, d1, f0, f1, nb0, nb1, true, true, 128 * (int64_t)blk + 2 * lane);
  };
  if (has_norms)
    stream_blocks<LEGACY, true>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, seg.pnorm + T.pn_base, b0, b1, slab, lane, base, on_block);
  else
    stream_blocks<LEGACY, false>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, nullptr, b0, b1, slab, lane, base, on_block);
  if (b1 == T.nblocks) {
    // 64 sentinel entries close every run: k_or_windows reads 64 entries from a cursor without knowing the length
    run[(int64_t)T.df + lane] = ScoredPosting{0x7fffffff, 0.0f};
    if (T.df == 1) {
      const bool v0 = lane == 0;
      const uint32_t nb0 = (has_norms && v0) ? seg.norms[T.singleton_doc] : 0u;
      emit(T.singleton_doc, 0, (uint32_t)T.singleton_freq, 1u, nb0, 0u, v0, false, 0);
    } else if (T.tail_n > 0) {
      const uint32_t toff = T.nblocks ? seg.dir_off[T.dir_base + T.nblocks] : 0u;
      int32_t d0, d1;
      uint32_t f0, f1;
      decode_tail(seg.doc + T.start_fp + toff, T.tail_n, base, slab, lane, d0, d1, f0, f1);
      const bool v0 = 2 * lane < T.tail_n, v1 = 2 * lane + 1 < T.tail_n;
      const uint32_t nb0 = (has_norms && v0) ? seg.norms[d0] : 0u, nb1 = (has_norms && v1) ? seg.norms[d1] : 0u;
      emit(d0, d1, f0, f1, nb0, nb1, v0, v1, 128 * (int64_t)T.nblocks + 2 * lane);
    }
  }
}

constexpr int OR_MAX_TERMS = 16;
constexpr int OR_RUN_PAD = 64;  // sentinel entries {doc = INT_MAX} after every clause's run (written by k_score_terms)
constexpr uint32_t OR_UNTOUCHED = 0xffffffffu;  // accumulator patterns no sum of scores produces (negative quiet NaNs)
constexpr uint32_t OR_EXCLUDED = 0xfffffffeu;

// items = (query, group of `windows_per_item` windows of `W` docs), one per wavefront
// HAS_NOT: some query of the launch carries MUST_NOT clauses; HAS_MSM: some query asks for min_should_match > 1
// (disjunction_scorer.rs:317-329: a doc is a hit only if that many SHOULD clauses hold it — a per-doc clause counter
// next to the accumulator). Separate instantiations keep the common kernel lean.
template <bool WIDE, bool HAS_NOT, bool HAS_MSM>
__global__ __launch_bounds__(WG_THREADS) void k_or_windows(SegView seg, const DevQuery* __restrict__ queries,
                                                           const DevTerm* __restrict__ terms,
                                                           const int64_t* __restrict__ run_prefix,
                                                           const ScoredPosting* __restrict__ runs, int n_queries,
                                                           int windows_per_query, int windows_per_item,
                                                           int items_per_query, int W, int k,
                                                           uint64_t* __restrict__ partial_keys,
                                                           int32_t* __restrict__ partial_counts,
                                                           unsigned long long* __restrict__ tau_slots) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = wave_id();
  // per-wave LDS slice: acc[W] f32 | hits[W] u16 (window offsets of touched docs, in first-touch order). An
  // untouched accumulator holds OR_UNTOUCHED, a NaN pattern no sum of scores produces; the hit scan at the end
  // of a window puts it back, so no per-doc flag array and no clearing pass are needed.
  float* acc = reinterpret_cast<float*>(smem + (size_t)wave * (size_t)W * (HAS_MSM ? 7 : 6));
  uint16_t* hits = reinterpret_cast<uint16_t*>(acc + W);
  uint8_t* cnt = reinterpret_cast<uint8_t*>(hits + W);  // HAS_MSM only: SHOULD clauses that hold the doc
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (item >= (int64_t)n_queries * items_per_query) return;
  const int q = (int)(item / items_per_query);
  const int g = (int)(item - (int64_t)q * items_per_query);
  const DevQuery Q = queries[q];
  const bool has_live = seg.live != nullptr;

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int count = 0;
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);
  const int win0 = g * windows_per_item;
  const int win1 = Q.n_terms > 0 ? min(windows_per_query, win0 + windows_per_item) : win0;

  // Lane t owns one clause's cursor: run base, length, and the first entry with doc >= this item's first window —
  // one lane-parallel binary search over all clauses at once. The n_not MUST_NOT clauses (stored after the n_terms
  // SHOULD clauses) take lanes 0 .. n_not-1 so that a window meets them first: ReqNotScorer over the disjunction
  // (boolean_query.rs:271-273, req_not_scorer.rs:47-63) — their docs are marked excluded before anything is summed.
  const int n_not = HAS_NOT ? Q.pad : 0;
  const int msm = HAS_MSM ? (Q.op >> 8) : 1;
  const bool mine = lane < Q.n_terms + n_not;
  const int my_clause = lane < n_not ? Q.n_terms + lane : lane - n_not;
  const int64_t my_base = mine ? run_prefix[Q.first_term + my_clause] : 0;
  const int my_len = mine ? terms[Q.first_term + my_clause].df : 0;
  int64_t my_at = my_base;  // absolute index of the entry under this clause's cursor (every run ends in sentinels)
  {
    const int32_t first_doc = win0 * W;
    int lo = 0, hi = my_len;
    while (__ballot(lo < hi)) {
      if (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (runs[my_base + mid].doc < first_doc) lo = mid + 1; else hi = mid;
      }
    }
    my_at += lo;
  }
  int32_t my_next = mine ? runs[my_at].doc : 0x7fffffff;  // doc under the cursor (INT_MAX: the run is exhausted)

  for (int i = lane; i < W; i += 64) acc[i] = __uint_as_float(OR_UNTOUCHED);
  wave_sync();
  for (int win = win0; win < win1; ++win) {
    const int32_t w0 = win * W;
    const int32_t w1 = min(seg.max_doc, w0 + W);
    int nhits = 0;
    // only clauses with a posting inside this window are visited, in clause order == summation order. The first
    // 64 run entries of EVERY such clause are requested up front (one exposed load latency per window instead of
    // one per clause: the window walk was latency bound), the rare longer stretches are fetched as they come.
    const uint64_t active0 = __ballot(my_next < w1);
    ScoredPosting pre[OR_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < OR_MAX_TERMS; ++t) {
      pre[t] = ScoredPosting{0x7fffffff, 0.f};
      if ((active0 >> t) & 1ull) pre[t] = runs[(int64_t)readlane64((uint64_t)my_at, t) + lane];  // wave-uniform branch
    }
#pragma unroll
    for (int t = 0; t < OR_MAX_TERMS; ++t) {
      if (!((active0 >> t) & 1ull)) continue;
      int taken = 0;
      int32_t next;
      ScoredPosting e = pre[t];
      while (true) {
        bool in = e.doc < w1;
        const int n = __popcll(__ballot(in));  // runs are doc-sorted: the in-window entries are a prefix
        if (in && has_live) in = doc_is_live(seg.live, e.doc);
        bool first = false;
        if (in) {
          const int o = e.doc - w0;
          const float a = acc[o];
          first = __float_as_uint(a) == OR_UNTOUCHED;
          if (t < n_not) {  // wave-uniform: a prohibited clause only marks
            if (first) acc[o] = __uint_as_float(OR_EXCLUDED);
          } else if (!HAS_NOT || __float_as_uint(a) != OR_EXCLUDED) {
            acc[o] = (first ? 0.0f : a) + e.score;  // 0.0f + s: the reference's `score = 0; score += s`
            if (HAS_MSM) cnt[o] = first ? (uint8_t)1 : (uint8_t)(cnt[o] + 1);
          }
        }
        const uint64_t fm = __ballot(first);
        if (first) hits[nhits + mbcnt(fm)] = (uint16_t)(e.doc - w0);
        nhits += __popcll(fm);
        taken += n;
        if (n < 64) { next = readlane(e.doc, n); break; }  // the entry now under the cursor (or the sentinel)
        e = runs[(int64_t)readlane64((uint64_t)my_at, t) + taken + lane];  // a stretch longer than 64: the rare case
      }
      if (lane == t) { my_at += taken; my_next = next; }
      wave_sync();
    }
    // every touched doc that no prohibited clause claimed is one collected hit
    if (!HAS_NOT && !HAS_MSM) count += nhits;
    for (int i0 = 0; i0 < nhits; i0 += 64) {  // uniform trip count: the offer is a wave-wide operation
      const bool valid = i0 + lane < nhits;
      const int o = valid ? hits[i0 + lane] : 0;
      const float a = valid ? acc[o] : 0.0f;
      const bool hit = valid && (!HAS_NOT || __float_as_uint(a) != OR_EXCLUDED) && (!HAS_MSM || (int)cnt[o] >= msm);
      if (HAS_NOT || HAS_MSM) count += __popcll(__ballot(hit));
      const uint64_t key = hit ? make_key(a, w0 + o) : 0ull;
      if (valid) acc[o] = __uint_as_float(OR_UNTOUCHED);
      if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
    }
    wave_sync();
  }
  shared.publish<WIDE>(top, k, lane);
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  if (lane == 0) partial_counts[item] = count;
}

}  // namespace rgpu
