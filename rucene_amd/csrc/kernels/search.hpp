// Fused query-evaluation kernels: block decode -> doc ids -> BM25 -> top-k, nothing materialised in HBM.
// GPU counterparts of (paths relative to /root/reference/src/core):
//   search/scorer/term_scorer.rs:43-67              TermScorer            -> search_term.hpp (k_search_term)
//   search/scorer/conjunction_scorer.rs:26-128      ConjunctionScorer     -> search_and.hpp (k_search_and)
//   search/scorer/disjunction_scorer.rs:24-104      DisjunctionSumScorer  -> search_or.hpp (k_score_terms, k_or_windows)
//   search/similarity/bm25_similarity.rs:203-212    BM25SimScorer::compute_score (f32, left to right)
//   search/scorer/bulk_scorer.rs:114-120            the per-leaf collect loop incl. the live-docs test
//   search/collector/top_docs.rs:67-94,157-172      TopDocsCollector::{add_doc, collect, finish_parallel}
//
// Top-k is kept per wavefront in registers as a sorted list of u64 keys (score order bits << 32 | ~doc), so
// "score desc, doc asc" is one integer compare; lists are merged by k_merge_items. Results do not depend on
// scheduling: the key order is total and every collected doc is offered exactly once.
#pragma once
#include <type_traits>

#include "decode_terms.hpp"

namespace rgpu {

struct HitOut {
  int32_t doc;
  float score;
};

// BM25SimScorer::compute_score: weight * (k1 + 1) * freq / (freq + norm)   — every step rounded to f32
__device__ __forceinline__ float bm25_score(float weight_k1p1, float freq, float norm) {
  return weight_k1p1 * freq / (freq + norm);
}

__device__ __forceinline__ bool doc_is_live(const uint64_t* __restrict__ live, int32_t doc) {
  return live == nullptr || ((live[doc >> 6] >> (doc & 63)) & 1ull);  // util/bit_set.rs:453-460
}
// A VInt tail or a singleton is decoded at query time from bytes nobody validated (FullBlocks are checked once by
// k_prepare_blocks): a doc id outside [0, max_doc) from a corrupt file must not turn into a wild gather. Such a posting
// simply is not collected (the reference would fail its bounds check; garbage in a corrupt index is not parity).
__device__ __forceinline__ bool doc_in_segment(const SegView& seg, int32_t doc) { return (uint32_t)doc < (uint32_t)seg.max_doc; }
__device__ __forceinline__ uint32_t norm_at(const SegView& seg, int32_t doc) { return doc_in_segment(seg, doc) ? seg.norms[doc] : 0u; }

// cache[] is indexed by whatever seg.norms holds: raw norm bytes (256 entries) or norm ranks (<= 64 entries)
__device__ __forceinline__ void load_sim_table(const SegView& seg, int id, float* cache, int lane, float& k1) {
  const float* src = seg.sim_tables + (size_t)id * 257;
  if (seg.n_norm_ranks > 0) {
    cache[lane] = src[seg.rank_to_norm[lane]];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) cache[lane + 64 * i] = src[lane + 64 * i];
  }
  k1 = src[256];
  wave_sync();
}

// Per-clause score table over (norm rank, freq 0..SCORE_TABLE_FREQS): Rucene clamps term freqs to 10 at write
// time (codec/postings/mod.rs:82), so for its own indexes every posting's BM25 score is one LDS read; each entry
// is produced by the same f32 expression as bm25_score, i.e. bit-identical (the freq-0 column only exists so
// that a block whose freq width is <= 3 bits needs no per-posting range test at all). Lives right after the
// 64 cache entries of the wave's LDS slice; lane r fills row r.
constexpr int SCORE_TABLE_FREQS = 10;
constexpr int SCORE_TABLE_COLS = SCORE_TABLE_FREQS + 1;
constexpr int WAVE_CACHE_FLOATS = 64 + 64 * SCORE_TABLE_COLS;  // >= 256 (raw-norm mode uses the first 256)
__device__ __forceinline__ void build_score_table(float* cache, float wk, int lane) {
  const float nrm = cache[lane];
  float* row = cache + 64 + lane * SCORE_TABLE_COLS;
#pragma unroll
  for (int f = 0; f <= SCORE_TABLE_FREQS; ++f) row[f] = bm25_score(wk, (float)f, nrm);
  wave_sync();
}
__device__ __forceinline__ float table_score(const float* cache, uint32_t rank, uint32_t freq) {
  return cache[64 + rank * SCORE_TABLE_COLS + freq];
}

#ifndef RGPU_MERGE_AHEAD
#define RGPU_MERGE_AHEAD 4  // lists of entering items requested together by k_merge_items (1: one after the other)
#endif
// ---- fold the per-item lists of each query: one wavefront per query --------------------------------------------
// `head_items` > 0 selects the TERM kernel's item layout: item q is query q's first chunk and the query's other
// chunks are items head_items + [item_prefix[q], item_prefix[q+1]); 0 = plain contiguous ranges.
// The fold itself — also run by the wavefront of k_search_term that finishes a query last (search_term.hpp, TermMerge):
// COHERENT: the lists were written by other wavefronts of THIS launch, through other XCDs' L2s, which are not coherent with this one's
// inside a launch. The one mechanism that is: read-modify-write atomics at agent scope, performed at the memory side. The writers
// EXCHANGE their keys and counts in (search_term.hpp), this reader fetches them with an OR of zero. (Measured first, round 6: agent-scope
// atomic STORES + a wait for their acknowledgement on the writer's side, an acquire fence + plain loads — or agent-scope atomic LOADS —
// on the reader's: scripts/fold_race_probe.py, batches of alternating item layouts, found a stale count or list in 5 of 3000
// launches either way: an acknowledged store has reached the writer's L2, not the memory behind it.) false: a finished launch's lists.
template <bool WIDE, bool COHERENT = false>
__device__ __forceinline__ void merge_query_items(int q, int64_t i0, int64_t n_mine, int head_items, int k, const uint64_t* partial_keys,
                                                  const int32_t* partial_counts, WaveTopK& top, int64_t& total, int lane) {
  auto key_at = [&](const uint64_t* p) -> uint64_t {
    if (!COHERENT) return *p;
    return (uint64_t)__hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(p)), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto count_at = [&](const int32_t* p) -> int32_t {
    if (!COHERENT) return *p;
    return __hip_atomic_fetch_or(const_cast<int32_t*>(p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  uint64_t tau = 0;
  auto item_at = [&](int64_t j) -> int64_t {  // j-th item of this query
    if (head_items > 0) return j == 0 ? (int64_t)q : (int64_t)head_items + i0 + j - 1;
    return i0 + j;
  };
  for (int64_t g0 = 0; g0 < n_mine; g0 += 64) {
    // each lane looks at one item's best key (lists are sorted best-first): an item whose head cannot
    // enter the current top-k is skipped without reading the rest of its list
    const bool ok = g0 + lane < n_mine;
    const int64_t mine = ok ? item_at(g0 + lane) : 0;
    const uint64_t head = ok ? key_at(partial_keys + (size_t)mine * (size_t)k) : 0ull;
    total += wave_reduce_add(ok ? count_at(partial_counts + mine) : 0);
    uint64_t m = __ballot(head > tau);
    while (m) {
      // the lists of up to MERGE_AHEAD entering items are requested together (one list after the other was a chain of dependent
      // round trips: 27 us per launch behind a 0.22 ms conjunction kernel); an item whose head has fallen behind the threshold
      // by the time its turn comes offers keys that all stay outside — same rows either way
      constexpr int MERGE_AHEAD = RGPU_MERGE_AHEAD;
      uint64_t ka[MERGE_AHEAD], kb[MERGE_AHEAD];
      uint64_t rest = m;
#pragma unroll
      for (int j = 0; j < MERGE_AHEAD; ++j) {
        ka[j] = 0ull;
        kb[j] = 0ull;
        if (rest) {  // wave-uniform
          const int src = __builtin_ctzll(rest);
          rest &= rest - 1;
          const uint64_t* pk = partial_keys + (size_t)item_at(g0 + src) * (size_t)k;
          ka[j] = lane < k ? key_at(pk + lane) : 0ull;
          if (WIDE) kb[j] = lane + 64 < k ? key_at(pk + lane + 64) : 0ull;
        }
      }
#pragma unroll
      for (int j = 0; j < MERGE_AHEAD; ++j) topk_offer_sorted<WIDE>(top, ka[j], kb[j], tau, k, lane);  // (lists are sorted: wave.hpp)
      m = rest & __ballot(head > tau);
    }
  }
}
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_merge_items(const int64_t* __restrict__ item_prefix, int n_queries, int k,
                                                            const uint64_t* __restrict__ partial_keys,
                                                            const int32_t* __restrict__ partial_counts, int32_t doc_base,
                                                            int head_items, HitOut* __restrict__ hits_out,
                                                            int64_t* __restrict__ totals_out,
                                                            const int2* __restrict__ fixed_info = nullptr,
                                                            int32_t* __restrict__ low_flags = nullptr,
                                                            const int32_t* __restrict__ qmap = nullptr, int out_stride = 0, int col0 = 0,
                                                            unsigned long long* __restrict__ ceil_out = nullptr) {
  // out_stride / col0: the caller's rows are out_stride hits long and this pass fills columns [col0, col0 + k) (0 = k, 0);
  // ceil_out[row] = this pass's worst key when it filled all k slots, else 0 ("nothing left below") — the next pass's ceiling
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  WaveTopK top;
  int64_t total = 0;
  const int64_t i0 = item_prefix[q], i1 = item_prefix[q + 1];
  merge_query_items<WIDE>(q, i0, (i1 - i0) + (head_items > 0 ? 1 : 0), head_items, k, partial_keys, partial_counts, top, total, lane);
  // qmap: the group's query q is the caller's row qmap[q] (queries are partitioned by op on the host) — the rows are
  // written in place, no scatter pass
  const int row = qmap ? qmap[q] : q;
  HitOut* out = hits_out + (size_t)row * (size_t)(out_stride > 0 ? out_stride : k) + col0;
  if (ceil_out != nullptr) {
    const uint64_t kth = topk_threshold<WIDE>(top, k);
    if (lane == 0) ceil_out[row] = kth;
  }
  if (fixed_info != nullptr) {
    // k_or_wide's keys: the high word is a fixed-point total (search_or_wide.hpp). score = total * 2^-e, rounded to f32
    // once; a hit whose total is below the query's floor asks for the f32 path (low_flags)
    const int2 info = fixed_info[q];
    auto score_of = [&](uint64_t key) { return (float)ldexp((double)(uint32_t)(key >> 32), -info.x); };
    // Totals reach 2^31 and f32 keeps 24 bits: distinct totals may round to ONE f32 score, and the row must still come out
    // in the canonical TopDocs order of what the caller sees (score desc, doc asc) — so the (at most k) survivors are ranked
    // once more on (f32 score, doc). Which docs survive was decided on the exact totals; a doc at the boundary may thus
    // beat one whose ROUNDED score equals its own but whose doc id is smaller — inside the 1e-5 band this path is pinned to.
    {
      WaveTopK ranked;
      uint64_t t2 = 0;
      topk_offer<WIDE>(ranked, top.a ? make_key(score_of(top.a), key_doc(top.a)) : 0ull, t2, k, lane);
      if (WIDE) topk_offer<WIDE>(ranked, top.b ? make_key(score_of(top.b), key_doc(top.b)) : 0ull, t2, k, lane);
      if (lane < k) out[lane] = ranked.a ? HitOut{key_doc(ranked.a) + doc_base, key_score(ranked.a)} : HitOut{-1, 0.f};
      if (WIDE && lane + 64 < k) out[lane + 64] = ranked.b ? HitOut{key_doc(ranked.b) + doc_base, key_score(ranked.b)} : HitOut{-1, 0.f};
    }
    const bool low =(lane < k && top.a != 0 && (uint32_t)(top.a >> 32) < (uint32_t)info.y) ||
                     (WIDE && lane + 64 < k && top.b != 0 && (uint32_t)(top.b >> 32) < (uint32_t)info.y);
    const uint64_t any_low = __ballot(low);
    if (lane == 0) { totals_out[row] = total; low_flags[q] = any_low ? 1 : 0; }
    return;
  }
  if (lane < k) out[lane] = top.a ? HitOut{key_doc(top.a) + doc_base, key_score(top.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = top.b ? HitOut{key_doc(top.b) + doc_base, key_score(top.b)} : HitOut{-1, 0.f};
  if (lane == 0) totals_out[row] = total;
}

// A query with hundreds of item lists (a small batch of >= 10-clause disjunctions: one list per 16384-doc window — 612 for a
// 10 M-doc segment) folded by ONE wavefront of k_merge_items is a serial chain of insertions (167 us for a single query's 612
// lists of up to 100 keys, behind 290 us of search). This pass folds groups of `group` consecutive lists in parallel, one
// wavefront per group: the group's first list receives the merged keys (best first, as every list) and the summed count, the
// others are emptied (head key 0, count 0) — k_merge_items then meets items / group lists. Plain item ranges only.
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_premerge_items(const int64_t* __restrict__ item_prefix, int n_queries, int k, int group,
                                                               int groups_per_query, uint64_t* __restrict__ partial_keys,
                                                               int32_t* __restrict__ partial_counts) {
  const int lane = lane_id();
  const int64_t w = (int64_t)blockIdx.x * WG_WAVES + wave_id();
  const int q = (int)(w / groups_per_query);
  if (q >= n_queries) return;
  const int64_t i0 = item_prefix[q] + (w % groups_per_query) * (int64_t)group;
  const int64_t i1 = min(item_prefix[q + 1], i0 + group);
  if (i1 - i0 < 2) return;
  WaveTopK top;
  uint64_t tau = 0;
  int total = 0;
  for (int64_t g0 = i0; g0 < i1; g0 += 4) {  // four lists requested together
    uint64_t ka[4], kb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t* pk = partial_keys + (size_t)min(g0 + j, i1 - 1) * (size_t)k;
      const bool real = g0 + j < i1;  // wave-uniform
      ka[j] = (real && lane < k) ? pk[lane] : 0ull;
      kb[j] = (WIDE && real && lane + 64 < k) ? pk[lane + 64] : 0ull;
      total += real ? partial_counts[min(g0 + j, i1 - 1)] : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) topk_offer_sorted<WIDE>(top, ka[j], kb[j], tau, k, lane);
  }
  uint64_t* first = partial_keys + (size_t)i0 * (size_t)k;
  if (lane < k) first[lane] = top.a;
  if (WIDE && lane + 64 < k) first[lane + 64] = top.b;
  for (int64_t i = i0 + 1 + lane; i < i1; i += 64) { partial_keys[(size_t)i * (size_t)k] = 0ull; partial_counts[i] = 0; }
  if (lane == 0) partial_counts[i0] = total;
}

// TopDocsCollector::finish_parallel across leaves / shards: list l's rows start at hits_in + l * hits_stride
// ([query][k], already in global doc ids) and its hit counts at totals_in + l * totals_stride — [list][query][k] and
// [list][query] arrays, or the records of one all-gather ([list][hits | totals | status]). One wavefront per query; k > 128 in
// passes of 128 inside the kernel (each pass re-reads the lists and keeps what lies below the previous pass's worst key).
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_merge_lists(const HitOut* __restrict__ hits_in, const int64_t* __restrict__ totals_in,
                                                            int64_t hits_stride, int64_t totals_stride,
                                                            int n_lists, int n_queries, int k, HitOut* __restrict__ hits_out,
                                                            int64_t* __restrict__ totals_out) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  int64_t total = 0;
  for (int l = 0; l < n_lists; ++l) total += totals_in[(size_t)l * (size_t)totals_stride + q];
  HitOut* out = hits_out + (size_t)q * (size_t)k;
  uint64_t ceil = ~0ull;
  for (int col0 = 0; col0 < k; col0 += 128) {
    const int kp = min(128, k - col0);
    WaveTopK top;
    uint64_t tau = 0;
    for (int l = 0; l < n_lists; ++l) {
      const HitOut* in = hits_in + (size_t)l * (size_t)hits_stride + (size_t)q * (size_t)k;
      for (int r = 0; r < k; r += 64) {
        uint64_t key = 0;
        if (r + lane < k) { const HitOut h = in[r + lane]; if (h.doc >= 0) key = below(make_key(h.score, h.doc), ceil); }
        if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, kp, lane);
      }
    }
    if (lane < kp) out[col0 + lane] = top.a ? HitOut{key_doc(top.a), key_score(top.a)} : HitOut{-1, 0.f};
    if (WIDE && lane + 64 < kp) out[col0 + lane + 64] = top.b ? HitOut{key_doc(top.b), key_score(top.b)} : HitOut{-1, 0.f};
    ceil = topk_threshold<WIDE>(top, kp);  // 0 when this pass did not fill up: nothing is left for the next one
  }
  if (lane == 0) totals_out[q] = total;
}

// rows of `stride` hits, columns [col0, col0 + k) set to "no hit"
__global__ void k_init_hits(HitOut* __restrict__ hits, int64_t n_rows, int k, int stride, int col0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows * k) hits[(i / k) * stride + col0 + (i % k)] = HitOut{-1, 0.f};
}

}  // namespace rgpu
