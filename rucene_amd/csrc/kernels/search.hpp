// Fused query-evaluation kernels: block decode -> doc ids -> BM25 -> top-k, nothing materialised in HBM.
// GPU counterparts of (paths relative to /root/reference/src/core):
//   search/scorer/term_scorer.rs:43-67              TermScorer            -> k_search_term
//   search/scorer/conjunction_scorer.rs:26-128      ConjunctionScorer     -> k_search_window<AND>
//   search/scorer/disjunction_scorer.rs:24-104      DisjunctionSumScorer  -> k_search_window<OR>
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

constexpr int WINDOW_LDS_FIXED = WG_WAVES * SLAB_BYTES + WG_WAVES * 1024 + WG_WAVES * 128 * 8 + 16;  // bytes before acc[]

// ---- AND / OR: items = (query, group of doc-id windows); one workgroup accumulates a window in LDS ------------
// Term-at-a-time inside the window keeps the reference's f32 summation order: AND = cost-sorted
// lead1, lead2, others (conjunction_scorer.rs:87-95, the host sorts clauses by df), OR = clause order
// (SimpleQueue score_sum, disjunction_scorer.rs:213-225). A doc matches AND when every clause touched it.
template <bool LEGACY, bool WIDE, bool IS_AND>
__global__ __launch_bounds__(WG_THREADS) void k_search_window(SegView seg, const DevQuery* __restrict__ queries,
                                                              const DevTerm* __restrict__ terms, int n_queries,
                                                              int windows_per_query, int windows_per_item,
                                                              int items_per_query, int W, int k,
                                                              uint64_t* __restrict__ partial_keys,
                                                              int32_t* __restrict__ partial_counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // carve: slabs | caches | merge area | wave counts | acc[W] | cnt[W]
  uint8_t* slab = smem + wave_id() * SLAB_BYTES;
  float* cache = reinterpret_cast<float*>(smem + WG_WAVES * SLAB_BYTES) + wave_id() * 256;
  uint64_t* merge = reinterpret_cast<uint64_t*>(smem + WG_WAVES * SLAB_BYTES + WG_WAVES * 1024);  // [WG_WAVES][128]
  int* s_counts = reinterpret_cast<int*>(smem + WG_WAVES * SLAB_BYTES + WG_WAVES * 1024 + WG_WAVES * 128 * 8);
  float* acc = reinterpret_cast<float*>(smem + WINDOW_LDS_FIXED);
  uint8_t* cnt = reinterpret_cast<uint8_t*>(acc + W);

  const int lane = lane_id();
  const int wave = wave_id();
  const int tid = (int)threadIdx.x;
  const int64_t item = blockIdx.x;
  const int q = (int)(item / items_per_query);
  const int g = (int)(item - (int64_t)q * items_per_query);
  const DevQuery Q = queries[q];
  const bool has_norms = seg.norms != nullptr;

  WaveTopK top;
  uint64_t tau = 0;
  int count = 0;
  int cur_table = -1;
  float k1 = 0.f;

  const int win0 = g * windows_per_item;
  const int win1 = Q.n_terms > 0 ? min(windows_per_query, win0 + windows_per_item) : win0;
  for (int win = win0; win < win1; ++win) {
    const int32_t w0 = win * W;
    const int32_t w1 = min(seg.max_doc, w0 + W);
    for (int i = tid * 4; i < W; i += WG_THREADS * 4) {
      *reinterpret_cast<float4*>(acc + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<uint32_t*>(cnt + i) = 0u;
    }
    __syncthreads();
    bool alive = true;  // AND: some doc of this window still matches every clause so far
    for (int ti = 0; ti < Q.n_terms && alive; ++ti) {
      const DevTerm T = terms[Q.first_term + ti];
      if (T.sim_table != cur_table) {
        load_sim_table(seg, T.sim_table, cache, lane, k1);
        cur_table = T.sim_table;
      }
      const float wk = T.weight * (k1 + 1.0f);
      bool touched = false;
      auto visit = [&](int32_t doc, uint32_t freq, bool valid) {
        valid = valid && doc >= w0 && doc < w1 && doc_is_live(seg.live, doc);
        if (valid) {
          const int o = doc - w0;
          const float nrm = has_norms ? cache[seg.norms[doc]] : k1;
          const float s = bm25_score(wk, (float)(int32_t)freq, nrm);
          if (IS_AND) {
            if (cnt[o] == (uint8_t)ti) { acc[o] += s; cnt[o] = (uint8_t)(ti + 1); touched = true; }
          } else {
            acc[o] += s;
            cnt[o] = 1;
          }
        }
      };
      if (T.df == 1) {
        if (tid == 0) visit(T.singleton_doc, (uint32_t)T.singleton_freq, true);
      } else if (T.df > 1) {
        const int blo = find_block(seg.dir_last, T.dir_base, T.nblocks, w0);
        const int bhi = find_block(seg.dir_last, T.dir_base, T.nblocks, w1 - 1);
        for (int blk = blo + wave; blk <= bhi && blk < T.nblocks; blk += WG_WAVES) {
          const int32_t base = blk == 0 ? 0 : seg.dir_last[T.dir_base + blk - 1];
          const BlockPair bp = decode_block<LEGACY>(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + blk],
                                                     seg.dir_hdr[T.dir_base + blk], slab, lane);
          int32_t d0, d1;
          deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
          visit(d0, bp.f0, true);
          visit(d1, bp.f1, true);
        }
        if (bhi == T.nblocks && T.tail_n > 0 && wave == (T.nblocks & (WG_WAVES - 1))) {
          const uint32_t toff = T.nblocks ? seg.dir_off[T.dir_base + T.nblocks] : 0u;
          const int32_t base = T.nblocks ? seg.dir_last[T.dir_base + T.nblocks - 1] : 0;
          int32_t d0, d1;
          uint32_t f0, f1;
          decode_tail(seg.doc + T.start_fp + toff, T.tail_n, base, slab, lane, d0, d1, f0, f1);
          visit(d0, f0, 2 * lane < T.tail_n);
          visit(d1, f1, 2 * lane + 1 < T.tail_n);
        }
      }
      if (IS_AND) alive = __syncthreads_or(touched ? 1 : 0) != 0;
      else __syncthreads();
    }
    if (alive) {
      const uint8_t want = IS_AND ? (uint8_t)Q.n_terms : (uint8_t)1;
      for (int i = tid; i < W; i += WG_THREADS) {
        const bool hit = cnt[i] == want;
        const uint64_t key = hit ? make_key(acc[i], w0 + i) : 0ull;
        count += __popcll(__ballot(hit));
        topk_offer<WIDE>(top, key, tau, k, lane);
      }
    }
    __syncthreads();
  }
  // fold the four wave lists into wave 0's
  if (lane < 64) { merge[wave * 128 + lane] = top.a; merge[wave * 128 + 64 + lane] = WIDE ? top.b : 0ull; }
  if (lane == 0) s_counts[wave] = count;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < WG_WAVES; ++w) {
      topk_offer<WIDE>(top, merge[w * 128 + lane], tau, k, lane);
      if (WIDE) topk_offer<WIDE>(top, merge[w * 128 + 64 + lane], tau, k, lane);
      count += s_counts[w];
    }
    uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
    if (lane < k) pk[lane] = top.a;
    if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
    if (lane == 0) partial_counts[item] = count;
  }
}

// ---- fold the per-item lists of each query: one wavefront per query --------------------------------------------
// `head_items` > 0 selects the TERM kernel's item layout: item q is query q's first chunk and the query's other
// chunks are items head_items + [item_prefix[q], item_prefix[q+1]); 0 = plain contiguous ranges.
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_merge_items(const int64_t* __restrict__ item_prefix, int n_queries, int k,
                                                            const uint64_t* __restrict__ partial_keys,
                                                            const int32_t* __restrict__ partial_counts, int32_t doc_base,
                                                            int head_items, HitOut* __restrict__ hits_out,
                                                            int64_t* __restrict__ totals_out) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  WaveTopK top;
  uint64_t tau = 0;
  int64_t total = 0;
  const int64_t i0 = item_prefix[q], i1 = item_prefix[q + 1];
  const int64_t n_mine = (i1 - i0) + (head_items > 0 ? 1 : 0);
  auto item_at = [&](int64_t j) -> int64_t {  // j-th item of this query
    if (head_items > 0) return j == 0 ? (int64_t)q : (int64_t)head_items + i0 + j - 1;
    return i0 + j;
  };
  for (int64_t g0 = 0; g0 < n_mine; g0 += 64) {
    // each lane looks at one item's best key (lists are sorted best-first): an item whose head cannot
    // enter the current top-k is skipped without reading the rest of its list
    const bool ok = g0 + lane < n_mine;
    const int64_t mine = ok ? item_at(g0 + lane) : 0;
    const uint64_t head = ok ? partial_keys[(size_t)mine * (size_t)k] : 0ull;
    total += wave_reduce_add(ok ? partial_counts[mine] : 0);
    uint64_t m = __ballot(head > tau);
    while (m) {
      const int src = __builtin_ctzll(m);
      const uint64_t* pk = partial_keys + (size_t)item_at(g0 + src) * (size_t)k;
      topk_offer<WIDE>(top, lane < k ? pk[lane] : 0ull, tau, k, lane);
      if (WIDE) topk_offer<WIDE>(top, lane + 64 < k ? pk[lane + 64] : 0ull, tau, k, lane);
      m &= m - 1;
      m &= __ballot(head > tau);
    }
  }
  HitOut* out = hits_out + (size_t)q * (size_t)k;
  if (lane < k) out[lane] = top.a ? HitOut{key_doc(top.a) + doc_base, key_score(top.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = top.b ? HitOut{key_doc(top.b) + doc_base, key_score(top.b)} : HitOut{-1, 0.f};
  if (lane == 0) totals_out[q] = total;
}

// TopDocsCollector::finish_parallel across leaves / shards: lists laid out [list][query][k] (already in global
// doc ids), one wavefront per query.
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_merge_lists(const HitOut* __restrict__ hits_in, const int64_t* __restrict__ totals_in,
                                                            int n_lists, int n_queries, int k, HitOut* __restrict__ hits_out,
                                                            int64_t* __restrict__ totals_out) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  WaveTopK top;
  uint64_t tau = 0;
  int64_t total = 0;
  for (int l = 0; l < n_lists; ++l) {
    const HitOut* in = hits_in + ((size_t)l * n_queries + q) * (size_t)k;
    for (int r = 0; r < k; r += 64) {
      uint64_t key = 0;
      if (r + lane < k) { const HitOut h = in[r + lane]; if (h.doc >= 0) key = make_key(h.score, h.doc); }
      topk_offer<WIDE>(top, key, tau, k, lane);
    }
    total += totals_in[(size_t)l * n_queries + q];
  }
  HitOut* out = hits_out + (size_t)q * (size_t)k;
  if (lane < k) out[lane] = top.a ? HitOut{key_doc(top.a), key_score(top.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = top.b ? HitOut{key_doc(top.b), key_score(top.b)} : HitOut{-1, 0.f};
  if (lane == 0) totals_out[q] = total;
}

// group-local result rows -> the caller's rows (queries are partitioned by op on the host)
__global__ void k_scatter_rows(const HitOut* __restrict__ hits, const int64_t* __restrict__ totals, const int32_t* __restrict__ qmap,
                               int k, HitOut* __restrict__ hits_out, int64_t* __restrict__ totals_out) {
  const int q = (int)blockIdx.x;
  const int dst = qmap[q];
  for (int i = (int)threadIdx.x; i < k; i += (int)blockDim.x) hits_out[(size_t)dst * k + i] = hits[(size_t)q * k + i];
  if (threadIdx.x == 0) totals_out[dst] = totals[q];
}

__global__ void k_init_hits(HitOut* __restrict__ hits, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hits[i] = HitOut{-1, 0.f};
}

}  // namespace rgpu
