// Disjunction (OR) on the GPU in two launches, nothing shared between wavefronts inside either:
//   1. k_score_terms   the SPARSE clauses' postings (and the VInt tails of the dense ones) are decoded and BM25-scored
//                      exactly once (the TermScorer work of each sub-scorer, term_scorer.rs:43-67) into a {doc, score}
//                      run per clause in HBM;
//   2. k_or_windows    each wavefront owns a range of doc-id windows. Per window it goes through the clauses *in clause
//                      order* — the summation order of SubScorers::score_sum over a SimpleQueue
//                      (search/scorer/disjunction_scorer.rs:213-225) — and adds each one's scores into an LDS
//                      accumulator: a sparse clause's from its run (walked from a cursor), a DENSE clause's (up to
//                      OR_DENSE_MAX per query: the lists that hold ~90 % of a Zipfian query's postings) straight from
//                      the block store — its FullBlocks overlapping the window are unpacked, scored through the
//                      clause's LDS score table and added, nothing materialised. Then the window is scanned: every
//                      touched doc is one collected hit (bulk_scorer.rs:114-120) offered to the wave's top-k.
// Round 1 sent every clause through a run: 8 B per posting written and read back (22 GB per 1024-query batch against
// 6 GB of postings) and 2.6 VALU instructions per posting; the dense path costs 0.6.
// This is DisjunctionSumScorer's doc-at-a-time merge (disjunction_scorer.rs:24-104, util/disi.rs) turned
// term-at-a-time per window; for >= 10 clauses the reference sums in heap order, so only 1e-5 relative holds
// there. No block is decoded twice and no posting is scored twice, whatever the clause density.
#pragma once
#include "search_and.hpp"

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
  const int t = upper_slot_wave(item_prefix, n_terms, item, lane);
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

  // a clause whose FullBlocks the window kernel decodes itself (TERM_FLAG_OR_DENSE) only sends its VInt tail through
  // a run: one item, no blocks, the tail at the run's start
  const bool dense = (T.flags & TERM_FLAG_OR_DENSE) != 0u;
  const int b0 = dense ? T.nblocks : chunk * blocks_per_item;
  const int b1 = dense ? T.nblocks : min(T.nblocks, b0 + blocks_per_item);
  const int64_t run_len = dense ? (int64_t)T.tail_n : (int64_t)T.df;
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
    run[run_len + lane] = ScoredPosting{0x7fffffff, 0.0f};
    if (T.df == 1) {
      const bool v0 = lane == 0;
      const uint32_t nb0 = (has_norms && v0) ? norm_at(seg, T.singleton_doc) : 0u;
      emit(T.singleton_doc, 0, (uint32_t)T.singleton_freq, 1u, nb0, 0u, v0, false, 0);
    } else if (T.tail_n > 0) {
      int32_t d0, d1;
      uint32_t f0, f1;
      tail_load(term_rows, seg.dir_row[T.dir_base + T.nblocks], lane, d0, d1, f0, f1);  // decoded and validated at prepare time
      const bool v0 = 2 * lane < T.tail_n, v1 = 2 * lane + 1 < T.tail_n;
      const uint32_t nb0 = (has_norms && v0) ? norm_at(seg, d0) : 0u, nb1 = (has_norms && v1) ? norm_at(seg, d1) : 0u;
      emit(d0, d1, f0, f1, nb0, nb1, v0, v1, (dense ? 0 : 128 * (int64_t)T.nblocks) + 2 * lane);
    }
  }
}

constexpr int OR_MAX_TERMS = 64;  // one cursor per clause position (SHOULD + MUST_NOT) in a lane
constexpr int OR_RUN_PAD = 64;  // sentinel entries {doc = INT_MAX} after every clause's run (written by k_score_terms)
constexpr int OR_DENSE_MAX = 4;  // clauses per query decoded inside the window kernel (score tables: one per wave of a workgroup)
// The window walk is a chain of dependent steps per clause (LDS read-modify-write of the accumulator, cursor hand-over),
// so the kernel lives on occupancy: eight wavefronts per workgroup share the dense clauses' score tables (12 KB), and
// the per-wave state is kept under 80 VGPRs — a first version that held directory windows and prefetched block heads in
// registers (135 VGPRs, 3 waves/SIMD) ran 2.2x slower than the run-only kernel it replaced.
constexpr int OR_PREFETCH = 8;  // run heads requested at the start of a window (two VGPRs each)
constexpr int OR_WAVES = 8;
constexpr int OR_THREADS = 64 * OR_WAVES;
constexpr uint32_t OR_UNTOUCHED = 0xffffffffu;  // accumulator patterns no sum of scores produces (negative quiet NaNs)
constexpr uint32_t OR_EXCLUDED = 0xfffffffeu;
static_assert(OR_DENSE_MAX <= OR_WAVES, "wave w of a workgroup builds dense clause w's score table");

__host__ __device__ constexpr size_t or_wave_lds_bytes(int W, bool msm) { return (size_t)(2 * SLAB_STREAM) + (size_t)W * (msm ? 5 : 4); }
__host__ __device__ constexpr size_t or_lds_bytes(int W, bool msm) {
  return (size_t)OR_DENSE_MAX * WAVE_CACHE_FLOATS * 4 + (size_t)OR_WAVES * or_wave_lds_bytes(W, msm);
}

// items = (query, group of `windows_per_item` windows of `W` docs), one per wavefront; items_per_query is a multiple
// of OR_WAVES, so the wavefronts of a workgroup always work on the same query and share its dense clauses' score
// tables. DevQuery::op carries the query's dense-clause mask in bits 16.. (bit i = SHOULD clause i).
// HAS_NOT: some query of the launch carries MUST_NOT clauses; HAS_MSM: some query asks for min_should_match > 1
// (disjunction_scorer.rs:317-329: a doc is a hit only if that many SHOULD clauses hold it — a per-doc clause counter
// next to the accumulator). Separate instantiations keep the common kernel lean.
template <bool LEGACY, bool WIDE, bool HAS_NOT, bool HAS_MSM>
__global__ __launch_bounds__(OR_THREADS, 6) void k_or_windows(SegView seg, const DevQuery* __restrict__ queries,
                                                              const DevTerm* __restrict__ terms,
                                                              const int64_t* __restrict__ run_prefix,
                                                              const ScoredPosting* __restrict__ runs, int n_queries,
                                                              int windows_per_query, int windows_per_item,
                                                              int items_per_query, int W, int k,
                                                              uint64_t* __restrict__ partial_keys,
                                                              int32_t* __restrict__ partial_counts,
                                                              unsigned long long* __restrict__ tau_slots,
                                                              const unsigned long long* __restrict__ ceil_slots = nullptr,
                                                              const int32_t* __restrict__ qmap = nullptr) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = wave_id();
  // LDS: [OR_DENSE_MAX score tables (norm cache 64 f32 + 64 x 11 scores)] then per wave [block staging slab | acc[W] f32
  // | cnt[W] u8 (HAS_MSM)]. An untouched accumulator holds OR_UNTOUCHED, a NaN pattern no sum of scores produces; the
  // scan at the end of a window puts it back, so there is no per-doc flag array and no clearing pass.
  float* tables = reinterpret_cast<float*>(smem);
  uint8_t* slice = smem + (size_t)OR_DENSE_MAX * WAVE_CACHE_FLOATS * 4 + (size_t)wave * or_wave_lds_bytes(W, HAS_MSM);
  uint8_t* slab = slice;
  float* acc = reinterpret_cast<float*>(slice + 2 * SLAB_STREAM);
  uint8_t* cnt = reinterpret_cast<uint8_t*>(acc + W);  // HAS_MSM only: SHOULD clauses that hold the doc
  // Workgroup b works on query b % n_queries: a query's workgroups are spread over the whole launch instead of running
  // side by side, so all but the first one or two start from the thresholds the earlier ones published (SharedTau) —
  // query-major order made every wavefront build its own top-k from nothing, ~460 insertions each, which was over half
  // of the kernel's time. The grid is exactly n_queries * items_per_query / OR_WAVES workgroups.
  const int q = (int)(blockIdx.x % (unsigned)n_queries);
  const int g = (int)(blockIdx.x / (unsigned)n_queries) * OR_WAVES + wave;
  const int64_t item = (int64_t)q * items_per_query + g;
  const DevQuery Q = queries[q];
  const bool has_live = seg.live != nullptr;
  const uint32_t dense_mask = ((uint32_t)Q.op >> 16) & 0xffffu;
  const int nd = __popc(dense_mask);
  auto nth_bit = [](uint32_t m, int n) -> int {  // index of the n-th set bit (n < popcount)
    for (int i = 0; i < n; ++i) m &= m - 1;
    return (int)__builtin_ctz(m);
  };
  // wave w builds dense clause w's score table (the same f32 expression as bm25_score per entry: bit-identical)
  if (wave < nd) {
    const DevTerm T = terms[Q.first_term + nth_bit(dense_mask, wave)];
    float* tbl = tables + wave * WAVE_CACHE_FLOATS;
    float k1;
    load_sim_table(seg, T.sim_table, tbl, lane, k1);
    build_score_table(tbl, T.weight * (k1 + 1.0f), lane);
  }
  __syncthreads();

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int hits_lane = 0;  // collected docs this lane saw (summed over the wave at the end: TopDocs::total_hits)
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);
  const uint64_t ceil = ceil_slots != nullptr ? ceil_slots[qmap[q]] : ~0ull;  // k > 128: this pass's hits stay below it (wave.hpp)
  const int win0 = g * windows_per_item;
  const int win1 = Q.n_terms > 0 ? min(windows_per_query, win0 + windows_per_item) : win0;
  const int32_t first_doc = win0 * W;

  // ---- runs: lane t owns one clause's cursor: run base, length, and the first entry with doc >= this item's first
  // window — one lane-parallel binary search over all clauses at once. The n_not MUST_NOT clauses (stored after the
  // n_terms SHOULD clauses) take lanes 0 .. n_not-1 so that a window meets them first: ReqNotScorer over the disjunction
  // (boolean_query.rs:271-273, req_not_scorer.rs:47-63) — their docs are marked excluded before anything is summed.
  const int n_not = HAS_NOT ? Q.pad : 0;
  const int msm = HAS_MSM ? ((Q.op >> 8) & 0xff) : 1;
  const int n_pos = Q.n_terms + n_not;  // clause positions of a window, in summation order
  const bool mine = lane < n_pos;
  const int my_clause = lane < n_not ? Q.n_terms + lane : lane - n_not;
  const bool my_dense = mine && lane >= n_not && my_clause < 16 && ((dense_mask >> my_clause) & 1u);  // (a dense clause is one of the first 16)
  const int64_t my_base = mine ? run_prefix[Q.first_term + my_clause] : 0;
  int my_len = 0;
  if (mine) { const DevTerm* Tm = terms + Q.first_term + my_clause; my_len = my_dense ? Tm->tail_n : Tm->df; }
  int64_t my_at = my_base;  // absolute index of the entry under this clause's cursor (every run ends in sentinels)
  {
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

  // ---- dense clauses: lane s (< nd) holds slot s's term fields and its cursor (the first block that may still hold a doc
  // of the current window); directory entries are read through the scalar cache (wave-uniform addresses)
  uint64_t d_bs = 0, d_pn = 0;
  uint32_t d_dir = 0;
  int32_t d_nb = 0, d_cb = 0;
  float d_wk = 0.f;
  if (lane < nd) {
    const DevTerm* Td = terms + Q.first_term + nth_bit(dense_mask, lane);
    d_bs = Td->bs_base; d_pn = Td->pn_base; d_dir = Td->dir_base; d_nb = Td->nblocks;
    d_wk = Td->weight * (seg.sim_tables[(size_t)Td->sim_table * 257 + 256] + 1.0f);
  }
  for (int s = 0; s < nd; ++s) {
    const int cb = find_block_wave(seg.dir_last, (uint32_t)readlane((int)d_dir, s), 0, readlane(d_nb, s), first_doc, lane);
    d_cb = lane == s ? cb : d_cb;
  }

  for (int i = lane; i < W; i += 64) acc[i] = __uint_as_float(OR_UNTOUCHED);
  wave_sync();
  for (int win = win0; win < win1; ++win) {
    const int32_t w0 = win * W;
    const int32_t w1 = min(seg.max_doc, w0 + W);
    const uint32_t wlen = (uint32_t)(w1 - w0);
    // ---- loads first: the first 64 run entries of EVERY clause with a posting in this window (one exposed load latency
    // per window instead of one per clause: the window walk is latency bound), and the first block of every dense clause
    const uint64_t active0 = __ballot(my_next < w1);
    ScoredPosting pre[OR_PREFETCH];  // clause positions beyond OR_PREFETCH load when their turn comes
#pragma unroll
    for (int t = 0; t < OR_PREFETCH; ++t) {
      pre[t] = ScoredPosting{0x7fffffff, 0.f};
      if ((active0 >> t) & 1ull) pre[t] = runs[(int64_t)readlane64((uint64_t)my_at, t) + lane];  // wave-uniform branch
    }
    bool touched_any = false;  // wave-uniform: some accumulator of this window was written

    // one posting into the window's accumulator (all docs of one clause are distinct: no two lanes meet)
    auto add = [&](int32_t doc, float sc, bool valid, bool prohibited) {
      const uint32_t o = (uint32_t)(doc - w0);
      bool in = valid && o < wlen;
      if (in && has_live) in = doc_is_live(seg.live, doc);  // doc < w1 <= max_doc here
      if (in) {
        const float a = acc[o];
        const uint32_t ab = __float_as_uint(a);
        const bool first = ab == OR_UNTOUCHED;
        if (prohibited) {  // wave-uniform: a prohibited clause only marks
          if (first) acc[o] = __uint_as_float(OR_EXCLUDED);
        } else if (!HAS_NOT || ab != OR_EXCLUDED) {
          acc[o] = (first ? 0.0f : a) + sc;  // 0.0f + s: the reference's `score = 0; score += s`
          if (HAS_MSM) cnt[o] = first ? (uint8_t)1 : (uint8_t)(cnt[o] + 1);
        }
      }
    };

    for (int t = 0; t < n_pos; ++t) {
      // ---- the clause's run (a sparse clause's postings, a dense clause's VInt tail)
      if ((active0 >> t) & 1ull) {
        ScoredPosting e;
        switch (t) {  // register select behind scalar branches: the loop body exists once, not once per clause position
          case 0: e = pre[0]; break; case 1: e = pre[1]; break; case 2: e = pre[2]; break; case 3: e = pre[3]; break;
          case 4: e = pre[4]; break; case 5: e = pre[5]; break; case 6: e = pre[6]; break; case 7: e = pre[7]; break;
          default: e = runs[(int64_t)readlane64((uint64_t)my_at, t) + lane]; break;
        }
        int taken = 0;
        int32_t next;
        while (true) {
          const bool in = e.doc < w1;
          const int n = __popcll(__ballot(in));  // runs are doc-sorted: the in-window entries are a prefix
          add(e.doc, e.score, in, t < n_not);
          taken += n;
          if (n < 64) { next = readlane(e.doc, n); break; }  // the entry now under the cursor (or the sentinel)
          e = runs[(int64_t)readlane64((uint64_t)my_at, t) + taken + lane];  // a stretch longer than 64: the rare case
        }
        if (lane == t) { my_at += taken; my_next = next; }
        touched_any = true;
        wave_sync();
      }
      // ---- the clause's FullBlocks, when it is a dense one
      const int ci = t - n_not;
      if (ci >= 0 && ci < 16 && ((dense_mask >> ci) & 1u)) {
        const int s = __popc(dense_mask & ((1u << ci) - 1u));
        int cb = readlane(d_cb, s);
        const int nb = readlane(d_nb, s);
        const uint32_t dir = (uint32_t)readlane((int)d_dir, s);
        const uint8_t* term_rows = seg.bstore + readlane64(d_bs, s);
        const uint8_t* pn = seg.pnorm + readlane64(d_pn, s);
        const float* tbl = tables + s * WAVE_CACHE_FLOATS;
        const float wk = __int_as_float(readlane(__float_as_int(d_wk), s));
        while (cb < nb) {
          const int32_t base = cb == 0 ? 0 : seg.dir_last[dir + cb - 1];
          if (base >= w1 - 1) break;  // the block's docs lie beyond this window
          const int32_t last = seg.dir_last[dir + cb];
          const uint32_t hdr = seg.dir_hdr[dir + cb];
          const uint4 rows = block_rows_load(block_rows_at(term_rows, seg.dir_row[dir + cb]), hdr, lane);
          const uint32_t nn = *reinterpret_cast<const uint16_t*>(pn + (128u * (uint32_t)cb + 2u * (uint32_t)lane));
          stage_rows(rows, slab, lane);
          wave_sync();
          uint32_t x0, x1, f0, f1;
          staged_doc_deltas<LEGACY>(slab, rows, hdr, lane, x0, x1);
          staged_freqs<LEGACY>(slab, rows, hdr, lane, f0, f1);
          wave_sync();  // slab is free for the next block
          int32_t e0, e1;
          deltas_to_docs(x0, x1, base, e0, e1);
          const uint32_t nb0 = nn & 0xffu, nb1 = nn >> 8;
          float s0, s1;
          if (hdr_bfreq(hdr) <= 3 || !__ballot((f0 > f1 ? f0 : f1) > (uint32_t)SCORE_TABLE_FREQS)) {
            s0 = table_score(tbl, nb0, f0);
            s1 = table_score(tbl, nb1, f1);
          } else {  // a freq beyond the table's columns: the formula the table memoises (tbl[0..63] = the norm cache by rank)
            s0 = bm25_score(wk, (float)(int32_t)f0, tbl[nb0]);
            s1 = bm25_score(wk, (float)(int32_t)f1, tbl[nb1]);
          }
          add(e0, s0, true, false);
          add(e1, s1, true, false);
          touched_any = true;
          wave_sync();
          if (last >= w1) break;  // the block reaches into the next window: the cursor stays on it
          ++cb;
        }
        d_cb = lane == s ? cb : d_cb;
      }
    }

    // ---- scan the window: every touched doc that no prohibited clause claimed (and that enough SHOULD clauses hold)
    // is one collected hit; four docs per lane per step (one ds_read_b128), accumulators go back to "untouched"
#ifndef RGPU_OR_ABL  // developer ablations (variant builds only; results are wrong)
#define RGPU_OR_ABL 0
#endif
    if (touched_any && RGPU_OR_ABL != 2) {
      for (uint32_t i0 = 0; i0 < wlen; i0 += 256) {  // uniform trip count: the offer is a wave-wide operation
        float4* cell = reinterpret_cast<float4*>(acc + i0 + 4 * lane);
        const float4 v = *cell;
        const uint32_t r0 = __float_as_uint(v.x), r1 = __float_as_uint(v.y), r2 = __float_as_uint(v.z), r3 = __float_as_uint(v.w);
        if (__ballot((r0 & r1 & r2 & r3) != OR_UNTOUCHED)) {
          // >= OR_EXCLUDED: untouched or excluded — no hit
          bool h0 = r0 < OR_EXCLUDED, h1 = r1 < OR_EXCLUDED, h2 = r2 < OR_EXCLUDED, h3 = r3 < OR_EXCLUDED;
          if (HAS_MSM) {
            const uint32_t c4 = *reinterpret_cast<const uint32_t*>(cnt + i0 + 4 * lane);
            h0 = h0 && (int)(c4 & 0xffu) >= msm;
            h1 = h1 && (int)((c4 >> 8) & 0xffu) >= msm;
            h2 = h2 && (int)((c4 >> 16) & 0xffu) >= msm;
            h3 = h3 && (int)(c4 >> 24) >= msm;
          }
          hits_lane += (int)h0 + (int)h1 + (int)h2 + (int)h3;
          *cell = make_float4(__uint_as_float(OR_UNTOUCHED), __uint_as_float(OR_UNTOUCHED), __uint_as_float(OR_UNTOUCHED), __uint_as_float(OR_UNTOUCHED));
          // candidates: score order bits against the entry threshold's before any key is built
          const uint32_t thi = (uint32_t)(tau >> 32);
          const uint32_t o0 = float_order_bits(v.x), o1 = float_order_bits(v.y), o2 = float_order_bits(v.z), o3 = float_order_bits(v.w);
          const bool c0 = h0 && o0 >= thi, c1 = h1 && o1 >= thi, c2 = h2 && o2 >= thi, c3 = h3 && o3 >= thi;
          if (RGPU_OR_ABL != 1 && __ballot(c0 || c1 || c2 || c3)) {
            const int32_t d = w0 + (int32_t)i0 + 4 * lane;
            uint64_t key = c0 ? below(make_key(v.x, d), ceil) : 0ull;
            if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
            key = c1 ? below(make_key(v.y, d + 1), ceil) : 0ull;
            if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
            key = c2 ? below(make_key(v.z, d + 2), ceil) : 0ull;
            if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
            key = c3 ? below(make_key(v.w, d + 3), ceil) : 0ull;
            if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          }
        }
      }
      wave_sync();
    }
  }
  shared.publish<WIDE>(top, k, lane);
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  const int count = wave_reduce_add(hits_lane);
  if (lane == 0) partial_counts[item] = count;
}

}  // namespace rgpu
