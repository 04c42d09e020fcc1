// Exact PhraseQuery (slop 0) on the GPU, three launches:
//   1. k_search_and in "emit" mode    the conjunction of the phrase's terms (PhraseWeight::create_scorer drives an
//                                     ExactPhraseScorer through a ConjunctionScorer over the terms' postings,
//                                     query/phrase_query.rs:262-330, scorer/phrase_scorer.rs:131-160): every doc that
//                                     holds all terms is appended to the query's candidate list;
//   2. k_phrase_match                 one wavefront per candidate: for every term find the doc's posting (block
//                                     directory -> block -> index, as BlockPostingIterator::advance does,
//                                     posting_reader.rs:1439-1587), turn "positions buffered before this block" + the
//                                     freqs of the block's earlier docs into the doc's place in the term's position
//                                     stream (skip_positions, :1326-1350), unpack its `freq` position deltas from the
//                                     .pos blocks (refill_positions, :1285-1324: 128-value ForUtil blocks, a trailing
//                                     VInt block) and prefix-sum them into positions; ExactPhraseScorer::phrase_freq
//                                     (phrase_scorer.rs:179-229) is then the number of positions p of the first term
//                                     with p - offset_0 + offset_i present in term i's list for every i — an
//                                     intersection of sorted lists of (position - phrase offset); the score is
//                                     BM25(phrase freq, norm) with the phrase's summed-idf weight (:246-251);
//   3. k_phrase_collect               TopDocsCollector over the candidates with phrase freq > 0.
// Fields with payloads or offsets (a third file, .pay) are refused at upload; sloppy phrases (slop > 0) are not served.
#pragma once
#include "search_and.hpp"

namespace rgpu {

constexpr int RGPU_MAX_K_DEV = 128;  // = RGPU_MAX_K: the widest list a wavefront's registers hold
constexpr int PHRASE_LIST_CAP = 1024;  // positions of one term inside one doc that the LDS lists hold
constexpr int32_t PHRASE_DEAD = (int32_t)0x80000000;

template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_match(SegView seg, const DevQuery* __restrict__ queries,
                                                             const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                             const int64_t* __restrict__ emit_prefix,
                                                             const unsigned long long* __restrict__ emit_count,
                                                             const int32_t* __restrict__ emit_docs, int n_queries, int64_t n_slots,
                                                             int64_t pos_len, uint64_t* __restrict__ keys_out, int* err) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ int32_t lists_a[WG_WAVES][PHRASE_LIST_CAP];
  __shared__ int32_t lists_c[WG_WAVES][PHRASE_LIST_CAP];
  __shared__ float caches[WG_WAVES][256];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t slot = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (slot >= n_slots) return;
  const int q = upper_slot(emit_prefix, n_queries, slot);
  const int64_t idx = slot - emit_prefix[q];
  if ((unsigned long long)idx >= emit_count[q]) {  // the conjunction produced fewer matches than the lead term has docs
    if (lane == 0) keys_out[slot] = 0ull;
    return;
  }
  const int32_t doc = emit_docs[slot];
  const DevQuery Q = queries[q];
  uint8_t* slab = slabs[wave];
  int32_t* A = lists_a[wave];
  int32_t* C = lists_c[wave];
  auto give_up = [&](int code) {
    if (lane == 0) { atomicMin(err, code); keys_out[slot] = 0ull; }
  };
  int n_a = 0;
  for (int c = 0; c < Q.n_terms; ++c) {
    const DevTerm T = terms[Q.first_term + c];
    const PosTerm P = pterms[Q.first_term + c];
    // ---- 1. the doc's posting in this term: where its positions start in the term's position stream
    int64_t fp = (int64_t)P.pos_start_fp;
    int skip = 0, freq = 0;
    if (T.df == 1) {
      freq = T.singleton_freq;
    } else {
      const int blk = find_block(seg.dir_last, T.dir_base, T.nblocks, doc);
      int32_t e0, e1;
      uint32_t g0, g1;
      bool v0 = true, v1 = true;
      if (blk < T.nblocks) {
        const int32_t base = blk == 0 ? 0 : seg.dir_last[T.dir_base + blk - 1];
        const BlockPair bp = decode_block<LEGACY>(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + blk], seg.dir_hdr[T.dir_base + blk], slab, lane);
        deltas_to_docs(bp.d0, bp.d1, base, e0, e1);
        g0 = bp.f0; g1 = bp.f1;
      } else {
        tail_load(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + T.nblocks], lane, e0, e1, g0, g1);
        v0 = 2 * lane < T.tail_n; v1 = 2 * lane + 1 < T.tail_n;
      }
      const uint64_t m0 = __ballot(v0 && e0 == doc), m1 = __ballot(v1 && e1 == doc);
      if (!(m0 | m1)) { give_up(-1); return; }  // the conjunction said the doc is here
      const int pair = (int)((v0 ? g0 : 0u) + (v1 ? g1 : 0u));
      const int excl = wave_incl_scan(pair) - pair;  // freqs of the block's docs in the lanes before this one
      int before;
      if (m0) {
        const int src = (int)__builtin_ctzll(m0);
        before = readlane(excl, src);
        freq = readlane((int)g0, src);
      } else {
        const int src = (int)__builtin_ctzll(m1);
        before = readlane(excl, src) + readlane((int)g0, src);
        freq = readlane((int)g1, src);
      }
      const uint64_t st = seg.dir_pos[T.dir_base + blk];
      fp += (int64_t)(uint32_t)st;
      skip = (int)(st >> 32) + before;
    }
    if (freq <= 0 || freq > PHRASE_LIST_CAP) { give_up(freq <= 0 ? -4 : -5); return; }
    // ---- 2. whole position blocks that hold only earlier docs' positions (ForUtil::skip_block, for_util.rs:263-272)
    while (skip >= 128) {
      if (fp == P.last_pos_block_fp || fp + 2 > pos_len) { give_up(-4); return; }
      const uint32_t b = seg.pos[fp];
      if (b > 32u) { give_up(-4); return; }
      int vlen = 0;
      if (b == 0) (void)read_vint_uniform(seg.pos + fp + 1, &vlen);
      fp += 1 + (b ? 16 * (int64_t)b : (int64_t)vlen);
      skip -= 128;
    }
    // ---- 3. this doc's `freq` positions: deltas from the stream, a running sum from 0 (posting_reader.rs:1357-1380)
    int32_t* L = c == 0 ? A : C;
    int got = 0;
    int32_t carry = 0;
    while (got < freq) {
      uint32_t x0, x1;
      int nvals = 128;
      if (fp < 0 || fp + 2 > pos_len) { give_up(-4); return; }
      if (fp == P.last_pos_block_fp) {
        decode_vint_block(seg.pos + fp, slab, lane, x0, x1);
        nvals = (int)(P.total_term_freq % 128);
        fp = -2;  // nothing follows the trailing block
      } else {
        const uint32_t b = seg.pos[fp];
        if (b > 32u) { give_up(-4); return; }
        if (b == 0) {
          int vlen;
          x0 = x1 = read_vint_uniform(seg.pos + fp + 1, &vlen);
          fp += 1 + vlen;
        } else {
          if (lane < 32) *reinterpret_cast<uint4*>(slab + 16 * lane) = load16_unaligned(seg.pos + fp + 1 + 16 * lane);
          wave_sync();
          extract_pair<LEGACY>(slab, (int)b, lane, x0, x1);
          wave_sync();
          fp += 1 + 16 * (int64_t)b;
        }
      }
      const int take = min(nvals - skip, freq - got);
      if (take <= 0) { give_up(-4); return; }  // the stream ends before the doc's positions do
      const int i0 = 2 * lane, i1 = 2 * lane + 1;
      const bool in0 = i0 >= skip && i0 < skip + take, in1 = i1 >= skip && i1 < skip + take;
      const int d0 = in0 ? (int)x0 : 0, d1 = in1 ? (int)x1 : 0;
      const int pr = d0 + d1;
      const int incl = wave_incl_scan(pr);
      const int32_t p0 = carry + incl - pr + d0, p1 = carry + incl;
      if (in0) L[got + i0 - skip] = p0 - P.phrase_pos;
      if (in1) L[got + i1 - skip] = p1 - P.phrase_pos;
      carry += readlane(incl, 63);
      got += take;
      skip = 0;
    }
    wave_sync();
    // ---- 4. keep the first term's positions that line up with this term's
    if (c == 0) {
      n_a = freq;
    } else {
      for (int i = lane; i < n_a; i += 64) {
        const int32_t a = A[i];
        if (a != PHRASE_DEAD) {
          int lo = 0, hi = freq;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (C[mid] < a) lo = mid + 1; else hi = mid;
          }
          if (lo >= freq || C[lo] != a) A[i] = PHRASE_DEAD;
        }
      }
      wave_sync();
    }
  }
  int alive = 0;
  for (int i = lane; i < n_a; i += 64) alive += A[i] != PHRASE_DEAD ? 1 : 0;
  const int phrase_freq = wave_reduce_add(alive);
  uint64_t key = 0ull;
  if (phrase_freq > 0) {
    const DevTerm T0 = terms[Q.first_term];
    float k1;
    load_sim_table(seg, T0.sim_table, caches[wave], lane, k1);
    const float wk = T0.weight * (k1 + 1.0f);
    const float nrm = seg.norms != nullptr ? caches[wave][seg.norms[doc]] : k1;
    key = make_key(bm25_score(wk, (float)phrase_freq, nrm), doc);
  }
  if (lane == 0) keys_out[slot] = key;
}

// TopDocsCollector over one query's candidates: a key of 0 = "phrase freq 0" (not a hit). One wavefront per query.
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_collect(const int64_t* __restrict__ emit_prefix,
                                                               const unsigned long long* __restrict__ emit_count,
                                                               const uint64_t* __restrict__ keys, int n_queries, int k, int32_t doc_base,
                                                               HitOut* __restrict__ hits_out, int64_t* __restrict__ totals_out) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  WaveTopK top;
  uint64_t tau = 0;
  int64_t total = 0;
  const int64_t base = emit_prefix[q], n = (int64_t)emit_count[q];
  for (int64_t i0 = 0; i0 < n; i0 += 64) {
    const uint64_t key = i0 + lane < n ? keys[base + i0 + lane] : 0ull;
    total += __popcll(__ballot(key != 0ull));
    if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane);
  }
  HitOut* out = hits_out + (size_t)q * (size_t)k;
  if (lane < k) out[lane] = top.a ? HitOut{key_doc(top.a) + doc_base, key_score(top.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = top.b ? HitOut{key_doc(top.b) + doc_base, key_score(top.b)} : HitOut{-1, 0.f};
  if (lane == 0) totals_out[q] = total;
}

// ---- QueryRescorer (search/scorer/rescorer.rs:129-374) with a batched second-pass scorer (the BatchScorer hook, :32-36) ----
// One wavefront per first-pass hit: the rescore query's scorer is "advanced" to the hit's doc in every clause (block
// directory -> block -> index: BlockDocIterator::advance) and, where the query matches the doc, its score is combined
// with the first-pass score: combine_score(:337-352) = mode.combine(first * query_weight, second * rescore_weight),
// first * query_weight when the query does not match. TERM / all-MUST / all-SHOULD term queries; clause sums in the
// order their scorers use (conjunction: cost order, the host sorts; disjunction: clause order).
struct RescoreParams {
  float query_weight, rescore_weight;
  int32_t mode;    // RescoreMode: 0 Avg, 1 Max, 2 Min, 3 Total, 4 Multiply (rescorer.rs:97-116)
  int32_t window;  // hits of the row that are rescored
};

__device__ __forceinline__ float rescore_combine(int mode, float primary, float secondary) {
  switch (mode) {
    case 0: return (primary + secondary) / 2.0f;
    case 1: return fmaxf(primary, secondary);  // f32::max
    case 2: return fminf(primary, secondary);
    case 3: return primary + secondary;
    default: return primary * secondary;
  }
}

template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_rescore(SegView seg, const DevQuery* __restrict__ queries, const DevTerm* __restrict__ terms,
                                                        const RescoreParams* __restrict__ params, int n_queries, int k,
                                                        HitOut* __restrict__ hits, int finish) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][2 * SLAB_STREAM];
  __shared__ float caches[WG_WAVES][256];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t slot = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (slot >= (int64_t)n_queries * k) return;
  const int q = (int)(slot / k), i = (int)(slot - (int64_t)q * k);
  const RescoreParams R = params[q];
  HitOut* hit = hits + slot;
  const HitOut h = *hit;
  if (h.doc < 0) return;
  const float first = h.score * R.query_weight;
  if (i >= R.window) {  // combine_docs (:356-374): hits past the window only take the query weight — once, in the finishing call
    if (lane == 0 && finish) hit->score = first;
    return;
  }
  const int32_t doc = h.doc - seg.doc_base;
  if (doc < 0 || doc >= seg.max_doc) return;  // another leaf's doc: that leaf's call handles it
  const DevQuery Q = queries[q];
  const bool conj = (Q.op & 0xff) != 2;  // TERM / AND: every clause must hold the doc; OR: any
  uint8_t* slab = slabs[wave];
  float sum = 0.0f;
  bool any = false, all = Q.n_terms > 0;
  int cur_table = -1;
  float k1 = 0.f;
  for (int c = 0; c < Q.n_terms; ++c) {
    const DevTerm T = terms[Q.first_term + c];
    uint32_t freq = 0;
    if (T.df == 1) {
      if (T.singleton_doc == doc) freq = (uint32_t)T.singleton_freq;
    } else {
      const int blk = find_block(seg.dir_last, T.dir_base, T.nblocks, doc);
      int32_t e0, e1;
      uint32_t g0, g1;
      bool v0 = true, v1 = true;
      bool have = true;
      if (blk < T.nblocks) {
        const int32_t base = blk == 0 ? 0 : seg.dir_last[T.dir_base + blk - 1];
        const BlockPair bp = decode_block<LEGACY>(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + blk], seg.dir_hdr[T.dir_base + blk], slab, lane);
        deltas_to_docs(bp.d0, bp.d1, base, e0, e1);
        g0 = bp.f0; g1 = bp.f1;
      } else if (T.tail_n > 0) {
        tail_load(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + T.nblocks], lane, e0, e1, g0, g1);
        v0 = 2 * lane < T.tail_n; v1 = 2 * lane + 1 < T.tail_n;
      } else {
        have = false; e0 = e1 = 0; g0 = g1 = 0u; v0 = v1 = false;
      }
      const uint64_t m0 = __ballot(have && v0 && e0 == doc), m1 = __ballot(have && v1 && e1 == doc);
      if (m0) freq = (uint32_t)readlane((int)g0, (int)__builtin_ctzll(m0));
      else if (m1) freq = (uint32_t)readlane((int)g1, (int)__builtin_ctzll(m1));
    }
    if (freq == 0u) { all = false; if (conj) break; continue; }
    any = true;
    if (T.sim_table != cur_table) { load_sim_table(seg, T.sim_table, caches[wave], lane, k1); cur_table = T.sim_table; }
    const float nrm = seg.norms != nullptr ? caches[wave][seg.norms[doc]] : k1;
    sum += bm25_score(T.weight * (k1 + 1.0f), (float)(int32_t)freq, nrm);  // 0.0f + s for the first clause
  }
  const bool match = conj ? all : any;
  if (lane == 0) hit->score = match ? rescore_combine(R.mode, first, sum * R.rescore_weight) : first;
}

// hits.sort() over the rescored window (rescorer.rs:330: score desc, then doc asc — ScoreDocHit's order,
// sort_field/collapse_top_docs.rs:186-202), written back to the top of the row: one wavefront per query.
__global__ __launch_bounds__(WG_THREADS) void k_rescore_sort(const RescoreParams* __restrict__ params, int n_queries, int k, HitOut* __restrict__ hits) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  HitOut* row = hits + (size_t)q * (size_t)k;
  const int window = min(params[q].window, k);
  WaveTopK top;
  uint64_t tau = 0;
  int n = 0;
  for (int r = 0; r < window; r += 64) {
    uint64_t key = 0;
    if (r + lane < window) { const HitOut h = row[r + lane]; if (h.doc >= 0) key = make_key(h.score, h.doc); }
    n += __popcll(__ballot(key != 0ull));
    topk_offer<true>(top, key, tau, RGPU_MAX_K_DEV, lane);
  }
  if (lane < n) row[lane] = HitOut{key_doc(top.a), key_score(top.a)};
  if (lane + 64 < n) row[lane + 64] = HitOut{key_doc(top.b), key_score(top.b)};
}

}  // namespace rgpu
