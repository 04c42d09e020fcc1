// PhraseQuery on the GPU. Exact phrases (slop 0), three launches:
//   1. k_search_and in "emit" mode    the conjunction of the phrase's terms (PhraseWeight::create_scorer drives an
//                                     ExactPhraseScorer through a ConjunctionScorer over the terms' postings,
//                                     query/phrase_query.rs:262-330, scorer/phrase_scorer.rs:131-160): every doc that
//                                     holds all terms is appended to the query's candidate list;
//   2. k_phrase_match_lanes           (packed segments) 64 candidates per wavefront, one per lane; what it hands on, and every
//      / k_phrase_match               candidate of a legacy segment, one wavefront per candidate. Either way, per candidate:
//                                     for every term find the doc's posting (block
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
// Sloppy phrases (slop > 0): k_sloppy_match_lanes (2..6 distinct terms), k_sloppy_groups + k_sloppy_match below instead.
// Fields with payloads or offsets (a third file, .pay): the position BLOCKS are the same; the trailing VInt block of a term
// carries payload bytes / offset words between its deltas (decode_vint_block_everything) and the skip entries two more words.
#pragma once
#include "search_and.hpp"

namespace rgpu {

constexpr int RGPU_MAX_K_DEV = 128;  // = RGPU_MAX_K: the widest list a wavefront's registers hold
constexpr int PHRASE_LIST_CAP = 1024;  // positions of one term inside one doc that the LDS lists hold
constexpr int32_t PHRASE_DEAD = (int32_t)0x80000000;

// The positions of `doc` in term T (its clause's PosTerm P), as (position - phrase offset), ascending, into L[0 .. freq):
// steps 1-3 of k_phrase_match's header comment. Returns freq (>= 1), or a negative code: -1 the doc is not in the term's
// postings (the conjunction said it is: internal error), -4 corrupt position data, -5 more than `cap` positions.
template <bool LEGACY>
__device__ __forceinline__ int phrase_doc_positions(const SegView& seg, const DevTerm& T, const PosTerm& P, int32_t doc, int64_t pos_len,
                                                    uint8_t* slab, int32_t* L, int cap, int lane) {
  // ---- 1. the doc's posting in this term: where its positions start in the term's position stream
  int64_t fp = (int64_t)P.pos_start_fp;
  int skip = 0, freq = 0;
  if (T.df == 1) {
    freq = T.singleton_freq;
  } else {
    // (a 64-ary search by the whole wavefront: two or three dependent loads where the one-lane binary search took fourteen)
    const int blk = find_block_wave(seg.dir_last, T.dir_base, 0, T.nblocks, doc, lane);
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
    if (!(m0 | m1)) return -1;  // the conjunction said the doc is here
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
  if (freq <= 0) return -4;
  if (freq > cap) return -5;
  // ---- 2. whole position blocks that hold only earlier docs' positions (ForUtil::skip_block, for_util.rs:263-272)
  while (skip >= 128) {
    if (fp == P.last_pos_block_fp || fp + 2 > pos_len) return -4;
    const uint32_t b = seg.pos[fp];
    if (b > 32u) return -4;
    int vlen = 0;
    if (b == 0) (void)read_vint_uniform(seg.pos + fp + 1, &vlen);
    fp += 1 + (b ? 16 * (int64_t)b : (int64_t)vlen);
    skip -= 128;
  }
  // ---- 3. this doc's `freq` positions: deltas from the stream, a running sum from 0 (posting_reader.rs:1357-1380)
  int got = 0;
  int32_t carry = 0;
  while (got < freq) {
    uint32_t x0, x1;
    int nvals = 128;
    if (fp < 0 || fp + 2 > pos_len) return -4;
    if (fp == P.last_pos_block_fp) {
      nvals = (int)(P.total_term_freq % 128);
      if (seg.pos_tail_flags == 0) decode_vint_block(seg.pos + fp, slab, lane, x0, x1);
      else if (!decode_vint_block_everything(seg.pos + fp, pos_len - fp, nvals, seg.pos_tail_flags, slab, lane, x0, x1)) return -4;
      fp = -2;  // nothing follows the trailing block
    } else {
      const uint32_t b = seg.pos[fp];
      if (b > 32u) return -4;
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
    if (take <= 0) return -4;  // the stream ends before the doc's positions do
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
  return freq;
}

// (Round 4: a fixed launch whose wavefronts stride — or take contiguous stretches — over the candidates instead of one wavefront
// per slot was slower by more than two, same box. Those timings, like every timing of this kernel on the 100 M-slot benchmark
// batch before the launches were cut into PHRASE_LAUNCH_SLOTS pieces, covered the first 32.6 M slots only: the runtime keeps the
// low 32 bits of a grid's work-item count. What the kernel is bound by is scalar issue — ~1300 instructions per candidate, two
// thirds of them scalar — which is why the 64-candidate kernels below exist.)
// slops (nullable): per query PhraseQuery::slop; this kernel serves the queries with slop 0 (the others: k_sloppy_match)
// CAP: positions of one term inside one doc that the wavefront's two LDS lists hold: two lists of PHRASE_LIST_CAP
// positions are 32 KB per workgroup — three wavefronts per SIMD. Nearly every doc holds a term a handful of times (Rucene
// clamps freqs to 10 at write time), so the launch runs with PHRASE_SMALL_CAP-entry lists, eight wavefronts per SIMD; a candidate
// that does not fit leaves PHRASE_REDO in its key and raises *redo — the host then runs the CAP = PHRASE_LIST_CAP instantiation,
// which looks at the marked slots only (REDO_ONLY).
constexpr int PHRASE_SMALL_CAP = 128;
constexpr int64_t PHRASE_LAUNCH_SLOTS = (int64_t)1 << 25;  // slots (wavefronts) per launch of a one-wavefront-per-slot kernel: 2^31 work-items
constexpr uint64_t PHRASE_REDO = 1ull;  // (no real key has a zero high word)
// bits of *redo: which later pass some candidate is waiting for
constexpr int PHRASE_REDO_WIDE = 1;    // k_phrase_match with PHRASE_LIST_CAP lists
constexpr int PHRASE_REDO_LANES = 2;   // k_phrase_match at all (left by k_phrase_match_lanes)
constexpr int PHRASE_REDO_SLOPPY = 4;  // k_sloppy_match with the SLOPPY_POOL pool
constexpr int PHRASE_REDO_SLOPPY_LANES = 8;  // k_sloppy_match at all (left by k_sloppy_match_lanes)
template <bool LEGACY, int CAP, bool REDO_ONLY>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_match(SegView seg, const DevQuery* __restrict__ queries,
                                                             const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                             const int64_t* __restrict__ emit_prefix,
                                                             const unsigned long long* __restrict__ emit_count,
                                                             const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops, int n_queries,
                                                             int64_t n_slots, int64_t pos_len, uint64_t* __restrict__ keys_out, int* err, int* redo,
                                                             const int64_t* __restrict__ redo_list, int64_t first) {
  // first / n_slots: this launch takes the slots [first, n_slots) — a grid holds at most 2^32 work-items, i.e. 2^26 wavefronts
  // (PHRASE_LAUNCH_SLOTS); redo_list (REDO_ONLY; nullable): the slots to look at, [first, n_slots) then indexes the list
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ int32_t lists_a[WG_WAVES][CAP];
  __shared__ int32_t lists_c[WG_WAVES][CAP];
  __shared__ float caches[WG_WAVES][256];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t work = first + (int64_t)blockIdx.x * WG_WAVES + wave;
  if (work >= n_slots) return;
  const int64_t slot = (REDO_ONLY && redo_list != nullptr) ? redo_list[work] : work;
  if (REDO_ONLY && keys_out[slot] != PHRASE_REDO) return;
  const int q = upper_slot_wave(emit_prefix, n_queries, slot, lane);
  if (slops != nullptr && slops[q] > 0) return;
  const int64_t idx = slot - emit_prefix[q];
  if ((unsigned long long)idx >= emit_count[q]) {  // the conjunction produced fewer matches than the lead term has docs
    if (lane == 0) keys_out[slot] = 0ull;
    return;
  }
  const int32_t doc = emit_docs[slot];
  if (doc < 0) {  // a deleted doc (marked by the conjunction): an approximation that is never checked (bulk_scorer.rs:100)
    if (lane == 0) keys_out[slot] = 0ull;
    return;
  }
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
    const int freq = phrase_doc_positions<LEGACY>(seg, T, P, doc, pos_len, slab, c == 0 ? A : C, CAP, lane);
    if (freq == -5 && CAP < PHRASE_LIST_CAP) {  // does not fit the small lists: the big instantiation takes this candidate
      if (lane == 0) { keys_out[slot] = PHRASE_REDO; atomicOr(redo, PHRASE_REDO_WIDE); }
      return;
    }
    if (freq < 0) { give_up(freq); return; }
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

// ---- exact phrases, 64 candidates per wavefront ------------------------------------------------------------------------------------
// k_phrase_match above spends ~1300 instructions on ONE candidate, two thirds of them scalar (the control flow of the look-ups is
// wave-uniform): rocprofv3 puts the kernel at the scalar issue rate, not at memory. Here a wavefront takes 64 consecutive slots of
// one query's candidate list (the host pads every query's slots to a multiple of 64), one candidate per lane, and the scalar work is
// shared: per term, every lane finds its doc's block by a binary search over the directory (gathers), the DISTINCT blocks among the
// lanes are decoded once each by the whole wavefront (docs, freqs and the running sum of freqs parked in LDS; each lane of that block
// finds its doc there), and then every lane reads its own doc's position deltas straight out of the packed position block —
// value i of a BP128 block of width b sits at bit (i >> 2) * b of 32-bit column i & 3 (packed_simd.rs:126-163), two dword gathers.
// Each lane keeps its lists in its own LDS column. The kernel only takes what is common: packed (non-legacy) blocks, at most
// PHRASE_LANE_CAP positions per term and doc (Rucene clamps freqs to 10 when it writes), positions inside packed position blocks.
// Anything else — a doc in the trailing VInt block of a term's positions, an all-equal block, more positions, a corrupt stream —
// leaves PHRASE_REDO in the candidate's slot and raises bit 1 of *redo: k_phrase_match<.., REDO_ONLY> takes those candidates (and
// reports the errors). Results are the same by construction: the same positions, the same intersection, the same score expression.
#ifndef RGPU_PHRASE_LANE_CAP
#define RGPU_PHRASE_LANE_CAP 10
#endif
constexpr int PHRASE_LANE_CAP = RGPU_PHRASE_LANE_CAP;
#ifndef RGPU_LANES_ABL  // developer ablations (variant builds only; results are wrong): 1 no positions / intersection, 2 + no block decodes
#define RGPU_LANES_ABL 0
#endif
constexpr int64_t PHRASE_REDO_LIST_CAP = 1 << 25;  // slots the list of left-over candidates holds (at most; 256 MB for a batch of that many slots); beyond: every slot is looked at
__device__ __forceinline__ uint32_t bp128_value_at(const uint8_t* __restrict__ payload, uint32_t b, int i) {
  const uint32_t p = (uint32_t)(i >> 2) * b;
  const uint8_t* at = payload + 4 * (i & 3) + 16 * (p >> 5);
  const uint64_t win = ((uint64_t)load4_unaligned(at + 16) << 32) | load4_unaligned(at);
  return (uint32_t)(win >> (p & 31)) & (0xffffffffu >> (32 - b));
}
// One term of a 64-candidate wavefront (steps 1-3 of the kernels below): the positions of every lane's `doc` in term T, as
// (position - phrase offset), ascending, into the lane's column of Lc (entry j at Lc[64 j + lane]). `act`: the lane holds a
// candidate; `again` (in / out): the candidate is handed on to the one-candidate kernel. Returns the lane's freq, 0 for a lane
// that takes no part (or has just been handed on). `area`: 384 words of LDS (the block decoder's staging area, then the decoded
// block {doc, freqs before, freq} x 128). ListT = int16_t: a position outside its range hands the candidate on.
// The end of a 64-candidate kernel: every lane's key goes to its slot; the slots handed on to a one-candidate kernel are marked
// PHRASE_REDO, `bit` is raised in *redo and the slots are listed (that pass then costs what they cost, not a wavefront per slot
// of the launch); a list that overflows is ignored by the host: *redo_n says so.
__device__ __forceinline__ void lanes_finish(bool again, uint64_t key, int64_t slot, uint64_t* __restrict__ keys_out, int* redo, int bit,
                                             int64_t* __restrict__ redo_list, int redo_cap, int* redo_n, int lane) {
  keys_out[slot] = again ? PHRASE_REDO : key;
  const uint64_t m = __ballot(again);
  if (m) {
    int at = 0;
    if (lane == 0) { atomicOr(redo, bit); at = atomicAdd(redo_n, (int)__popcll(m)); }
    at = readlane(at, 0) + (int)mbcnt(m);
    if (again && at < redo_cap) redo_list[at] = slot;
  }
}

template <typename ListT>
__device__ __forceinline__ int lanes_doc_positions(const SegView& seg, const DevTerm& T, const PosTerm& P, int32_t doc, bool act, bool& again,
                                                   int64_t pos_len, int32_t* area, ListT* Lc, int lane) {
  uint8_t* slab = reinterpret_cast<uint8_t*>(area);
  int32_t* Dd = area;
  int32_t* Db = Dd + 128;
  int32_t* Df = Dd + 256;
  int freq = 0, skip = 0;
  uint32_t pofs = 0;  // of the position block the doc's block starts in, from the term's pos_start_fp
  if (T.df == 1) {
    freq = T.singleton_freq;
    if (act && doc != T.singleton_doc) again = true;  // (the conjunction said the doc is here)
  } else {
    // ---- 1. every lane: the first directory slot whose last doc is >= its doc (slot nblocks: the tail)
    // (four-way: the three probes of a round are in flight together — half the dependent round trips of a binary search)
    int lo = 0, hi = T.nblocks;
    while (__ballot(lo < hi)) {
      const int span = hi - lo;
      const int m1 = lo + (span >> 2), m2 = lo + (span >> 1), m3 = lo + span - (span >> 2) - (span > 3 ? 0 : 1);
      const int top = T.nblocks - 1;
      const int32_t l1 = seg.dir_last[T.dir_base + min(max(m1, 0), top)];
      const int32_t l2 = seg.dir_last[T.dir_base + min(max(m2, 0), top)];
      const int32_t l3 = seg.dir_last[T.dir_base + min(max(m3, 0), top)];
      if (lo < hi) {  // lo <= m1 <= m2 <= m3 < hi; slots below lo end before doc, slot hi does not
        if (l1 >= doc) hi = m1;
        else if (l2 >= doc) { lo = m1 + 1; hi = m2; }
        else if (l3 >= doc) { lo = m2 + 1; hi = m3; }
        else lo = m3 + 1;
      }
    }
    const int blk = lo;
    // ---- 2. the distinct blocks among the lanes, decoded once each
    uint64_t pend = __ballot(act && !again);
    if (RGPU_LANES_ABL == 2) { pend = 0; freq = 1 + (blk & 1); }
    while (pend) {
      const int b = readlane(blk, (int)__builtin_ctzll(pend));
      const uint64_t st = seg.dir_pos[T.dir_base + b];
      const uint32_t row = seg.dir_row[T.dir_base + b];
      int32_t e0, e1;
      uint32_t g0, g1;
      if (b < T.nblocks) {
        const uint32_t hdr = (uint32_t)seg.dir_hdr[T.dir_base + b];
        const int32_t base = b > 0 ? seg.dir_last[T.dir_base + b - 1] : 0;
        const BlockPair bp = decode_block<false>(seg.bstore + T.bs_base, row, hdr, slab, lane);
        deltas_to_docs(bp.d0, bp.d1, base, e0, e1);
        g0 = bp.f0; g1 = bp.f1;
      } else if (T.tail_n > 0) {
        tail_load(seg.bstore + T.bs_base, row, lane, e0, e1, g0, g1);  // (INT_MAX / 0 past the tail's end)
      } else {
        e0 = e1 = 0x7fffffff; g0 = g1 = 0u;
      }
      const int pair = (int)(g0 + g1);
      const int excl = wave_incl_scan(pair) - pair;
      Dd[2 * lane] = e0; Dd[2 * lane + 1] = e1;
      Db[2 * lane] = excl; Db[2 * lane + 1] = excl + (int)g0;
      Df[2 * lane] = (int32_t)g0; Df[2 * lane + 1] = (int32_t)g1;
      wave_sync();
      const bool mine = act && !again && blk == b;
      if (mine) {
        int at = 0;  // the first of the block's 128 docs that is >= doc
#pragma unroll
        for (int step = 64; step >= 1; step >>= 1) at += Dd[at + step - 1] < doc ? step : 0;
        if (Dd[at] != doc) {
          again = true;  // (the conjunction said the doc is here)
        } else {
          freq = Df[at];
          skip = (int)(st >> 32) + Db[at];
          pofs = (uint32_t)st;
        }
      }
      pend &= ~__ballot(blk == b);
      wave_sync();  // the area is rewritten by the next block
    }
  }
  bool live = act && !again;
  if (live && (freq <= 0 || freq > PHRASE_LANE_CAP)) { again = true; live = false; }
  // ---- 3. the doc's positions: `freq` deltas from value `skip` of the position stream at pofs on
  if (RGPU_LANES_ABL == 1 || RGPU_LANES_ABL == 2) return live ? freq : 0;
  // (a lane without a candidate reads the term's first position block, value 0: the loads below are unconditional)
  int64_t fp = (int64_t)P.pos_start_fp, fp1 = fp;
  uint32_t b0 = 1u, b1 = 1u;
  if (live) {
    fp += (int64_t)pofs;
    // a packed position block at `at`: its header byte (1..32); 0 = not one (the trailing VInt block, an all-equal block, the end)
    auto packed_at = [&](int64_t at) -> uint32_t {
      if (at < 0 || at + 2 > pos_len || at == P.last_pos_block_fp) return 0u;
      const uint32_t b = seg.pos[at];
      // the whole block must lie inside the file: a truncated one is handed on to k_phrase_match, which reports it
      // as RGPU_ERR_CORRUPT_INDEX, instead of being read from the padding behind the file
      return (b <= 32u && at + 1 + 16 * (int64_t)b <= pos_len) ? b : 0u;
    };
    b0 = packed_at(fp);
    if (RGPU_LANES_ABL == 5) skip &= 127;
    while (b0 != 0u && skip >= 128) {  // whole blocks of earlier docs' positions (ForUtil::skip_block, for_util.rs:263-272)
      fp += 1 + 16 * (int64_t)b0;
      skip -= 128;
      b0 = packed_at(fp);
    }
    fp1 = fp + 1 + 16 * (int64_t)b0;  // the block behind: a doc's <= 10 positions straddle at most one boundary
    const bool straddles = skip + freq > 128;
    if (b0 != 0u && straddles) b1 = packed_at(fp1);
    if (b0 == 0u || (straddles && b1 == 0u)) {
      again = true;
      live = false;
    }
  }
  if (!live) { fp = fp1 = (int64_t)P.pos_start_fp; b0 = b1 = 1u; skip = 0; freq = 1; }
  // every lane's deltas asked for together (a load behind a per-lane branch waits for the one in front of it: ten round trips
  // where this takes one); j runs to the largest freq among the lanes, a lane past its own freq reads its last delta again
  const int maxf = (int)wave_reduce_max_u32(live ? (uint32_t)freq : 0u);
  // (first every load, then every use: a value unpacked inside the j-th step would make that step wait for its own two loads)
  uint32_t wlo[PHRASE_LANE_CAP], whi[PHRASE_LANE_CAP];
#pragma unroll
  for (int j = 0; j < PHRASE_LANE_CAP; ++j) {
    wlo[j] = whi[j] = 0u;
    if (j < maxf && RGPU_LANES_ABL != 4) {  // (wave-uniform)
      const int i = skip + min(j, freq - 1);
      const bool behind = i >= 128;
      const uint32_t p = (uint32_t)((behind ? i - 128 : i) >> 2) * (behind ? b1 : b0);
      const uint8_t* at = seg.pos + (behind ? fp1 : fp) + 1 + 4 * (i & 3) + 16 * (p >> 5);
      wlo[j] = load4_unaligned(at);
      whi[j] = load4_unaligned(at + 16);
    }
  }
  int32_t at_pos = -P.phrase_pos;  // (position - phrase offset; the doc's first delta is its first position)
  bool narrow = false;       // some position does not fit the list's element type
#pragma unroll
  for (int j = 0; j < PHRASE_LANE_CAP; ++j) {
    if (j < maxf) {  // (wave-uniform; not a `break`: leaving the unrolled loop early cost 130 registers in copies of wlo / whi)
      const int i = skip + min(j, freq - 1);
      const bool behind = i >= 128;
      const uint32_t b = behind ? b1 : b0;
      const uint32_t p = (uint32_t)((behind ? i - 128 : i) >> 2) * b;
      const uint32_t v = (uint32_t)((((uint64_t)whi[j] << 32) | wlo[j]) >> (p & 31)) & (0xffffffffu >> (32 - b));
      at_pos += (int32_t)v;
      if (live && j < freq) {
        Lc[j * 64 + lane] = (ListT)at_pos;
        narrow = narrow || (int32_t)(ListT)at_pos != at_pos;
      }
    }
  }
  if (live && narrow) { again = true; live = false; }
  return live ? freq : 0;
}

__global__ __launch_bounds__(WG_THREADS) void k_phrase_match_lanes(SegView seg, const DevQuery* __restrict__ queries,
                                                                   const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                                   const int64_t* __restrict__ emit_prefix,
                                                                   const unsigned long long* __restrict__ emit_count,
                                                                   const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops, int n_queries,
                                                                   int64_t n_groups, int64_t pos_len, uint64_t* __restrict__ keys_out, int* redo,
                                                                   int64_t* __restrict__ redo_list, int redo_cap, int* redo_n) {
  __shared__ __attribute__((aligned(16))) int32_t areas[WG_WAVES][384];
  __shared__ int32_t lists[WG_WAVES][2][PHRASE_LANE_CAP * 64];
  static_assert(sizeof(int32_t) * 384 >= 2 * SLAB_STREAM, "the staging area holds a block's doc and freq rows");
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t group = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (group >= n_groups) return;
  const int64_t slot = group * 64 + lane;
  const int q = upper_slot_wave(emit_prefix, n_queries, group * 64, lane);
  if (slops != nullptr && slops[q] > 0) return;
  const int64_t idx = slot - emit_prefix[q];
  const int64_t cnt = (int64_t)emit_count[q];
  if (idx - lane >= cnt) return;  // nothing in these 64 slots (the collectors read the first emit_count[q] slots only)
  int32_t doc = idx < cnt ? emit_docs[slot] : -1;
  bool act = doc >= 0;    // (a deleted doc travels with its sign bit set: an approximation that is never checked, bulk_scorer.rs:100)
  bool again = false;     // this candidate goes to k_phrase_match
  const DevQuery Q = queries[q];
  int32_t* A = lists[wave][0];
  int32_t* C = lists[wave][1];
  int n_a = 0;
  for (int c = 0; c < Q.n_terms; ++c) {
    if (!__ballot(act && !again)) break;
    const DevTerm T = terms[Q.first_term + c];
    const PosTerm P = pterms[Q.first_term + c];
    const int freq = lanes_doc_positions<int32_t>(seg, T, P, doc, act, again, pos_len, areas[wave], c == 0 ? A : C, lane);
    const bool live = freq > 0;
    if (RGPU_LANES_ABL == 1 || RGPU_LANES_ABL == 2) continue;
    // ---- 4. keep the first term's positions that line up with this term's (every lane reads and writes its own column only)
    if (c == 0) {
      n_a = live ? freq : 0;
    } else if (live && RGPU_LANES_ABL != 3) {
      // both lists ascend (a doc's positions do; dead entries of A are skipped): one pass over the two, not a search per entry
      int alive = 0, i = 0, j = 0;
      int32_t cv = C[lane];
      while (i < n_a) {
        const int32_t a = A[i * 64 + lane];
        if (a == PHRASE_DEAD) { ++i; continue; }
        while (cv < a && j + 1 < freq) { ++j; cv = C[j * 64 + lane]; }
        if (cv == a) ++alive; else A[i * 64 + lane] = PHRASE_DEAD;
        ++i;
      }
      if (alive == 0) act = false;  // no position of the first term lines up any more: phrase freq 0, key 0
    }
  }
  uint64_t key = 0ull;
  if (act && !again) {
    int phrase_freq = 0;
    for (int i = 0; i < n_a; ++i) phrase_freq += A[i * 64 + lane] != PHRASE_DEAD ? 1 : 0;
    if (phrase_freq > 0) {
      const DevTerm T0 = terms[Q.first_term];
      const float* table = seg.sim_tables + (size_t)T0.sim_table * 257;
      const float k1 = table[256];
      float nrm = k1;
      if (seg.norms != nullptr) {
        const uint32_t nb = seg.norms[doc];
        nrm = table[seg.n_norm_ranks > 0 ? (uint32_t)seg.rank_to_norm[nb] : nb];
      }
      key = make_key(bm25_score(T0.weight * (k1 + 1.0f), (float)phrase_freq, nrm), doc);
    }
  }
  lanes_finish(again, key, slot, keys_out, redo, PHRASE_REDO_LANES, redo_list, redo_cap, redo_n, lane);
}

// ---- SloppyPhraseScorer (scorer/phrase_scorer.rs:432-1071; PhraseQuery with slop > 0) ------------------------------------------
// The reference scores a candidate doc by walking its PhrasePositions (one per phrase term, position = term position -
// phrase offset) through a priority queue on (position, offset, ord): the least one is advanced, and every time it passes
// the next one the span end - position of that moment counts 1 / (span + 1) if it is within the slop (phrase_freq,
// :537-577). It is sequential and stateful: a wavefront takes ONE candidate doc and runs it with the whole state in
// registers — lane i = PhrasePositions i (QUERY order: the order decides ties in the queue), the queue's array in the
// lanes too — so that every step is scalar control flow over readlane / writelane; the positions of all terms sit in one
// LDS pool. Repeated terms ("a b a") go through the reference's collision machinery (advance_rpts, :651-701), which moves
// PhrasePositions that are INSIDE the queue: what a pop then returns depends on the ARRAY Rust's BinaryHeap keeps, so the
// heap is emulated operation by operation (push = sift_up, pop = last element into the root, sift_down_to_bottom, sift_up:
// util/external/binary_heap.rs:121-210), not replaced by "take the minimum".
// The repetition groups are what the reference finds on the FIRST candidate doc of the leaf (init_first_time, :805-871:
// repeating pps whose first positions coincide there): k_sloppy_groups does exactly that, once per query, on the query's
// smallest candidate doc.
constexpr int SLOPPY_POOL = 2048;  // positions of all the phrase's terms inside one doc that the LDS pool holds
constexpr int SLOPPY_SMALL_POOL = 256;  // ... in the first launch (k_sloppy_match<.., POOL, REDO_ONLY>)
constexpr int SLOPPY_MAX_TERMS = 16;

// per query, reduced chunk by chunk over its candidates (k_phrase_cutoff_items): the smallest doc with a key, the smallest LIVE
// candidate (what SloppyPhraseScorer::init_first_time sees: k_sloppy_groups), the candidates in front of the first doc with a key
struct PhraseCut {
  int32_t first;
  int32_t dmin;
  unsigned long long before;
};
struct SloppyGroups {  // per query: PhrasePositions::{rpt_group, rpt_ind} by query-order index; -1 = not a repeater
  int8_t grp[SLOPPY_MAX_TERMS];
  int8_t ind[SLOPPY_MAX_TERMS];
};

template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_sloppy_groups(SegView seg, const DevQuery* __restrict__ queries, const DevTerm* __restrict__ terms,
                                                              const PosTerm* __restrict__ pterms, const int64_t* __restrict__ emit_prefix,
                                                              const unsigned long long* __restrict__ emit_count,
                                                              const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops,
                                                              int n_queries, int64_t pos_len, SloppyGroups* __restrict__ groups, int* err,
                                                              const PhraseCut* __restrict__ cut) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ int32_t lists[WG_WAVES][PHRASE_LIST_CAP];
  const int lane = lane_id();
  const int wave = wave_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave;
  if (q >= n_queries) return;
  int32_t out_grp = -1, out_ind = 0;  // lane i: pp i's group and index in it (written by the lanes themselves: a local struct filled
                                      // through a loop variable lived in scratch, 36 bytes per lane)
  const DevQuery Q = queries[q];
  const int64_t n_cand = (int64_t)emit_count[q];
  const int n = Q.n_terms;
  // lane i: pp i (query order): its offset, which earlier pp names the same term, its device clause
  int32_t off = 0, same_as = lane, clause = 0;
  for (int c = 0; c < n; ++c) {
    const PosTerm P = pterms[Q.first_term + c];
    if (lane == P.query_ord) { off = P.phrase_pos; same_as = P.same_as; clause = c; }
  }
  bool repeats = false;  // this pp's term occurs more than once in the phrase (repeating_terms / repeating_pps)
  for (int i = 0; i < n; ++i) {
    const int s_i = readlane(same_as, i);
    repeats = repeats || (lane < n && lane != i && same_as == s_i);
  }
  const uint64_t rpp = __ballot(repeats);
  if (slops[q] > 0 && rpp != 0ull && n_cand > 0) {
    // the first candidate doc of the leaf: the conjunction's smallest live match
    // (BulkScorer tests live docs before it calls matches() — bulk_scorer.rs:100 — so the scorer's init_first_time runs on the
    // first LIVE match; deleted candidates carry the sign bit. Found chunk by chunk by k_phrase_cutoff_items<2>.)
    (void)emit_prefix;
    (void)emit_docs;
    const int32_t dmin = cut[q].dmin;
    // Every candidate of the leaf is a deleted doc (the conjunction emits them with the sign bit set, so n_cand > 0
    // does not promise a live one): matches() is never called in this leaf, init_first_time never runs — the groups stay at
    // their -1 defaults and no position is looked up (the old code went looking for doc 0x7fffffff and failed the batch).
    const bool any_live = dmin != 0x7fffffff;
    // tp_pos of every repeating pp there = the term's first position in that doc
    int32_t tp = 0;
    uint64_t m = any_live ? rpp : 0ull;
    while (m) {
      const int i = (int)__builtin_ctzll(m);
      m &= m - 1;
      const int c = readlane(clause, i);
      const int freq = phrase_doc_positions<LEGACY>(seg, terms[Q.first_term + c], pterms[Q.first_term + c], dmin, pos_len, slabs[wave], lists[wave],
                                                    PHRASE_LIST_CAP, lane);
      if (freq < 0) { if (lane == 0) atomicMin(err, freq == -1 ? -1 : freq); break; }
      const int32_t first = lists[wave][0] + readlane(off, i);  // (the loader stores position - offset)
      tp = lane == i ? first : tp;
      wave_sync();
    }
    // gather_rpt_groups, the arm without multi-term postings (:841-871), then sort_rpt_groups (:826-838)
    int32_t grp = -1;
    int n_groups = 0;
    uint64_t m1 = any_live ? rpp : 0ull;
    while (m1) {
      const int i1 = (int)__builtin_ctzll(m1);
      m1 &= m1 - 1;
      if (readlane(grp, i1) >= 0) continue;  // already marked as a repetition
      const int32_t tp1 = readlane(tp, i1), off1 = readlane(off, i1);
      const uint64_t joins = __ballot(repeats && lane > i1 && grp < 0 && off != off1 && tp == tp1);
      if (joins) {
        grp = (lane == i1 || ((joins >> lane) & 1ull)) ? n_groups : grp;
        ++n_groups;
      }
    }
    // a member's index in its group: members ordered by (offset, pp index) — a stable sort of the discovery order by offset
    int32_t ind = 0;
    for (int j = 0; j < n; ++j) {
      const int32_t gj = readlane(grp, j), oj = readlane(off, j);
      ind += (grp >= 0 && gj == grp && (oj < off || (oj == off && j < lane))) ? 1 : 0;
    }
    if (lane < n) { out_grp = grp; out_ind = ind; }
  }
  if (lane < SLOPPY_MAX_TERMS) {
    groups[q].grp[lane] = (int8_t)out_grp;
    groups[q].ind[lane] = (int8_t)out_ind;
  }
}

// POOL / REDO_ONLY: as k_phrase_match's CAP — the launch runs with SLOPPY_SMALL_POOL positions per wavefront (seven wavefronts per
// SIMD instead of three); a doc whose terms hold more leaves PHRASE_REDO and the POOL = SLOPPY_POOL instantiation takes it.
template <bool LEGACY, int POOL, bool REDO_ONLY>
__global__ __launch_bounds__(WG_THREADS) void k_sloppy_match(SegView seg, const DevQuery* __restrict__ queries,
                                                             const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                             const int64_t* __restrict__ emit_prefix,
                                                             const unsigned long long* __restrict__ emit_count,
                                                             const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops,
                                                             const SloppyGroups* __restrict__ groups, int n_queries, int64_t n_slots,
                                                             int64_t pos_len, uint64_t* __restrict__ keys_out, int* err, int* redo,
                                                             const int64_t* __restrict__ redo_list, int64_t first) {
  // (first / n_slots / redo_list: as k_phrase_match's)
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ int32_t pools[WG_WAVES][POOL];
  __shared__ float caches[WG_WAVES][256];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t work = first + (int64_t)blockIdx.x * WG_WAVES + wave;
  if (work >= n_slots) return;
  const int64_t slot = (REDO_ONLY && redo_list != nullptr) ? redo_list[work] : work;
  if (REDO_ONLY && keys_out[slot] != PHRASE_REDO) return;
  const int q = upper_slot_wave(emit_prefix, n_queries, slot, lane);
  const int slop = slops[q];
  if (slop <= 0) return;  // an exact phrase: k_phrase_match's
  const int64_t idx = slot - emit_prefix[q];
  if ((unsigned long long)idx >= emit_count[q]) {
    if (lane == 0) keys_out[slot] = 0ull;
    return;
  }
  const int32_t doc = emit_docs[slot];
  if (doc < 0) {  // a deleted doc (marked by the conjunction): an approximation that is never checked (bulk_scorer.rs:100)
    if (lane == 0) keys_out[slot] = 0ull;
    return;
  }
  const DevQuery Q = queries[q];
  const int n = Q.n_terms;
  int32_t* pool = pools[wave];
  auto give_up = [&](int code) {
    if (lane == 0) { atomicMin(err, code); keys_out[slot] = 0ull; }
  };
  // ---- every term's positions in this doc; lane i keeps PhrasePositions i (query order)
  int32_t s_pos = 0, s_cnt = 0, s_at = 0, s_off = 0, s_start = 0, s_n = 0;
  int used = 0;
  for (int c = 0; c < n; ++c) {
    const DevTerm T = terms[Q.first_term + c];
    const PosTerm P = pterms[Q.first_term + c];
    const int freq = phrase_doc_positions<LEGACY>(seg, T, P, doc, pos_len, slabs[wave], pool + used, POOL - used, lane);
    if (freq == -5 && POOL < SLOPPY_POOL) {  // does not fit the small pool: the big instantiation takes this candidate
      if (lane == 0) { keys_out[slot] = PHRASE_REDO; atomicOr(redo, PHRASE_REDO_SLOPPY); }
      return;
    }
    if (freq < 0) { give_up(freq); return; }
    if (lane == P.query_ord) { s_off = P.phrase_pos; s_start = used; s_n = freq; }
    used += freq;
  }
  const SloppyGroups G = groups[q];
  int32_t s_grp = -1, s_ind = 0;
#pragma unroll
  for (int i = 0; i < SLOPPY_MAX_TERMS; ++i) { s_grp = lane == i ? (int32_t)G.grp[i] : s_grp; s_ind = lane == i ? (int32_t)G.ind[i] : s_ind; }
  const bool has_rpts = __ballot(lane < n && s_grp >= 0) != 0ull;  // (a phrase whose repeated terms found no group behaves like one without)
  bool any_rpt_terms = false;
  {  // has_rpts of the reference = "some term repeats" (repeating_terms), whatever the groups turned out to be
    int32_t same_as = lane;
    for (int c = 0; c < n; ++c) { const PosTerm P = pterms[Q.first_term + c]; if (lane == P.query_ord) same_as = P.same_as; }
    for (int i = 0; i < n; ++i) any_rpt_terms = any_rpt_terms || (__ballot(lane < n && lane != i && same_as == readlane(same_as, i)) != 0ull);
  }
  (void)has_rpts;
  // ---- scalar helpers over the lanes' state (every index is wave-uniform)
  auto next_position = [&](int i) -> bool {  // PhrasePositions::next_position (:363-373)
    const int c = readlane(s_cnt, i);
    if (c <= 0) return false;
    const int a = readlane(s_at, i);
    const int32_t p = pool[readlane(s_start, i) + a];  // (position - offset already)
    s_cnt = lane == i ? c - 1 : s_cnt;
    s_at = lane == i ? a + 1 : s_at;
    s_pos = lane == i ? p : s_pos;
    return true;
  };
  auto first_position = [&](int i) {  // :356-360
    s_cnt = lane == i ? s_n : s_cnt;
    s_at = lane == i ? 0 : s_at;
    (void)next_position(i);
  };
  int32_t end = (int32_t)0x80000000;
  auto advance_pp = [&](int i) -> bool {  // :638-646
    if (!next_position(i)) return false;
    const int32_t p = readlane(s_pos, i);
    end = p > end ? p : end;
    return true;
  };
  // std BinaryHeap<PPElement>: lane j = data[j]; PPElement's reversed order: a <= b  <=>  key(a) >= key(b)
  int32_t heap = 0;
  int hn = 0;
  auto key_less = [&](int a, int b) -> bool {
    const int32_t pa = readlane(s_pos, a), pb = readlane(s_pos, b);
    if (pa != pb) return pa < pb;
    const int32_t oa = readlane(s_off, a), ob = readlane(s_off, b);
    if (oa != ob) return oa < ob;
    return a < b;
  };
  auto sift_up = [&](int start, int pos) {
    const int elt = readlane(heap, pos);
    while (pos > start) {
      const int parent = (pos - 1) / 2;
      const int pe = readlane(heap, parent);
      if (!key_less(elt, pe)) break;  // elt <= parent
      heap = lane == pos ? pe : heap;
      pos = parent;
    }
    heap = lane == pos ? elt : heap;
  };
  auto heap_push = [&](int i) {
    heap = lane == hn ? i : heap;
    ++hn;
    sift_up(0, hn - 1);
  };
  auto heap_pop = [&]() -> int {
    --hn;
    int item = readlane(heap, hn);
    if (hn > 0) {
      const int root = readlane(heap, 0);
      heap = lane == 0 ? item : heap;
      item = root;
      int pos = 0;
      const int elt = readlane(heap, 0);
      int child = 1;
      while (child < hn) {
        const int right = child + 1;
        if (right < hn && !key_less(readlane(heap, child), readlane(heap, right))) child = right;  // the greater of the two children
        const int ce = readlane(heap, child);
        heap = lane == pos ? ce : heap;
        pos = child;
        child = 2 * pos + 1;
      }
      heap = lane == pos ? elt : heap;
      sift_up(0, pos);
    }
    return item;
  };
  auto tp_pos = [&](int i) -> int32_t { return readlane(s_pos, i) + readlane(s_off, i); };
  auto member = [&](int g, int k) -> int {  // rpt_group[g][k]
    return (int)__builtin_ctzll(__ballot(lane < n && s_grp == g && s_ind == k) | (1ull << 63));
  };
  auto collide = [&](int i) -> int {  // :716-726: the group index of a pp of i's group standing on the same term position, or -1
    const int g = readlane(s_grp, i);
    const int len = __popcll(__ballot(lane < n && s_grp == g));
    const int32_t tp = tp_pos(i);
    for (int k = 0; k < len; ++k) {
      const int j = member(g, k);
      if (j != i && tp_pos(j) == tp) return k;
    }
    return -1;
  };
  auto lesser = [&](int a, int b) -> int {  // :704-713
    const int32_t pa = readlane(s_pos, a), pb = readlane(s_pos, b);
    return (pa < pb || (pa == pb && readlane(s_off, a) < readlane(s_off, b))) ? a : b;
  };
  auto advance_rpts = [&](int pp) -> bool {  // :651-701
    const int g = readlane(s_grp, pp);
    if (g < 0) return true;  // not a repeater
    const int len = __popcll(__ballot(lane < n && s_grp == g));
    uint32_t bits = 0;
    const int k0 = readlane(s_ind, pp);
    int cur = pp;
    while (true) {
      const int k = collide(cur);
      if (k < 0) break;
      cur = lesser(cur, member(g, k));  // always advance the lesser of the (only) two colliding pps
      if (!advance_pp(cur)) return false;
      if (k != k0) bits |= 1u << k;  // mark only those currently in the queue
    }
    // collisions resolved, now re-queue: pop until every marked pp has come out, then push them all back
    int32_t stack = 0;
    int ns = 0;
    while (bits && hn > 0) {  // (the reference would panic on an empty queue; it cannot get there: a marked pp is in the queue)
      const int p2 = heap_pop();
      stack = lane == ns ? p2 : stack;
      ++ns;
      const int g2 = readlane(s_grp, p2), k2 = readlane(s_ind, p2);
      if (g2 >= 0 && k2 < len && ((bits >> k2) & 1u)) bits &= ~(1u << k2);
    }
    for (int i = 0; i < ns; ++i) heap_push(readlane(stack, ns - 1 - i));
    return true;
  };
  // ---- init_phrase_positions (:590-627, 733-790): every doc starts the same way once the groups are known
  bool alive = true;
  for (int i = 0; i < n; ++i) first_position(i);  // place_first_positions
  if (any_rpt_terms) {  // advance_repeat_groups, single-term arm: the j-th pp of a group (by offset) advances j times
    const uint64_t grouped = __ballot(lane < n && s_grp >= 0);
    uint64_t m = grouped;
    // (group by group in the order the groups were found, members by index in the group: the order only matters for which
    // positions are consumed, and each pp's count of advances is fixed: its index in its group)
    while (m && alive) {
      const int i = (int)__builtin_ctzll(m);
      m &= m - 1;
      const int times = readlane(s_ind, i);
      for (int t = 0; t < times && alive; ++t) alive = next_position(i);
    }
  }
  float freq = 0.0f;
  if (alive) {
    // fill_queue (init_simple pushes in the same order)
    for (int i = 0; i < n; ++i) {
      const int32_t p = readlane(s_pos, i);
      end = p > end ? p : end;
      heap_push(i);
    }
    // ---- phrase_freq (:537-577)
    int pp = heap_pop();
    int32_t match_length = end - readlane(s_pos, pp);
    int32_t next = readlane(s_pos, readlane(heap, 0));
    while (advance_pp(pp)) {
      if (any_rpt_terms && !advance_rpts(pp)) break;  // pps exhausted
      const int32_t p = readlane(s_pos, pp);
      if (p > next) {  // done minimizing current match-length
        if (match_length <= slop) freq += 1.0f / ((float)match_length + 1.0f);  // compute_slop_factor (bm25_similarity.rs:65-67)
        heap_push(pp);
        pp = heap_pop();
        next = readlane(s_pos, readlane(heap, 0));
        match_length = end - readlane(s_pos, pp);
      } else {
        const int32_t ml2 = end - p;
        match_length = ml2 < match_length ? ml2 : match_length;
      }
    }
    if (match_length <= slop) freq += 1.0f / ((float)match_length + 1.0f);
  }
  uint64_t key = 0ull;
  if (freq > 1.1920929e-07f) {  // matches(): sloppy_freq > f32::EPSILON (:1041-1045)
    const DevTerm T0 = terms[Q.first_term];
    float k1;
    load_sim_table(seg, T0.sim_table, caches[wave], lane, k1);
    const float wk = T0.weight * (k1 + 1.0f);
    const float nrm = seg.norms != nullptr ? caches[wave][seg.norms[doc]] : k1;
    key = make_key(bm25_score(wk, freq, nrm), doc);
  }
  if (lane == 0) keys_out[slot] = key;
}

// ---- sloppy phrases, 64 candidates per wavefront ------------------------------------------------------------------------------------
// k_sloppy_match is the one-candidate kind (98 ms for 1024 two-term phrases with slop 2 over 10 M docs: scalar issue, as
// k_phrase_match was). Without repeated terms the scorer's queue never has its keys changed from outside: a pop is the smallest
// (position, offset, ord), the array order of Rust's BinaryHeap does not show — and the walk (phrase_freq, :537-577) is a few
// dozen steps over at most n x 10 positions. So each lane runs it for its own candidate: every term's positions in the lane's
// LDS columns (16-bit: position - offset of a doc longer than 32 k tokens hands the candidate on), the PhrasePositions' state in
// registers (unrolled over SLOPPY_LANE_TERMS), the f32 sum in the reference's order. Phrases with a repeated term, with one or
// more than SLOPPY_LANE_TERMS terms, and whatever lanes_doc_positions hands on, go to k_sloppy_match through the list.
constexpr int SLOPPY_LANE_TERMS = 6;
__global__ __launch_bounds__(WG_THREADS) void k_sloppy_match_lanes(SegView seg, const DevQuery* __restrict__ queries,
                                                                   const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                                   const int64_t* __restrict__ emit_prefix,
                                                                   const unsigned long long* __restrict__ emit_count,
                                                                   const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops, int n_queries,
                                                                   int64_t n_groups, int64_t pos_len, uint64_t* __restrict__ keys_out, int* redo,
                                                                   int64_t* __restrict__ redo_list, int redo_cap, int* redo_n, int rpt_lanes) {
  // rpt_lanes != 0: k_sloppy_rpt_lanes runs behind this launch and takes the phrases that repeat a term
  __shared__ __attribute__((aligned(16))) int32_t areas[WG_WAVES][384];
  __shared__ int16_t lists[WG_WAVES][SLOPPY_LANE_TERMS][PHRASE_LANE_CAP * 64];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t group = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (group >= n_groups) return;
  const int64_t slot = group * 64 + lane;
  const int q = upper_slot_wave(emit_prefix, n_queries, group * 64, lane);
  const int slop = slops[q];
  if (slop <= 0) return;  // an exact phrase: k_phrase_match_lanes'
  const int64_t idx = slot - emit_prefix[q];
  const int64_t cnt = (int64_t)emit_count[q];
  if (idx - lane >= cnt) return;
  const int32_t doc = idx < cnt ? emit_docs[slot] : -1;
  const bool act = doc >= 0;  // (a deleted doc: an approximation that is never checked)
  bool again = false;
  const DevQuery Q = queries[q];
  const int n = Q.n_terms;
  // the PhrasePositions' fixed parts (wave-uniform), clause by clause: offset, ord; a term that occurs twice sends the phrase away
  int32_t off[SLOPPY_LANE_TERMS], ord[SLOPPY_LANE_TERMS];
  bool takes = n >= 2 && n <= SLOPPY_LANE_TERMS;
#pragma unroll
  for (int c = 0; c < SLOPPY_LANE_TERMS; ++c) {
    off[c] = 0; ord[c] = c;
    if (takes && c < n) {
      const PosTerm P = pterms[Q.first_term + c];
      off[c] = P.phrase_pos; ord[c] = P.query_ord;
      if (P.same_as != P.query_ord) takes = false;
    }
  }
  // (a phrase that repeats a term is k_sloppy_rpt_lanes' when it has 2..SLOPPY_LANE_TERMS terms — that kernel writes its slots)
  if (!takes && rpt_lanes && n >= 2 && n <= SLOPPY_LANE_TERMS) return;
  if (!takes) again = act;
  // PPElement's order (:393-430) is (position, offset, ord): bit 8 t + u of `ties` = "at equal positions pp t comes before pp u"
  uint64_t ties = 0ull;
#pragma unroll
  for (int t = 0; t < SLOPPY_LANE_TERMS; ++t)
#pragma unroll
    for (int u = 0; u < SLOPPY_LANE_TERMS; ++u)
      if (off[t] < off[u] || (off[t] == off[u] && ord[t] < ord[u])) ties |= 1ull << (8 * t + u);
  // ---- every term's positions in the lane's doc
  int32_t fr[SLOPPY_LANE_TERMS];
#pragma unroll
  for (int t = 0; t < SLOPPY_LANE_TERMS; ++t) fr[t] = 0;
  for (int c = 0; takes && c < n; ++c) {
    if (!__ballot(act && !again)) break;
    const DevTerm T = terms[Q.first_term + c];
    const PosTerm P = pterms[Q.first_term + c];
    const int f = lanes_doc_positions<int16_t>(seg, T, P, doc, act, again, pos_len, areas[wave], lists[wave][c], lane);
#pragma unroll
    for (int t = 0; t < SLOPPY_LANE_TERMS; ++t) fr[t] = t == c ? f : fr[t];
  }
  const bool live = act && !again;
  float sfreq = 0.0f;
  if (live) {
    const int16_t* L = &lists[wave][0][0];
    // init_simple (:604-617): every pp on its first position, `end` the largest of them
    int32_t pos[SLOPPY_LANE_TERMS], nx[SLOPPY_LANE_TERMS];
    int32_t end = (int32_t)0x80000000;
#pragma unroll
    for (int t = 0; t < SLOPPY_LANE_TERMS; ++t) {
      pos[t] = 0x7fffffff; nx[t] = 1;
      if (t < n) { pos[t] = (int32_t)L[(t * PHRASE_LANE_CAP) * 64 + lane]; end = pos[t] > end ? pos[t] : end; }
    }
    // the pp the queue hands out is the least one
    auto least = [&](int& cur, int32_t& cur_pos, int32_t& next) {
      cur = 0; cur_pos = pos[0];
#pragma unroll
      for (int t = 1; t < SLOPPY_LANE_TERMS; ++t) {
        if (t < n) {
          const bool lt = pos[t] < cur_pos || (pos[t] == cur_pos && ((ties >> (8 * t + cur)) & 1ull) != 0ull);
          if (lt) { cur = t; cur_pos = pos[t]; }
        }
      }
      next = 0x7fffffff;  // the position on top of the queue once `cur` is out: the least position among the others
#pragma unroll
      for (int t = 0; t < SLOPPY_LANE_TERMS; ++t) if (t < n && t != cur) next = pos[t] < next ? pos[t] : next;
    };
    int cur;
    int32_t cur_pos, next;
    least(cur, cur_pos, next);
    int32_t match_length = end - cur_pos;
    // phrase_freq (:537-577)
    while (true) {
      int32_t kc = nx[0], fc = fr[0];
#pragma unroll
      for (int t = 1; t < SLOPPY_LANE_TERMS; ++t) { kc = cur == t ? nx[t] : kc; fc = cur == t ? fr[t] : fc; }
      if (kc >= fc) break;  // advance_pp: the pp has no position left
      const int32_t p = (int32_t)L[(cur * PHRASE_LANE_CAP + kc) * 64 + lane];
#pragma unroll
      for (int t = 0; t < SLOPPY_LANE_TERMS; ++t) { nx[t] = cur == t ? kc + 1 : nx[t]; pos[t] = cur == t ? p : pos[t]; }
      end = p > end ? p : end;
      if (p > next) {  // done minimizing current match-length
        if (match_length <= slop) sfreq += 1.0f / ((float)match_length + 1.0f);  // compute_slop_factor (bm25_similarity.rs:65-67)
        least(cur, cur_pos, next);
        match_length = end - cur_pos;
      } else {
        const int32_t ml2 = end - p;
        match_length = ml2 < match_length ? ml2 : match_length;
      }
    }
    if (match_length <= slop) sfreq += 1.0f / ((float)match_length + 1.0f);
  }
  uint64_t key = 0ull;
  if (live && sfreq > 1.1920929e-07f) {  // matches(): sloppy_freq > f32::EPSILON (:1041-1045)
    const DevTerm T0 = terms[Q.first_term];
    const float* table = seg.sim_tables + (size_t)T0.sim_table * 257;
    const float k1 = table[256];
    float nrm = k1;
    if (seg.norms != nullptr) {
      const uint32_t nb = seg.norms[doc];
      nrm = table[seg.n_norm_ranks > 0 ? (uint32_t)seg.rank_to_norm[nb] : nb];
    }
    key = make_key(bm25_score(T0.weight * (k1 + 1.0f), sfreq, nrm), doc);
  }
  lanes_finish(again, key, slot, keys_out, redo, PHRASE_REDO_SLOPPY_LANES, redo_list, redo_cap, redo_n, lane);
}

// ---- sloppy phrases that REPEAT a term, 64 candidates per wavefront ---------------------------------------------------------------
// A phrase like [x, x] made k_sloppy_match — one candidate per wavefront, lane i = PhrasePositions i — the whole cost of a batch:
// every doc of x is a candidate, two lanes of 64 do the work, 78 ms of the 101 ms that 1024 two-term slop-2 phrases took
// (~30 of them name one term twice). Here every LANE runs the scorer for its own candidate, the repeats machinery included:
// advance_repeat_groups, collide / lesser / advance_rpts, and the priority queue as the ARRAY Rust's BinaryHeap keeps
// (push = sift_up, pop = swap-in the last, sift_down_to_bottom, sift_up: util/external/binary_heap.rs:121-210) — advance_rpts
// changes the keys of queued PhrasePositions behind the heap's back, so which element a pop hands out depends on that array
// (phrase_scorer.rs:651-701). The lane's state — position, next index, freq of each pp, the heap array — sits in its own LDS
// column (16-bit cells: every index is per-lane, so registers would be select chains), the phrase's constants (offsets, groups:
// SloppyGroups from k_sloppy_groups) in a small table the lanes share. pp i is the phrase's i-th term in QUERY order (the
// PhrasePositions' ord); statement for statement phrase_scorer.rs:537-790 (what the tests compare it with restates the same lines).
constexpr int SLOPPY_RPT_GROUPS = SLOPPY_LANE_TERMS / 2;  // repetition groups a phrase of <= 6 terms can have
__global__ __launch_bounds__(WG_THREADS) void k_sloppy_rpt_lanes(SegView seg, const DevQuery* __restrict__ queries,
                                                                 const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                                 const int64_t* __restrict__ emit_prefix,
                                                                 const unsigned long long* __restrict__ emit_count,
                                                                 const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops,
                                                                 const SloppyGroups* __restrict__ groups, int n_queries,
                                                                 int64_t n_groups, int64_t pos_len, uint64_t* __restrict__ keys_out, int* redo,
                                                                 int64_t* __restrict__ redo_list, int redo_cap, int* redo_n) {
  constexpr int NT = SLOPPY_LANE_TERMS;
  __shared__ __attribute__((aligned(16))) int32_t areas[WG_WAVES][384];
  __shared__ int16_t lists[WG_WAVES][NT][PHRASE_LANE_CAP * 64];
  __shared__ int16_t states[WG_WAVES][4 * NT][64];                 // rows: position, next index, freq, heap array — one column per lane
  __shared__ int16_t tables[WG_WAVES][3 * NT + SLOPPY_RPT_GROUPS * (NT + 1)];  // offset, group, index in group; per group: length, members
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t group = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (group >= n_groups) return;
  const int64_t slot = group * 64 + lane;
  const int q = upper_slot_wave(emit_prefix, n_queries, group * 64, lane);
  const int slop = slops[q];
  if (slop <= 0) return;
  const DevQuery Q = queries[q];
  const int n = Q.n_terms;
  if (n < 2 || n > NT) return;  // (k_sloppy_match_lanes handed those on itself)
  bool repeats = false;
  for (int c = 0; c < n; ++c) { const PosTerm P = pterms[Q.first_term + c]; repeats = repeats || P.same_as != P.query_ord; }
  if (!repeats) return;         // k_sloppy_match_lanes' phrase
  const int64_t idx = slot - emit_prefix[q];
  const int64_t cnt = (int64_t)emit_count[q];
  if (idx - lane >= cnt) return;
  const int32_t doc = idx < cnt ? emit_docs[slot] : -1;
  const bool act = doc >= 0;  // (a deleted doc: an approximation that is never checked)
  bool again = false;
  // ---- the phrase's constants, by pp (query order)
  int16_t* const T16 = tables[wave];
  int16_t* const OFFT = T16;
  int16_t* const GRPT = T16 + NT;
  int16_t* const INDT = T16 + 2 * NT;
  int16_t* const GLEN = T16 + 3 * NT;                       // [g]
  int16_t* const GMEM = T16 + 3 * NT + SLOPPY_RPT_GROUPS;   // [g][k]: the pp with rpt_group g, rpt_ind k
  const SloppyGroups* const G = groups + q;  // (read in place: a local copy indexed by a loop variable lives in scratch)
  if (lane < SLOPPY_RPT_GROUPS * (NT + 1)) T16[3 * NT + lane] = lane < SLOPPY_RPT_GROUPS ? 0 : -1;
  if (lane < NT) { OFFT[lane] = 0; GRPT[lane] = -1; INDT[lane] = 0; }
  wave_sync();
  bool fits = true;  // (groups beyond what the table holds cannot come from <= 6 terms; a corrupt SloppyGroups hands the phrase on)
  for (int c = 0; c < n; ++c) {
    const PosTerm P = pterms[Q.first_term + c];
    const int i = P.query_ord;
    const int g = (i >= 0 && i < SLOPPY_MAX_TERMS) ? (int)G->grp[i] : -1, k = (i >= 0 && i < SLOPPY_MAX_TERMS) ? (int)G->ind[i] : 0;
    if (i < 0 || i >= NT || g >= SLOPPY_RPT_GROUPS || (g >= 0 && (k < 0 || k >= NT)) || P.phrase_pos > 0x7fff || P.phrase_pos < -0x8000) { fits = false; continue; }
    if (lane == 0) {
      OFFT[i] = (int16_t)P.phrase_pos;
      GRPT[i] = (int16_t)g;
      INDT[i] = (int16_t)k;
      if (g >= 0) GMEM[g * NT + k] = (int16_t)i;
    }
  }
  wave_sync();
  if (lane < SLOPPY_RPT_GROUPS) {  // a group's members are numbered 0 .. len - 1
    int len = 0;
    while (len < NT && GMEM[lane * NT + len] >= 0) ++len;
    GLEN[lane] = (int16_t)len;
  }
  wave_sync();
  if (!fits) again = act;
  // ---- every pp's positions in the lane's doc: lists[pp], freq in the lane's state column
  int16_t* const S = &states[wave][0][0];
  auto POS = [&](int i) -> int16_t& { return S[(0 * NT + i) * 64 + lane]; };
  auto NXT = [&](int i) -> int16_t& { return S[(1 * NT + i) * 64 + lane]; };
  auto FRQ = [&](int i) -> int16_t& { return S[(2 * NT + i) * 64 + lane]; };
  auto HEAP = [&](int j) -> int16_t& { return S[(3 * NT + j) * 64 + lane]; };
  for (int c = 0; fits && c < n; ++c) {
    if (!__ballot(act && !again)) break;
    const DevTerm T = terms[Q.first_term + c];
    const PosTerm P = pterms[Q.first_term + c];
    const int f = lanes_doc_positions<int16_t>(seg, T, P, doc, act, again, pos_len, areas[wave], lists[wave][P.query_ord], lane);
    FRQ(P.query_ord) = (int16_t)f;
  }
  wave_sync();
  const bool live = act && !again;
  float sfreq = 0.0f;
  if (live) {
    const int16_t* const L = &lists[wave][0][0];
    int32_t end = (int32_t)0x80000000;
    int hn = 0;
    auto next_position = [&](int i) -> bool {  // PhrasePositions::next_position (:363-373)
      const int a = NXT(i);
      if (a >= (int)FRQ(i)) return false;
      POS(i) = L[(i * PHRASE_LANE_CAP + a) * 64 + lane];
      NXT(i) = (int16_t)(a + 1);
      return true;
    };
    auto advance_pp = [&](int i) -> bool {  // :638-646
      if (!next_position(i)) return false;
      const int32_t p = POS(i);
      end = p > end ? p : end;
      return true;
    };
    auto key_less = [&](int a, int b) -> bool {  // PPElement: (position, offset, ord)
      const int32_t pa = POS(a), pb = POS(b);
      if (pa != pb) return pa < pb;
      const int32_t oa = OFFT[a], ob = OFFT[b];
      if (oa != ob) return oa < ob;
      return a < b;
    };
    auto sift_up = [&](int start, int pos) {
      const int elt = HEAP(pos);
      while (pos > start) {
        const int parent = (pos - 1) / 2;
        const int pe = HEAP(parent);
        if (!key_less(elt, pe)) break;  // elt <= parent
        HEAP(pos) = (int16_t)pe;
        pos = parent;
      }
      HEAP(pos) = (int16_t)elt;
    };
    auto heap_push = [&](int i) {
      HEAP(hn) = (int16_t)i;
      ++hn;
      sift_up(0, hn - 1);
    };
    auto heap_pop = [&]() -> int {
      --hn;
      int item = HEAP(hn);
      if (hn > 0) {
        const int root = HEAP(0);
        HEAP(0) = (int16_t)item;
        item = root;
        int pos = 0;
        const int elt = HEAP(0);
        int child = 1;
        while (child < hn) {
          const int right = child + 1;
          if (right < hn && !key_less(HEAP(child), HEAP(right))) child = right;  // the greater of the two children
          HEAP(pos) = HEAP(child);
          pos = child;
          child = 2 * pos + 1;
        }
        HEAP(pos) = (int16_t)elt;
        sift_up(0, pos);
      }
      return item;
    };
    auto tp_pos = [&](int i) -> int32_t { return (int32_t)POS(i) + (int32_t)OFFT[i]; };
    auto collide = [&](int i) -> int {  // :716-726
      const int g = GRPT[i];
      const int len = GLEN[g];
      const int32_t tp = tp_pos(i);
      for (int k = 0; k < len; ++k) {
        const int j = GMEM[g * NT + k];
        if (j != i && tp_pos(j) == tp) return k;
      }
      return -1;
    };
    auto lesser = [&](int a, int b) -> int {  // :704-713
      const int32_t pa = POS(a), pb = POS(b);
      return (pa < pb || (pa == pb && OFFT[a] < OFFT[b])) ? a : b;
    };
    auto advance_rpts = [&](int pp) -> bool {  // :651-701
      const int g = GRPT[pp];
      if (g < 0) return true;  // not a repeater
      const int len = GLEN[g];
      uint32_t bits = 0;
      const int k0 = INDT[pp];
      int cur = pp;
      while (true) {
        const int k = collide(cur);
        if (k < 0) break;
        cur = lesser(cur, GMEM[g * NT + k]);  // always advance the lesser of the (only) two colliding pps
        if (!advance_pp(cur)) return false;
        if (k != k0) bits |= 1u << k;  // mark only those currently in the queue
      }
      // collisions resolved, now re-queue: pop until every marked pp has come out, then push them all back
      uint32_t stack = 0;  // three bits per entry
      int ns = 0;
      while (bits && hn > 0) {  // (the reference would panic on an empty queue; it cannot get there: a marked pp is in the queue)
        const int p2 = heap_pop();
        stack |= (uint32_t)p2 << (3 * ns);
        ++ns;
        const int g2 = GRPT[p2], k2 = INDT[p2];
        if (g2 >= 0 && k2 < len && ((bits >> k2) & 1u)) bits &= ~(1u << k2);
      }
      for (int i = 0; i < ns; ++i) heap_push((int)((stack >> (3 * (ns - 1 - i))) & 7u));
      return true;
    };
    // ---- init_complex (:620-627): place_first_positions, advance_repeat_groups (single-term arm: the j-th pp of a group
    // advances j times), fill_queue
    bool alive = true;
    for (int i = 0; i < n; ++i) { NXT(i) = 0; (void)next_position(i); }
    for (int i = 0; i < n && alive; ++i) {
      if (GRPT[i] < 0) continue;
      const int times = INDT[i];
      for (int t = 0; t < times && alive; ++t) alive = next_position(i);
    }
    if (alive) {
      for (int i = 0; i < n; ++i) {
        const int32_t p = POS(i);
        end = p > end ? p : end;
        heap_push(i);
      }
      // ---- phrase_freq (:537-577)
      int pp = heap_pop();
      int32_t match_length = end - (int32_t)POS(pp);
      int32_t next = POS(HEAP(0));
      while (advance_pp(pp)) {
        if (!advance_rpts(pp)) break;  // pps exhausted
        const int32_t p = POS(pp);
        if (p > next) {  // done minimizing current match-length
          if (match_length <= slop) sfreq += 1.0f / ((float)match_length + 1.0f);  // compute_slop_factor (bm25_similarity.rs:65-67)
          heap_push(pp);
          pp = heap_pop();
          next = POS(HEAP(0));
          match_length = end - (int32_t)POS(pp);
        } else {
          const int32_t ml2 = end - p;
          match_length = ml2 < match_length ? ml2 : match_length;
        }
      }
      if (match_length <= slop) sfreq += 1.0f / ((float)match_length + 1.0f);
    }
  }
  uint64_t key = 0ull;
  if (live && sfreq > 1.1920929e-07f) {  // matches(): sloppy_freq > f32::EPSILON (:1041-1045)
    const DevTerm T0 = terms[Q.first_term];
    const float* table = seg.sim_tables + (size_t)T0.sim_table * 257;
    const float k1 = table[256];
    float nrm = k1;
    if (seg.norms != nullptr) {
      const uint32_t nb = seg.norms[doc];
      nrm = table[seg.n_norm_ranks > 0 ? (uint32_t)seg.rank_to_norm[nb] : nb];
    }
    key = make_key(bm25_score(T0.weight * (k1 + 1.0f), sfreq, nrm), doc);
  }
  lanes_finish(again, key, slot, keys_out, redo, PHRASE_REDO_SLOPPY_LANES, redo_list, redo_cap, redo_n, lane);
}

// TopDocsCollector over one query's candidates: a key of 0 = "phrase freq 0" (not a hit). One wavefront per query.
// Any k up to RGPU_MAX_K (collector/top_docs.rs:28-95): a wavefront's registers hold 128 keys, so k > 128 runs as passes of
// 128 over the candidates' keys — pass p keeps what lies strictly below pass p - 1's worst key (keys are unique per doc and
// totally ordered: the passes partition the ranking exactly, as in k_merge_lists). WIDE: k > 64.
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_collect(const int64_t* __restrict__ emit_prefix,
                                                               const unsigned long long* __restrict__ emit_count,
                                                               const uint64_t* __restrict__ keys, const int32_t* __restrict__ emit_docs,
                                                               const int32_t* __restrict__ slops, const int32_t* __restrict__ next_limits,
                                                               int n_queries, int k, int32_t doc_base,
                                                               HitOut* __restrict__ hits_out, int64_t* __restrict__ totals_out) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  int64_t total = 0;
  const int64_t base = emit_prefix[q], n = (int64_t)emit_count[q];
  HitOut* out = hits_out + (size_t)q * (size_t)k;
  // SloppyPhraseScorer is two-phase, so BulkScorer runs it through its two-phase loop (bulk_scorer.rs:97-113): every
  // approximation — a conjunction match, phrase or not, live or deleted — counts, and once more than next_limit of them went by
  // with nothing collected the leaf is abandoned: the query then has NO hits in this segment. In terms of the candidate list:
  // the first collected doc F is the smallest doc with a key, and it is reached iff at most next_limit candidates precede it.
  if (slops[q] > 0 && next_limits[q] >= 0) {
    int32_t first = 0x7fffffff;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
      const bool hit = i0 + lane < n && keys[base + i0 + lane] != 0ull;
      if (hit) first = min(first, emit_docs[base + i0 + lane]);
    }
    first = 0x7fffffff - (int32_t)wave_reduce_max_u32((uint32_t)(0x7fffffff - first));  // min over the lanes (doc ids are >= 0)
    int64_t before = 0;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
      const bool earlier = i0 + lane < n && (emit_docs[base + i0 + lane] & 0x7fffffff) < first;
      before += __popcll(__ballot(earlier));
    }
    if (first == 0x7fffffff || before > (int64_t)next_limits[q]) {
      for (int i = lane; i < k; i += 64) out[i] = HitOut{-1, 0.f};
      if (lane == 0) totals_out[q] = 0;
      return;
    }
  }
  uint64_t ceil = ~0ull;
  for (int col0 = 0; col0 < k; col0 += 128) {
    const int kp = min(128, k - col0);
    WaveTopK top;
    uint64_t tau = 0;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
      const uint64_t raw = i0 + lane < n ? keys[base + i0 + lane] : 0ull;
      if (col0 == 0) total += __popcll(__ballot(raw != 0ull));
      const uint64_t key = below(raw, ceil);
      if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, kp, lane);
    }
    if (lane < kp) out[col0 + lane] = top.a ? HitOut{key_doc(top.a) + doc_base, key_score(top.a)} : HitOut{-1, 0.f};
    if (WIDE && lane + 64 < kp) out[col0 + lane + 64] = top.b ? HitOut{key_doc(top.b) + doc_base, key_score(top.b)} : HitOut{-1, 0.f};
    ceil = topk_threshold<WIDE>(top, kp);  // 0 when this pass did not fill up: nothing is left for the next one
  }
  if (lane == 0) totals_out[q] = total;
}

// The same collector for k <= 128 with a query's candidates cut into chunks — one wavefront per (query, PHRASE_COLLECT_CHUNK
// candidates), partial lists folded by k_merge_items: one wavefront walking the 2 M keys of a common pair of terms was 12 of
// the 15 ms k_phrase_collect took on the benchmark batch. Items are planned by the host from the lead's doc_freq (an upper bound
// of the candidates); a chunk past the query's candidate count leaves an empty list. k_phrase_cutoff_items / _decide apply the two-phase
// rule of the sloppy scorer (see k_phrase_collect) beforehand: abandoned[q] = 1 empties every chunk of the query.
constexpr int PHRASE_COLLECT_CHUNK = 8192;
// The two-phase rule's decision with the candidates cut into the collector's chunks (round 5: one wavefront per query walking the
// 2 M candidates of a common pair of terms twice was 7 ms of the 19 ms a batch of 1024 two-term slop-2 phrases took): PASS 0 — the smallest doc with a
// key, per query (an atomic min per chunk); PASS 1 — the candidates in front of it (an atomic add per chunk); then one lane per
// query decides. cut[q] = {first doc, smallest live candidate, candidates before the first doc}: {INT_MAX, INT_MAX, 0} from the host.
// PASS 2 — run in front of k_sloppy_groups — is the smallest live candidate of every sloppy query (that kernel used to look for it
// with one wavefront per query: 2 ms for a query with 2 M candidates).
template <int PASS>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_cutoff_items(const int64_t* __restrict__ item_prefix, const int64_t* __restrict__ emit_prefix,
                                                                    const unsigned long long* __restrict__ emit_count, const uint64_t* __restrict__ keys,
                                                                    const int32_t* __restrict__ emit_docs, const int32_t* __restrict__ slops,
                                                                    const int32_t* __restrict__ next_limits, int n_queries, int64_t n_items,
                                                                    PhraseCut* __restrict__ cut) {
  const int lane = lane_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave_id();
  if (item >= n_items) return;
  const int q = upper_slot_wave(item_prefix, n_queries, item, lane);
  if (PASS == 2) {  // the smallest live candidate of a sloppy query (deleted candidates carry the sign bit)
    if (slops[q] <= 0) return;
    const int64_t base2 = emit_prefix[q], n2 = (int64_t)emit_count[q];
    const int64_t lo2 = (item - item_prefix[q]) * PHRASE_COLLECT_CHUNK, hi2 = min(n2, lo2 + PHRASE_COLLECT_CHUNK);
    if (lo2 >= hi2) return;
    int32_t dmin = 0x7fffffff;
    for (int64_t i0 = lo2; i0 < hi2; i0 += 64 * 8) {
      int32_t dd[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dd[u] = emit_docs[base2 + min(i0 + 64 * u + lane, hi2 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) dmin = min(dmin, dd[u] < 0 ? 0x7fffffff : dd[u]);
    }
    dmin = 0x7fffffff - (int32_t)wave_reduce_max_u32((uint32_t)(0x7fffffff - dmin));
    if (lane == 0 && dmin != 0x7fffffff) atomicMin(&cut[q].dmin, dmin);
    return;
  }
  if (!(slops[q] > 0 && next_limits[q] >= 0)) return;
  const int64_t base = emit_prefix[q], n = (int64_t)emit_count[q];
  // Fewer candidates than the limit: the misses in front of the first match cannot exceed it, and a query without any match
  // collects nothing whether it is marked abandoned or not — nothing to decide (and no walk over the candidates).
  if (n <= (int64_t)next_limits[q]) return;
  const int64_t lo = (item - item_prefix[q]) * PHRASE_COLLECT_CHUNK, hi = min(n, lo + PHRASE_COLLECT_CHUNK);
  if (lo >= hi) return;
  if (PASS == 0) {
    int32_t first = 0x7fffffff;
    for (int64_t i0 = lo; i0 < hi; i0 += 64 * 8) {
      uint64_t kk[8];
      int32_t dd[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // (an index past the chunk reads its last candidate again and is masked out)
        const int64_t at = base + min(i0 + 64 * u + lane, hi - 1);
        kk[u] = keys[at];
        dd[u] = emit_docs[at];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) first = (i0 + 64 * u + lane < hi && kk[u] != 0ull) ? min(first, dd[u]) : first;
    }
    first = 0x7fffffff - (int32_t)wave_reduce_max_u32((uint32_t)(0x7fffffff - first));
    if (lane == 0 && first != 0x7fffffff) atomicMin(&cut[q].first, first);
  } else {
    const int32_t first = cut[q].first;
    unsigned long long before = 0;
    for (int64_t i0 = lo; i0 < hi; i0 += 64 * 8) {
      int32_t dd[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dd[u] = emit_docs[base + min(i0 + 64 * u + lane, hi - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) before += (unsigned long long)__popcll(__ballot(i0 + 64 * u + lane < hi && (dd[u] & 0x7fffffff) < first));
    }
    if (lane == 0 && before != 0ull) atomicAdd(&cut[q].before, before);
  }
}
__global__ void k_phrase_cutoff_decide(const unsigned long long* __restrict__ emit_count, const int32_t* __restrict__ slops,
                                       const int32_t* __restrict__ next_limits, int n_queries, const PhraseCut* __restrict__ cut,
                                       int32_t* __restrict__ abandoned) {
  const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (q >= n_queries || !(slops[q] > 0 && next_limits[q] >= 0) || (int64_t)emit_count[q] <= (int64_t)next_limits[q]) return;
  if (cut[q].first == 0x7fffffff || cut[q].before > (unsigned long long)next_limits[q]) abandoned[q] = 1;
}
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_phrase_collect_items(const int64_t* __restrict__ item_prefix, const int64_t* __restrict__ emit_prefix,
                                                                     const unsigned long long* __restrict__ emit_count, const uint64_t* __restrict__ keys,
                                                                     const int32_t* __restrict__ abandoned, int n_queries, int64_t n_items, int k,
                                                                     uint64_t* __restrict__ partial_keys, int32_t* __restrict__ partial_counts) {
  const int lane = lane_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave_id();
  if (item >= n_items) return;
  const int q = upper_slot_wave(item_prefix, n_queries, item, lane);
  const int64_t chunk = item - item_prefix[q];
  const int64_t base = emit_prefix[q], n = abandoned[q] ? 0 : (int64_t)emit_count[q];
  const int64_t lo = chunk * PHRASE_COLLECT_CHUNK, hi = min(n, lo + PHRASE_COLLECT_CHUNK);
  WaveTopK top;
  uint64_t tau = 0;
  int count = 0;
  for (int64_t i0 = lo; i0 < hi; i0 += 64) {
    const uint64_t key = i0 + lane < hi ? keys[base + i0 + lane] : 0ull;
    count += __popcll(__ballot(key != 0ull));
    if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane);
  }
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  if (lane == 0) partial_counts[item] = count;
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
