// Disjunctions of TEN OR MORE SHOULD clauses in one launch: k_or_wide.
//
// With >= 10 sub-scorers (and min_should_match <= 1) the reference's DisjunctionSumScorer keeps them in a
// DisiPriorityQueue and sums a doc's scores in heap order (search/scorer/disjunction_scorer.rs:41-45, 213-225;
// util/disi.rs): the order of the f32 additions depends on the heap's history, so the reference itself pins a doc's
// score only up to the rounding of a sum of non-negative terms (1e-5 relative). That freedom is what this
// kernel is built on — the additions happen in whatever order the wavefronts reach them, in FIXED POINT, so that the
// result does not depend on that order at all:
//   * a WORKGROUP (8 wavefronts) owns a window of `WS` doc ids as u32 accumulators in LDS (48 KB at the default 12288);
//   * every clause's blocks that overlap the window — FullBlocks and the prepared tail alike, found with one coalesced
//     look at the clause's block directory — form ONE flat list that is cut into eight equal pieces, one per
//     wavefront: no clause is "sparse" or "dense", nothing is materialised in HBM (no k_score_terms, no runs), and a
//     wavefront decodes 128 postings per step whatever the clause;
//   * a posting's score becomes max(1, round(score * 2^e)) — e per query, chosen by the host so that the sum of the
//     clauses' largest possible scores stays below 2^31 — and is added with ds_add_u32 (8 cycles per wavefront on gfx950;
//     ds_add_f32 was measured at 193: scripts/microbench/lds_atomics.hip). Integer sums are exact: a doc's total is the
//     same whatever the order, a touched doc is a non-zero cell, and TopDocs::total_hits stays exact. A hit's score is
//     the total scaled back and rounded to f32 once; it differs from any f32 summation order's by at most n/2 fixed-point
//     steps, i.e. by < 4e-6 relative whenever the total is >= n * 2^17 steps. k_merge_items flags the (pathological:
//     a top-k that reaches down to scores a thousand times smaller than the query's largest possible) queries that
//     return a smaller total, and the host runs those again through k_or_windows;
//   * the scan of a window (all 512 lanes, four docs per lane per step) counts the touched docs and offers the ones at or
//     above the threshold to the wavefront's top-k — keys are (total << 32 | ~doc), no float ordering tricks needed.
// Clauses < 10, MUST_NOT clauses, min_should_match > 1, deleted docs, raw norm bytes, negative or non-finite weights or
// similarity tables: k_or_windows (search_or.hpp), which sums f32 in clause order, bit-exact.
//
// Latency: everything a window needs is requested one or two windows ahead — block bounds two windows ahead (a chain of
// directory reads per clause, owned by wavefront c % 8), the list's directory words one window ahead (in flight during
// the scan), the first four blocks' payload rows before the barrier that ends the previous window.
#pragma once
#include "search_or.hpp"

namespace rgpu {

constexpr int ORX_WAVES = 8;
constexpr int ORX_THREADS = 64 * ORX_WAVES;
constexpr int ORX_TABLES = 4;       // clauses scored through an LDS score table (the longest lists); the rest use the formula
constexpr int ORX_MAX_TERMS = 16;   // == RGPU_MAX_QUERY_TERMS
constexpr int ORX_RING = 4;         // payload rows in flight per wavefront
constexpr int ORX_BOUNDS_RING = 3;  // bounds of windows n, n+1, n+2
constexpr int ORX_MAX_WINDOW = 126 * 128;  // a window's blocks of one clause must fit the 128 directory entries looked at
constexpr uint32_t ORX_FLOOR_PER_CLAUSE = 1u << 17;  // a returned total below n_clauses * this is summed again in f32 (see above)
#ifndef RGPU_ORX_ABL  // developer ablations (variant builds only; results are wrong)
#define RGPU_ORX_ABL 0
#endif

__host__ __device__ constexpr size_t orx_fixed_lds() {
  return (size_t)ORX_TABLES * WAVE_CACHE_FLOATS * 4 + (size_t)ORX_MAX_TERMS * 64 * 4 + (size_t)ORX_WAVES * 2 * SLAB_STREAM +
         (size_t)ORX_BOUNDS_RING * ORX_MAX_TERMS * 8;
}
__host__ __device__ constexpr size_t orx_lds_bytes(int WS) { return orx_fixed_lds() + (size_t)WS * 4 + 256; }  // + 64 spare cells

// one workgroup = (query, group of `windows_per_item` windows); workgroup b works on query b % n_queries (see
// k_or_windows: later workgroups start from the thresholds the earlier ones published). DevQuery::op bits 16.. = the
// clauses that get a score table, DevQuery::pad = the query's fixed-point exponent e. Every wavefront writes its own
// top-k list: item (q * items_per_query + g) * 8 + wave.
template <bool LEGACY, bool WIDE>
__global__ __launch_bounds__(ORX_THREADS, 4) void k_or_wide(SegView seg, const DevQuery* __restrict__ queries,
                                                            const DevTerm* __restrict__ terms, int n_queries,
                                                            int windows_per_query, int windows_per_item, int items_per_query,
                                                            int WS, int k, uint64_t* __restrict__ partial_keys,
                                                            int32_t* __restrict__ partial_counts,
                                                            unsigned long long* __restrict__ tau_slots) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = wave_id();
  float* tables = reinterpret_cast<float*>(smem);  // per table: 64 norm-cache floats, then 64 x 11 fixed-point scores (u32)
  float* caches = tables + ORX_TABLES * WAVE_CACHE_FLOATS;  // caches[c][rank] = the clause's norm cache by norm rank
  uint8_t* slab = reinterpret_cast<uint8_t*>(caches + ORX_MAX_TERMS * 64) + wave * 2 * SLAB_STREAM;
  int2* bounds = reinterpret_cast<int2*>(reinterpret_cast<uint8_t*>(caches + ORX_MAX_TERMS * 64) + ORX_WAVES * 2 * SLAB_STREAM);
  uint32_t* acc = reinterpret_cast<uint32_t*>(bounds + ORX_BOUNDS_RING * ORX_MAX_TERMS);

  const int q = (int)(blockIdx.x % (unsigned)n_queries);
  const int g = (int)(blockIdx.x / (unsigned)n_queries);
  const int64_t item = ((int64_t)q * items_per_query + g) * ORX_WAVES + wave;
  const DevQuery Q = queries[q];
  const int n = Q.n_terms;
  const uint32_t table_mask = ((uint32_t)Q.op >> 16) & 0xffffu;
  const float scale = ldexpf(1.0f, Q.pad);  // a power of two: score * scale is exact until it is rounded to an integer
  auto to_fixed = [&](float score) -> uint32_t {
    const uint32_t v = (uint32_t)rintf(score * scale);  // (the host's choice of e keeps every total below 2^31)
    return v > 1u ? v : 1u;                           // a touched cell is never zero
  };
  const int win0 = g * windows_per_item;
  const int win1 = min(windows_per_query, win0 + windows_per_item);
  const int32_t first_doc = win0 * WS;

  // ---- per-clause constants: lane c holds clause c's
  uint64_t c_bs = 0, c_pn = 0;
  uint32_t c_dir = 0;
  int32_t c_nb = 0, c_nbx = 0, c_tbl = -1, c_sdoc = -1;
  float c_wk = 0.f;
  uint32_t c_sfix = 0u;
  if (lane < n) {
    const DevTerm* T = terms + Q.first_term + lane;
    c_bs = T->bs_base; c_pn = T->pn_base; c_dir = T->dir_base; c_nb = T->nblocks;
    c_nbx = T->nblocks + (T->tail_n > 0 ? 1 : 0);  // the prepared tail is one more block (its directory slot holds its last doc)
    const float* sim = seg.sim_tables + (size_t)T->sim_table * 257;
    c_wk = T->weight * (sim[256] + 1.0f);
    if ((table_mask >> lane) & 1u) c_tbl = __popc(table_mask & ((1u << lane) - 1u));
    if (T->df == 1) {  // a singleton lives in the term dictionary entry: scored once, added by wavefront 0 in its window
      c_sdoc = T->singleton_doc;
      c_sfix = to_fixed(bm25_score(c_wk, (float)T->singleton_freq, sim[seg.rank_to_norm[norm_at(seg, c_sdoc)]]));
      if (!doc_in_segment(seg, c_sdoc)) c_sdoc = -1;
    }
  }
  // norm caches (wavefront w: clauses w and w + 8) and score tables (wavefront s: table s)
  for (int c = wave; c < n; c += ORX_WAVES)
    caches[c * 64 + lane] = seg.sim_tables[(size_t)terms[Q.first_term + c].sim_table * 257 + seg.rank_to_norm[lane]];
  {
    const int nt = __popc(table_mask);
    if (wave < nt) {
      uint32_t m = table_mask;
      for (int i = 0; i < wave; ++i) m &= m - 1;
      const DevTerm T = terms[Q.first_term + (int)__builtin_ctz(m)];
      float* tbl = tables + wave * WAVE_CACHE_FLOATS;
      float k1;
      load_sim_table(seg, T.sim_table, tbl, lane, k1);
      const float wk = T.weight * (k1 + 1.0f);
      const float nrm = tbl[lane];
      uint32_t* row = reinterpret_cast<uint32_t*>(tbl) + 64 + lane * SCORE_TABLE_COLS;  // lane r fills norm rank r's row
#pragma unroll
      for (int f = 0; f <= SCORE_TABLE_FREQS; ++f) row[f] = to_fixed(bm25_score(wk, (float)f, nrm));
    }
  }
  for (int i = (int)threadIdx.x; i < WS; i += ORX_THREADS) acc[i] = 0u;

  // ---- block bounds of a window, per clause: blocks [lo, lo + cnt) hold a doc of [w0, w1). The owner keeps a cursor
  // `cur` (a block at or before lo) and looks at the 128 directory entries from it.
  const int own_a = wave, own_b = wave + ORX_WAVES;
  int cur_a = 0, cur_b = 0;
  int32_t ea0 = 0, ea1 = 0, eb0 = 0, eb1 = 0;
  auto bounds_issue = [&](int c, int cur, int32_t& e0, int32_t& e1) {
    const uint32_t dir = (uint32_t)readlane((int)c_dir, c);
    const int nbx = readlane(c_nbx, c);
    const int p = cur + lane;
    e0 = p < nbx ? seg.dir_last[dir + p] : 0x7fffffff;
    e1 = p + 64 < nbx ? seg.dir_last[dir + p + 64] : 0x7fffffff;
  };
  auto bounds_finish = [&](int c, int& cur, int32_t e0, int32_t e1, int win) {
    const int nbx = readlane(c_nbx, c);
    const int32_t w0 = win * WS;
    const int32_t w1 = min(seg.max_doc, w0 + WS);
    int lo = cur + __popcll(__ballot(e0 < w0)) + __popcll(__ballot(e1 < w0));
    // block b > 0 holds docs in (dir_last[b-1], dir_last[b]]: it reaches into the window iff dir_last[b-1] <= w1 - 2
    int hi = min(nbx, cur + __popcll(__ballot(e0 <= w1 - 2)) + __popcll(__ballot(e1 <= w1 - 2)) + 1);
    if (win >= win1) { lo = 0; hi = 0; }
    if (lane == 0) bounds[(win % ORX_BOUNDS_RING) * ORX_MAX_TERMS + c] = make_int2(lo, max(0, hi - lo));
    if (win < win1) cur = max(cur, hi - 1);
  };
  if (own_a < n) cur_a = find_block_wave(seg.dir_last, (uint32_t)readlane((int)c_dir, own_a), 0, readlane(c_nbx, own_a), first_doc, lane);
  if (own_b < n) cur_b = find_block_wave(seg.dir_last, (uint32_t)readlane((int)c_dir, own_b), 0, readlane(c_nbx, own_b), first_doc, lane);
  for (int w = win0; w < win0 + 2; ++w) {  // the first two windows' bounds: the only exposed directory reads
    if (own_a < n) { bounds_issue(own_a, cur_a, ea0, ea1); bounds_finish(own_a, cur_a, ea0, ea1, w); }
    if (own_b < n) { bounds_issue(own_b, cur_b, eb0, eb1); bounds_finish(own_b, cur_b, eb0, eb1, w); }
  }
  if (own_a < n) bounds_issue(own_a, cur_a, ea0, ea1);  // for window win0 + 2
  if (own_b < n) bounds_issue(own_b, cur_b, eb0, eb1);
  __syncthreads();

  // ---- a wavefront's share of a window: entries [wave * per, wave * per + per) of the flat list of (clause, block)
  // pairs, `per` = ceil(total / 8); lane i of a List holds entry page + i with its directory words
  struct List {
    int c, b;
    uint32_t hdr, row;
    int32_t base;
    int n;      // entries held (<= 64), wave-uniform
    int mine;   // this wavefront's entries in the window (> 64: the rest goes through further pages)
  };
  auto build_list = [&](int win, int page) -> List {
    List L;
    const int2 bd = lane < n ? bounds[(win % ORX_BOUNDS_RING) * ORX_MAX_TERMS + lane] : make_int2(0, 0);
    const int incl = wave_incl_scan(bd.y);
    const int total = (RGPU_ORX_ABL == 4 || RGPU_ORX_ABL == 5) ? 0 : readlane(incl, 63);
    const int per = (total + ORX_WAVES - 1) / ORX_WAVES;
    const int j0 = wave * per;
    L.mine = max(0, min(total, j0 + per) - j0);
    L.n = max(0, min(64, L.mine - page));
    const int j = j0 + page + lane;
    int c = 0;
    for (int t = 0; t < n; ++t) c += j >= readlane(incl, t) ? 1 : 0;
    const bool valid = lane < L.n;
    c = valid ? c : 0;
    const int off = bd.x - (incl - bd.y);  // lane t: lo_t - (entries before clause t)
    L.c = c;
    const int lo_rel = __builtin_amdgcn_ds_bpermute(c << 2, off);  // all lanes active: a bpermute reads 0 from disabled lanes
    L.b = valid ? lo_rel + j : 0;  // (lanes past the list name clause 0's block 0: a safe address)
    const uint32_t gi = (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)c_dir) + (uint32_t)L.b;
    L.hdr = valid ? (uint32_t)seg.dir_hdr[gi] : 0u;
    L.row = valid ? seg.dir_row[gi] : 0u;
    L.base = (valid && L.b > 0) ? seg.dir_last[gi - 1] : 0;
    return L;
  };

  struct Slot {
    uint4 rows;   // a FullBlock's payload row of this lane — or, for a tail, {doc0, doc1, freq0, freq1}
    uint32_t nn;  // posting-order norm ranks of postings 2*lane, 2*lane+1
  };
  // The payload request of one list entry, given as scalars. Unconditional and always the same two loads, so that the
  // compiler's vmcnt bookkeeping keeps the whole ring in flight (a load behind a branch makes every later wait a
  // vmcnt(0)). A tail cell is 16 bytes per lane like a FullBlock row; its norms follow the FullBlocks' in pnorm.
  auto fetch_at = [&](int c, int b, uint32_t hdr, uint32_t row) -> Slot {
    Slot s;
    const uint8_t* rows0 = block_rows_at(seg.bstore + readlane64(c_bs, c), row);
    const bool full = b < readlane(c_nb, c);
    const uint32_t voff = full ? 16u * (uint32_t)(lane & 31) + __umul24(16u * (uint32_t)(lane >> 5), (uint32_t)store_doc_rows(hdr)) : 16u * (uint32_t)lane;
    s.rows = *reinterpret_cast<const uint4*>(rows0 + voff);
    s.nn = *reinterpret_cast<const uint16_t*>(seg.pnorm + readlane64(c_pn, c) + (128u * (uint32_t)b + 2u * (uint32_t)lane));
    return s;
  };
  auto fetch = [&](const List& L, int idx) -> Slot {
    return fetch_at(readlane(L.c, idx), readlane(L.b, idx), (uint32_t)readlane((int)L.hdr, idx), (uint32_t)readlane((int)L.row, idx));
  };
  // entry `ia` of list A or entry `ib` of list B (wave-uniform choice): scalar selects, one request
  auto fetch_either = [&](bool use_b, const List& A, int ia, const List& B, int ib) -> Slot {
    const int c = use_b ? readlane(B.c, ib) : readlane(A.c, ia);
    const int b = use_b ? readlane(B.b, ib) : readlane(A.b, ia);
    const uint32_t hdr = (uint32_t)(use_b ? readlane((int)B.hdr, ib) : readlane((int)A.hdr, ia));
    const uint32_t row = (uint32_t)(use_b ? readlane((int)B.row, ib) : readlane((int)A.row, ia));
    return fetch_at(c, b, hdr, row);
  };
  auto process = [&](const Slot& s, const List& L, int idx, int32_t w0, uint32_t wlen) {
    const int c = readlane(L.c, idx);
    const int b = readlane(L.b, idx);
    const uint32_t hdr = (uint32_t)readlane((int)L.hdr, idx);
    int32_t e0, e1;
    uint32_t f0, f1;
    const uint32_t nb0 = s.nn & 0xffu, nb1 = s.nn >> 8;
    bool small_freqs;
    if (RGPU_ORX_ABL == 8) {  // payload consumed, nothing decoded: one store keeps the loads alive
      if (lane == (int)(s.rows.x & 63u) && s.nn == 0x12345u) acc[lane] = __uint_as_float(s.rows.y);
      return;
    }
    if (RGPU_ORX_ABL == 7) {  // no unpacking: docs from the lane id
      e0 = readlane(L.base, idx) + 1 + 2 * lane; e1 = e0 + 1; f0 = s.rows.x & 7u; f1 = s.rows.y & 7u; small_freqs = true;
    } else
    if (b < readlane(c_nb, c)) {
      stage_rows(s.rows, slab, lane);
      wave_sync();
      uint32_t x0, x1;
      staged_doc_deltas<LEGACY>(slab, s.rows, hdr, lane, x0, x1);
      staged_freqs<LEGACY>(slab, s.rows, hdr, lane, f0, f1);
      wave_sync();  // slab is free for the next block
      deltas_to_docs(x0, x1, readlane(L.base, idx), e0, e1);
      small_freqs = hdr_bfreq(hdr) <= 3;
    } else {  // the tail: decoded and validated at prepare time; slots past its end hold doc INT_MAX, freq 0
      e0 = (int32_t)s.rows.x; e1 = (int32_t)s.rows.y; f0 = s.rows.z; f1 = s.rows.w;
      small_freqs = false;
    }
    const int slot = readlane(c_tbl, c);
    uint32_t s0, s1;
    if (slot >= 0 && (small_freqs || !__ballot((f0 > f1 ? f0 : f1) > (uint32_t)SCORE_TABLE_FREQS))) {
      const uint32_t* tbl = reinterpret_cast<const uint32_t*>(tables + slot * WAVE_CACHE_FLOATS) + 64;
      s0 = tbl[nb0 * SCORE_TABLE_COLS + f0];
      s1 = tbl[nb1 * SCORE_TABLE_COLS + f1];
    } else {
      const float* cache = caches + c * 64;
      const float wk = __int_as_float(readlane(__float_as_int(c_wk), c));
      s0 = to_fixed(bm25_score(wk, (float)(int32_t)f0, cache[nb0]));
      s1 = to_fixed(bm25_score(wk, (float)(int32_t)f1, cache[nb1]));
    }
    const uint32_t o0 = (uint32_t)(e0 - w0), o1 = (uint32_t)(e1 - w0);
    if (RGPU_ORX_ABL == 6) {  // plain stores instead of atomics
      if (o0 < wlen) acc[o0] = s0;
      if (o1 < wlen) acc[o1] = s1;
      return;
    }
    // postings outside the window (a block may reach into its neighbours) go to this lane's spare cell behind it:
    // one select instead of an exec-mask branch around the atomic
    const uint32_t spare = (uint32_t)(WS + lane);
    __hip_atomic_fetch_add(acc + (o0 < wlen ? o0 : spare), s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(acc + (o1 < wlen ? o1 : spare), s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int hits = 0;  // wave-uniform: touched docs this wavefront scanned
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);

  List cur_list = build_list(win0, 0);
  List next_list = build_list(win0 + 1, 0);
  Slot ring[ORX_RING];
#pragma unroll
  for (int j = 0; j < ORX_RING; ++j) ring[j] = Slot{make_uint4(0u, 0u, 0u, 0u), 0u};
#pragma unroll
  for (int j = 0; j < ORX_RING; ++j) ring[j] = fetch(cur_list, max(0, min(j, cur_list.n - 1)));
  __syncthreads();  // accumulators are cleared, tables and caches are built

  for (int win = win0; win < win1; ++win) {
    const int32_t w0 = win * WS;
    const uint32_t wlen = (uint32_t)(min(seg.max_doc, w0 + WS) - w0);
    const uint64_t seen = shared.peek();  // folded before the scan
    // ---- this wavefront's blocks; the ring ends up holding the next window's first blocks
    // Groups of ORX_RING blocks; after a block is done its ring slot requests the block ORX_RING further on — in the last
    // group that is block j of the NEXT window (static slot alignment). Entries past a list's end are clamped: a
    // redundant request instead of a branch.
    const int padded = max(ORX_RING, (cur_list.n + ORX_RING - 1) / ORX_RING * ORX_RING);
    for (int i = 0; i < padded; i += ORX_RING) {
      const bool last_group = i + ORX_RING >= padded;
#pragma unroll
      for (int j = 0; j < ORX_RING; ++j) {
        const int idx = i + j;
        if (idx < cur_list.n && RGPU_ORX_ABL != 2 && RGPU_ORX_ABL != 3) process(ring[j], cur_list, idx, w0, wlen);
        if (RGPU_ORX_ABL == 3) continue;
        ring[j] = fetch_either(last_group, cur_list, max(0, min(idx + ORX_RING, cur_list.n - 1)), next_list, max(0, min(j, next_list.n - 1)));
      }
    }
    for (int page = 64; page < cur_list.mine; page += 64) {  // > 64 blocks for one wavefront in one window: plain loop
      const List more = build_list(win, page);
      for (int idx = 0; idx < more.n; ++idx) process(fetch(more, idx), more, idx, w0, wlen);
    }
    if (wave == 0) {  // singletons (one lane per clause; two clauses may name the same doc: the add is atomic)
      const uint32_t o = (uint32_t)(c_sdoc - w0);
      if (c_sdoc >= 0 && o < wlen) __hip_atomic_fetch_add(acc + o, c_sfix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- bounds of window win + 2 from the directory entries requested one window ago; request the next ones
    if (RGPU_ORX_ABL != 5) {
      if (own_a < n) { bounds_finish(own_a, cur_a, ea0, ea1, win + 2); bounds_issue(own_a, cur_a, ea0, ea1); }
      if (own_b < n) { bounds_finish(own_b, cur_b, eb0, eb1, win + 2); bounds_issue(own_b, cur_b, eb0, eb1); }
    }
    __syncthreads();  // every add of this window has landed; bounds of win + 2 are visible
    const List after = build_list(win + 2, 0);  // its directory words arrive during the scan

    // ---- scan: four docs per lane per step; a touched accumulator is one collected hit (bulk_scorer.rs:114-120)
    shared.fold(seen, tau, floor);
    for (uint32_t i0 = 0; i0 < (uint32_t)WS && RGPU_ORX_ABL != 1; i0 += 4 * ORX_THREADS) {
      uint4* cell = reinterpret_cast<uint4*>(acc + i0 + 4 * threadIdx.x);
      const uint4 v = *cell;
      if (__ballot((v.x | v.y | v.z | v.w) != 0u)) {
        hits += __popcll(__ballot(v.x != 0u)) + __popcll(__ballot(v.y != 0u)) + __popcll(__ballot(v.z != 0u)) + __popcll(__ballot(v.w != 0u));
        *cell = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t thr = max(1u, (uint32_t)(tau >> 32));  // a key's high word is the doc's total
        const bool c0 = v.x >= thr, c1 = v.y >= thr, c2 = v.z >= thr, c3 = v.w >= thr;
        if (__ballot(c0 || c1 || c2 || c3)) {
          const uint32_t nd = ~(uint32_t)(w0 + (int32_t)i0 + 4 * (int32_t)threadIdx.x);  // ~doc: smaller doc id = larger key
          uint64_t key = c0 ? ((uint64_t)v.x << 32) | nd : 0ull;
          if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          key = c1 ? ((uint64_t)v.y << 32) | (nd - 1u) : 0ull;
          if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          key = c2 ? ((uint64_t)v.z << 32) | (nd - 2u) : 0ull;
          if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          key = c3 ? ((uint64_t)v.w << 32) | (nd - 3u) : 0ull;
          if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
        }
      }
    }
    shared.publish<WIDE>(top, k, lane);
    __syncthreads();  // the window is clear again
    cur_list = next_list;
    next_list = after;
  }
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  if (lane == 0) partial_counts[item] = hits;
}

}  // namespace rgpu
