// Disjunctions of TEN OR MORE SHOULD clauses in one launch: k_or_wide.
//
// With >= 10 sub-scorers (and min_should_match <= 1) the reference's DisjunctionSumScorer keeps them in a
// DisiPriorityQueue and sums a doc's scores in heap order (search/scorer/disjunction_scorer.rs:41-45, 213-225;
// util/disi.rs): the order of the f32 additions depends on the heap's history, so the reference itself pins a doc's
// score only up to the rounding of a sum of non-negative terms (1e-5 relative). That freedom is what this
// kernel is built on — the additions happen in whatever order the wavefronts reach them, in FIXED POINT, so that the
// result does not depend on that order at all:
//   * a WORKGROUP (8 wavefronts) owns a window of `WS` doc ids as u32 accumulators in LDS (64 KB at the default 16384);
//   * every clause's blocks that overlap the window — FullBlocks and the prepared tail alike, found with one coalesced
//     look at the clause's block directory — form ONE flat list that is dealt round-robin to the eight
//     wavefronts: no clause is "sparse" or "dense", nothing is materialised in HBM (no k_score_terms, no runs), and a
//     wavefront decodes 128 postings per step whatever the clause;
//   * a posting's score becomes max(1, round(score * 2^e)) — e per query, chosen by the host so that the sum of the
//     clauses' largest possible scores stays below 2^31 — and is added with ds_add_u32 (8 cycles per wavefront on gfx950;
//     ds_add_f32 was measured at 193: scripts/microbench/lds_atomics.hip). Integer sums are exact: a doc's total is the
//     same whatever the order, a touched doc is a non-zero cell, and TopDocs::total_hits stays exact. A hit's score is
//     the total scaled back and rounded to f32 once; it differs from any f32 summation order's by at most n/2 fixed-point
//     steps, i.e. by < 4e-6 relative whenever the total is >= n * 2^17 steps. k_merge_items flags the (pathological:
//     a top-k that reaches down to scores a thousand times smaller than the query's largest possible) queries that
//     return a smaller total, and the host runs those again through k_or_windows;
//   * the scan of a window (all 512 lanes, eight docs per lane per step) counts the touched docs and offers the ones at or
//     above the threshold to the wavefront's top-k — keys are (total << 32 | ~doc), no float ordering tricks needed;
//   * the eight lists of a workgroup share a bound on the query's k-th best: the smallest of their ceil(k/8)-th best totals
//     (every doc is scanned by exactly one wavefront, so k docs reach it).
// The kernel is bound by the NUMBER of instructions a wavefront issues per window (-DRGPU_ORX_TIME sums the
// wave-cycles of every phase of the window loop): anything added to the per-block path costs more than it looks.
// Clauses < 10, MUST_NOT clauses, min_should_match > 1, deleted docs, raw norm bytes, negative or non-finite weights or
// similarity tables: k_or_windows (search_or.hpp), which sums f32 in clause order, bit-exact.
//
// Latency: everything a window needs is requested one or two windows ahead — block bounds two windows ahead (a chain of
// directory reads per clause, owned by wavefront c % 8), the list's directory words one window ahead (in flight during
// the scan), the first four blocks' payload rows before the barrier that ends the previous window.
#pragma once
#include "search_or.hpp"

namespace rgpu {

#ifndef RGPU_ORX_WAVES
#define RGPU_ORX_WAVES 8
#endif
#ifndef RGPU_ORX_LOOK
#define RGPU_ORX_LOOK 3
#endif
constexpr int ORX_WAVES = RGPU_ORX_WAVES;  // 8: two workgroups per CU; 16: one, with twice the window
constexpr int ORX_THREADS = 64 * ORX_WAVES;
constexpr int ORX_OWN = (16 + ORX_WAVES - 1) / ORX_WAVES;  // clauses whose block bounds a wavefront keeps (c = wave + s * ORX_WAVES)
constexpr int ORX_LOOK = RGPU_ORX_LOOK;    // directory entries looked at per clause per window, in units of 64
constexpr int ORX_SCAN_STEP = 8 * ORX_THREADS;  // docs per scan step of the workgroup (two 16-byte reads per lane): windows are multiples of it
constexpr int ORX_WAVES_PER_SIMD = ORX_WAVES >= 16 ? 4 : (2 * ORX_WAVES + 3) / 4;  // two workgroups per CU (one of 16 wavefronts)
#ifndef RGPU_ORX_TABLES  // LDS is the budget (two workgroups per CU: 80 KB each): since the formula path takes a reciprocal instead of
#define RGPU_ORX_TABLES 1  // a division, a score table buys little (measured 2 / 4 / 6 tables: 9.95 / 9.88 / 10.0 ms at 12288-doc windows) and
#endif                     // 3 KB of window buy more — one table + 16384-doc windows: 8.9 ms

constexpr int ORX_TABLES = RGPU_ORX_TABLES;  // clauses scored through an LDS score table (the longest lists); the rest use the formula
constexpr int ORX_MAX_TERMS = 16;   // == RGPU_MAX_QUERY_TERMS
#ifndef RGPU_ORX_RING
#define RGPU_ORX_RING 4
#endif
constexpr int ORX_RING = RGPU_ORX_RING;  // payload rows in flight per wavefront
constexpr int ORX_BOUNDS_RING = 3;  // bounds of windows n, n+1, n+2
constexpr int ORX_MAX_WINDOW = (64 * ORX_LOOK - 2) * 128;  // a window's blocks of one clause must fit the directory entries looked at
constexpr uint32_t ORX_FLOOR_PER_CLAUSE = 1u << 17;  // a returned total below n_clauses * this is summed again in f32 (see above)
#ifndef RGPU_ORX_STRIDED
#define RGPU_ORX_STRIDED 1
#endif
#ifndef RGPU_ORX_ABL  // developer ablations (variant builds only; results are wrong)
#define RGPU_ORX_ABL 0
#endif

#ifdef RGPU_ORX_TIME  // developer instrumentation (variant builds only): wave-cycles per phase of the window loop
__device__ unsigned long long g_orx_dbg[8];  // [0] blocks [1] bounds [2] barrier 1 [3] scan loop [4] barrier 2 [5] list hand-over [6] list request + fold [7] publish
#define ORX_STAMP(t) const long long t = (long long)__builtin_readcyclecounter()
#define ORX_ADD(i, v) orx_t[i] += (v)
#else
#define ORX_STAMP(t) do {} while (0)
#define ORX_ADD(i, v) do {} while (0)
#endif

__host__ __device__ constexpr size_t orx_fixed_lds() {
  return (size_t)ORX_TABLES * WAVE_CACHE_FLOATS * 4 + (size_t)ORX_MAX_TERMS * 64 * 4 + (size_t)ORX_WAVES * 2 * SLAB_STREAM +
         (size_t)ORX_BOUNDS_RING * ORX_MAX_TERMS * 8 + (size_t)ORX_WAVES * 8;
}
__host__ __device__ constexpr size_t orx_lds_bytes(int WS) { return orx_fixed_lds() + (size_t)WS * 4 + 256; }  // + 64 spare cells

// one workgroup = (query, group of `windows_per_item` windows); workgroup b works on query b % n_queries (see
// k_or_windows: later workgroups start from the thresholds the earlier ones published). DevQuery::op bits 16.. = the
// clauses that get a score table, DevQuery::pad = the query's fixed-point exponent e. Every wavefront writes its own
// top-k list: item (q * items_per_query + g) * 8 + wave.
template <bool LEGACY, bool WIDE>
__global__ __launch_bounds__(ORX_THREADS, ORX_WAVES_PER_SIMD) void k_or_wide(SegView seg, const DevQuery* __restrict__ queries,
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
  uint32_t* wg_kth = reinterpret_cast<uint32_t*>(bounds + ORX_BOUNDS_RING * ORX_MAX_TERMS);  // per wavefront: its ceil(k/8)-th best total
  uint32_t* acc = wg_kth + 2 * ORX_WAVES;

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
  // `cur` (a block at or before lo) and looks at the 64 * ORX_LOOK directory entries from it.
  int own_cur[ORX_OWN];
  int32_t own_e[ORX_OWN][ORX_LOOK];
  auto bounds_issue = [&](int c, int cur, int32_t (&e)[ORX_LOOK]) {
    const uint32_t dir = (uint32_t)readlane((int)c_dir, c);
    const int nbx = readlane(c_nbx, c);
#pragma unroll
    for (int u = 0; u < ORX_LOOK; ++u) {
      const int p = cur + 64 * u + lane;
      e[u] = p < nbx ? seg.dir_last[dir + p] : 0x7fffffff;
    }
  };
  auto bounds_finish = [&](int c, int& cur, const int32_t (&e)[ORX_LOOK], int win) {
    const int nbx = readlane(c_nbx, c);
    const int32_t w0 = win * WS;
    const int32_t w1 = min(seg.max_doc, w0 + WS);
    int before = 0, reach = 0;
#pragma unroll
    for (int u = 0; u < ORX_LOOK; ++u) {
      before += __popcll(__ballot(e[u] < w0));
      // block b > 0 holds docs in (dir_last[b-1], dir_last[b]]: it reaches into the window iff dir_last[b-1] <= w1 - 2
      reach += __popcll(__ballot(e[u] <= w1 - 2));
    }
    int lo = cur + before;
    int hi = min(nbx, cur + reach + 1);
    if (win >= win1) { lo = 0; hi = 0; }
    if (lane == 0) bounds[(win % ORX_BOUNDS_RING) * ORX_MAX_TERMS + c] = make_int2(lo, max(0, hi - lo));
    if (win < win1) cur = max(cur, hi - 1);
  };
#pragma unroll
  for (int s = 0; s < ORX_OWN; ++s) {
    const int c = wave + s * ORX_WAVES;
    own_cur[s] = 0;
#pragma unroll
    for (int u = 0; u < ORX_LOOK; ++u) own_e[s][u] = 0;
    if (c < n) {
      own_cur[s] = find_block_wave(seg.dir_last, (uint32_t)readlane((int)c_dir, c), 0, readlane(c_nbx, c), first_doc, lane);
      for (int w = win0; w < win0 + 2; ++w) {  // the first two windows' bounds: the only exposed directory reads
        bounds_issue(c, own_cur[s], own_e[s]);
        bounds_finish(c, own_cur[s], own_e[s], w);
      }
      bounds_issue(c, own_cur[s], own_e[s]);  // for window win0 + 2
    }
  }
  __syncthreads();

  // ---- a wavefront's share of a window: entries [wave * per, wave * per + per) of the flat list of (clause, block)
  // pairs, `per` = ceil(total / 8). Lane i of a List holds entry page + i, everything a block needs as two addresses and
  // one word, so that the per-block code reads five lanes and does no address arithmetic:
  //   rows   address of the block's first payload row (or of the tail's cells)
  //   pn     address of the block's posting-order norm ranks
  //   meta   bits 0-15 the directory header word, 16-21 R = 16-byte rows between the doc and the freq payload (32 for a
  //          tail: then lane l's row is simply cell l), 22 FullBlock?, 23-26 clause, 27-30 score table + 1 (0: none)
  //   base   the doc id before the block's first posting
  // A page holds at most ORX_PAGE entries: the lanes behind them take the first ORX_RING entries of the NEXT window's list
  // (append_next), so the ring's look-ahead never has to choose between two lists.
  constexpr int ORX_PAGE = 64 - ORX_RING;
  struct List {
    uint64_t rows, pn;
    uint32_t meta;
    int32_t base;
    int n;      // entries held (<= ORX_PAGE), wave-uniform
    int mine;   // this wavefront's entries in the window (> ORX_PAGE: the rest goes through further pages)
  };
  struct Pending {  // a list whose directory words are still in flight
    List L;
    uint32_t hdr, row;
    int c;
    bool full, first;
  };
  auto build_list = [&](int win, int page) -> Pending {
    Pending P;
    const int2 bd = lane < n ? bounds[(win % ORX_BOUNDS_RING) * ORX_MAX_TERMS + lane] : make_int2(0, 0);
    const int incl = wave_incl_scan(bd.y);
    const int total = (RGPU_ORX_ABL == 4 || RGPU_ORX_ABL == 5) ? 0 : readlane(incl, 63);
#if RGPU_ORX_STRIDED
    // entry j belongs to wavefront j % 8: every wavefront gets a mix of clauses (table and formula blocks, FullBlocks
    // and tails) instead of a run of one clause's blocks, and their loads differ by at most one block
    P.L.mine = total > wave ? (total - wave + ORX_WAVES - 1) / ORX_WAVES : 0;
    P.L.n = max(0, min(ORX_PAGE, P.L.mine - page));
    const int j = wave + ORX_WAVES * (page + lane);
#else
    const int per = (total + ORX_WAVES - 1) / ORX_WAVES;
    const int j0 = wave * per;
    P.L.mine = max(0, min(total, j0 + per) - j0);
    P.L.n = max(0, min(ORX_PAGE, P.L.mine - page));
    const int j = j0 + page + lane;
#endif
    int c = 0;  // the clause of entry j = the number of clauses whose entries end at or before j
#pragma unroll
    for (int t = 0; t < ORX_MAX_TERMS - 1; ++t) c += j >= readlane(incl, t) ? 1 : 0;  // (lanes >= n hold the total: no entry reaches it; straight-line code beats a loop over n here)
    const bool valid = lane < P.L.n;
    c = valid ? c : 0;
    const int off = bd.x - (incl - bd.y);  // lane t: lo_t - (entries before clause t)
    // (bpermutes with every lane active: they read 0 from disabled lanes)
    const int lo_rel = __builtin_amdgcn_ds_bpermute(c << 2, off);
    const int b = valid ? lo_rel + j : 0;  // lanes past the list name clause 0's block 0: a safe address
    const uint32_t gi = (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)c_dir) + (uint32_t)b;
    const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)(uint32_t)(c_bs >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)(uint32_t)c_bs);
    const uint64_t pnb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)(uint32_t)(c_pn >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)(uint32_t)c_pn);
    P.full = b < __builtin_amdgcn_ds_bpermute(c << 2, c_nb);
    P.c = c | ((__builtin_amdgcn_ds_bpermute(c << 2, c_tbl) + 1) << 4);
    P.L.rows = (uint64_t)(uintptr_t)seg.bstore + bs;
    P.L.pn = (uint64_t)(uintptr_t)seg.pnorm + pnb + 128u * (uint64_t)(uint32_t)b;
    // unconditional loads (lanes past the list read clause 0's first slot): a load under a lane mask becomes a branch
    // and the compiler then waits for it on the spot, in front of the scan
    P.hdr = (uint32_t)seg.dir_hdr[gi];
    P.row = seg.dir_row[gi];
    P.L.base = seg.dir_last[gi - (b > 0 ? 1u : 0u)];  // (b == 0: not used as a base, replaced by 0 in finish_list)
    P.first = b == 0;
    P.L.meta = 0u;
    return P;
  };
  auto finish_list = [&](const Pending& P) -> List {  // once the directory words are here
    List L = P.L;
    L.rows += 16ull * (uint64_t)P.row;
    if (P.first) L.base = 0;
    const uint32_t R = P.full ? (uint32_t)store_doc_rows(P.hdr) : 32u;
    L.meta = (P.hdr & 0xffffu) | (R << 16) | (P.full ? 1u << 22 : 0u) | ((uint32_t)P.c << 23);
    return L;
  };
  // lanes [A.n, A.n + ORX_RING) of A take entries 0 .. ORX_RING-1 of B (clamped to B's last entry; B empty: its lane 0
  // names a safe address)
  auto append_next = [&](List& A, const List& B) {
    const int src = max(0, min(lane - A.n, B.n - 1));
    const bool take = lane >= A.n && lane < A.n + ORX_RING;
    const uint32_t rl = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)B.rows);
    const uint32_t rh = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(B.rows >> 32));
    const uint32_t pl = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)B.pn);
    const uint32_t ph = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(B.pn >> 32));
    const uint32_t mt = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)B.meta);
    const int32_t bs = __builtin_amdgcn_ds_bpermute(src << 2, B.base);
    if (take) { A.rows = ((uint64_t)rh << 32) | rl; A.pn = ((uint64_t)ph << 32) | pl; A.meta = mt; A.base = bs; }
  };

  struct Slot {
    uint4 rows;   // a FullBlock's payload row of this lane — or, for a tail, {doc0, doc1, freq0, freq1}
    uint32_t nn;  // posting-order norm ranks of postings 2*lane, 2*lane+1
  };
  // The payload request of one list entry: unconditional and always the same two loads, so that the compiler's vmcnt
  // bookkeeping keeps the whole ring in flight (a load behind a branch makes every later wait a vmcnt(0)).
  const uint32_t voff_lo = 16u * (uint32_t)(lane & 31), voff_hi = 16u * (uint32_t)(lane >> 5);
  auto fetch = [&](const List& L, int idx) -> Slot {
    Slot s;
    // addresses rebuilt from integers: say "global" explicitly, or the loads become FLAT ones (which also count as LDS
    // operations: every lgkmcnt wait of the unpacking code would then wait for the ring's requests)
    typedef const __attribute__((address_space(1))) uint8_t* gbytes;
    gbytes rows0 = (gbytes)(uintptr_t)readlane64(L.rows, idx);
    gbytes pn = (gbytes)(uintptr_t)readlane64(L.pn, idx);
    const uint32_t R = ((uint32_t)readlane((int)L.meta, idx) >> 16) & 63u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 r = *(const __attribute__((address_space(1))) u32x4*)(rows0 + (voff_lo + __umul24(voff_hi, R)));
    s.rows = make_uint4(r.x, r.y, r.z, r.w);
    s.nn = *(const __attribute__((address_space(1))) uint16_t*)(pn + 2u * (uint32_t)lane);
    return s;
  };
  const uint32_t lane_r = (uint32_t)(lane >> 1);                 // BP128: the row of this lane's pair of values
  const uint8_t* const slab_lane = slab + ((lane & 1) << 3);     // ... and its pair of streams inside a 16-byte stream word
  const uint32_t spare = (uint32_t)(WS + lane);                  // this lane's spare accumulator cell
  auto process = [&](const Slot& s, const List& L, int idx, int32_t w0, uint32_t wlen) {
    const uint32_t meta = (uint32_t)readlane((int)L.meta, idx);
    const uint32_t hdr = meta & 0xffffu;
    const int c = (int)((meta >> 23) & 15u);
    const int slot = (int)((meta >> 27) & 15u) - 1;
    int32_t e0, e1;
    uint32_t f0, f1;
    const uint32_t nb0 = s.nn & 0xffu, nb1 = s.nn >> 8;
    bool small_freqs;
    if (RGPU_ORX_ABL == 8) {  // payload consumed, nothing decoded: one store keeps the loads alive
      if (lane == (int)(s.rows.x & 63u) && s.nn == 0x12345u) acc[lane] = s.rows.y;
      return;
    }
    if (meta & (1u << 22)) {
      stage_rows(s.rows, slab, lane);
      wave_sync();
      uint32_t x0, x1;
      if constexpr (LEGACY) {
        staged_doc_deltas<LEGACY>(slab, s.rows, hdr, lane, x0, x1);
        staged_freqs<LEGACY>(slab, s.rows, hdr, lane, f0, f1);
      } else {
        // both streams in one go: the two LDS reads go out back to back and their round trips overlap. An all-equal
        // stream (b == 0: one row holding the value) reads row 0 like any other and is overwritten afterwards.
        const int bd = hdr_bdoc(hdr), bf = hdr_bfreq(hdr);
        const uint32_t pd = __umul24(lane_r, (uint32_t)bd), pf = __umul24(lane_r, (uint32_t)bf);
        const uint8_t* ad = slab_lane + (__builtin_amdgcn_ubfe(pd, 5, 6) << 4);  // 16 * (p / 32): the row of stream word p / 32
        const uint8_t* af = slab_lane + SLAB_STREAM + (__builtin_amdgcn_ubfe(pf, 5, 6) << 4);
        const uint2 dlo = lds_u2(ad), dhi = lds_u2(ad + 16);
        const uint2 flo = lds_u2(af), fhi = lds_u2(af + 16);
        const uint32_t md = 0xffffffffu >> ((32 - bd) & 31), mf = 0xffffffffu >> ((32 - bf) & 31);  // (b == 0: see below)
        x0 = (uint32_t)((((uint64_t)dhi.x << 32) | dlo.x) >> (pd & 31)) & md;
        x1 = (uint32_t)((((uint64_t)dhi.y << 32) | dlo.y) >> (pd & 31)) & md;
        f0 = (uint32_t)((((uint64_t)fhi.x << 32) | flo.x) >> (pf & 31)) & mf;
        f1 = (uint32_t)((((uint64_t)fhi.y << 32) | flo.y) >> (pf & 31)) & mf;
        if (bd == 0) x0 = x1 = (uint32_t)readlane((int)s.rows.x, 0);
        if (bf == 0) f0 = f1 = (uint32_t)readlane((int)s.rows.x, 32);
      }
      wave_sync();  // slab is free for the next block
      {  // deltas -> doc ids: with the inclusive scan of the pair sums, doc1 = base + scan and doc0 = doc1 - delta1
        const int incl = wave_incl_scan((int)(x0 + x1));
        e1 = readlane(L.base, idx) + incl;
        e0 = e1 - (int32_t)x1;
      }
      small_freqs = hdr_bfreq(hdr) <= 3;
    } else {  // the tail: decoded and validated at prepare time; slots past its end hold doc INT_MAX, freq 0
      e0 = (int32_t)s.rows.x; e1 = (int32_t)s.rows.y; f0 = s.rows.z; f1 = s.rows.w;
      small_freqs = false;
    }
    uint32_t s0, s1;
    if (slot >= 0 && (small_freqs || !__ballot((f0 > f1 ? f0 : f1) > (uint32_t)SCORE_TABLE_FREQS))) {
      const uint8_t* tbl = reinterpret_cast<const uint8_t*>(tables + slot * WAVE_CACHE_FLOATS + 64);
      s0 = lds_u32(tbl + __umul24(nb0, 4u * SCORE_TABLE_COLS) + (f0 << 2));  // (one multiply-add and one shift-add per posting)
      s1 = lds_u32(tbl + __umul24(nb1, 4u * SCORE_TABLE_COLS) + (f1 << 2));
    } else {
      const float* cache = caches + c * 64;
      const float wk = __int_as_float(readlane(__float_as_int(c_wk), c));
      // (v_rcp_f32 is good to one ulp; this kernel's scores are pinned to 1e-5 — the header — and rounded to fixed point
      // next: an IEEE division would be thirteen instructions per posting for nothing)
      const float q0 = (float)(int32_t)f0, q1 = (float)(int32_t)f1;
      s0 = to_fixed(wk * q0 * __builtin_amdgcn_rcpf(q0 + cache[nb0]));
      s1 = to_fixed(wk * q1 * __builtin_amdgcn_rcpf(q1 + cache[nb1]));
    }
    const uint32_t o0 = (uint32_t)(e0 - w0), o1 = (uint32_t)(e1 - w0);
    if (RGPU_ORX_ABL == 6) {  // plain stores instead of atomics
      if (o0 < wlen) acc[o0] = s0;
      if (o1 < wlen) acc[o1] = s1;
      return;
    }
    // postings outside the window (a block may reach into its neighbours) go to this lane's spare cell behind it:
    // one select instead of an exec-mask branch around the atomic
    // (a doc id is below max_doc, so an offset is either inside [0, wlen) or at least WS: one unsigned min does the select;
    // an offset in [WS, WS + lane) lands in another lane's spare cell, which is as good)
    (void)wlen;
    __hip_atomic_fetch_add(acc + min(o0, spare), s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(acc + min(o1, spare), s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int hits_wave = 0;  // touched docs this wavefront scanned (wave-uniform)
  const int kth_m = (k + ORX_WAVES - 1) / ORX_WAVES;  // <= 16: the m-th best key of a list is lane m - 1 of its first register
  uint32_t wg_seen = 0;  // lane l: wavefront (l % 8)'s m-th best total as of the previous window
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);

  List cur_list = finish_list(build_list(win0, 0));
  List next_list = finish_list(build_list(win0 + 1, 0));
  append_next(cur_list, next_list);
  Slot ring[ORX_RING];
#pragma unroll
  for (int j = 0; j < ORX_RING; ++j) ring[j] = fetch(cur_list, j);  // (entries past cur_list.n: the appended ones — harmless)
  __syncthreads();  // accumulators are cleared, tables and caches are built

#ifdef RGPU_ORX_TIME
  long long orx_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int win = win0; win < win1; ++win) {
    ORX_STAMP(t0);
    const int32_t w0 = win * WS;
    const uint32_t wlen = (uint32_t)(min(seg.max_doc, w0 + WS) - w0);
    const uint64_t seen = shared.peek();  // folded before the scan
    // ---- this wavefront's blocks, in groups of ORX_RING; after a block is done its ring slot requests the block
    // ORX_RING further on — in the last group that is block j of the NEXT window (lanes n + j: static slot alignment).
    // Entries past the list's end are clamped: a redundant request instead of a branch.
    const int nb_mine = cur_list.n;
    const int padded = max(ORX_RING, (nb_mine + ORX_RING - 1) / ORX_RING * ORX_RING);
    for (int i = 0; i < padded; i += ORX_RING) {
      const bool last_group = i + ORX_RING >= padded;
#pragma unroll
      for (int j = 0; j < ORX_RING; ++j) {
        const int idx = i + j;
        if (idx < nb_mine && RGPU_ORX_ABL != 2 && RGPU_ORX_ABL != 3) process(ring[j], cur_list, idx, w0, wlen);
        if (RGPU_ORX_ABL == 3) continue;
        ring[j] = fetch(cur_list, last_group ? nb_mine + j : min(idx + ORX_RING, max(0, nb_mine - 1)));
      }
    }
    for (int page = ORX_PAGE; page < cur_list.mine; page += ORX_PAGE) {  // more blocks than a page for one wavefront: plain loop
      const List more = finish_list(build_list(win, page));
      for (int idx = 0; idx < more.n; ++idx) process(fetch(more, idx), more, idx, w0, wlen);
    }
    ORX_STAMP(t1);
    if (wave == 0) {  // singletons (one lane per clause; two clauses may name the same doc: the add is atomic)
      const uint32_t o = (uint32_t)(c_sdoc - w0);
      if (c_sdoc >= 0 && o < wlen) __hip_atomic_fetch_add(acc + o, c_sfix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- bounds of window win + 2 from the directory entries requested one window ago; request the next ones
    if (RGPU_ORX_ABL != 5) {
#pragma unroll
      for (int s = 0; s < ORX_OWN; ++s) {
        const int c = wave + s * ORX_WAVES;
        if (c < n) { bounds_finish(c, own_cur[s], own_e[s], win + 2); bounds_issue(c, own_cur[s], own_e[s]); }
      }
    }
    ORX_STAMP(t2);
    __syncthreads();  // every add of this window has landed; bounds of win + 2 are visible
    ORX_STAMP(t3);
    const Pending after = build_list(win + 2, 0);  // its directory words arrive during the scan

    // ---- scan: eight docs per lane per step; a touched accumulator is one collected hit (bulk_scorer.rs:114-120)
    shared.fold(seen, tau, floor);
    if (RGPU_ORX_ABL != 14) {  // the workgroup's bound as of the previous window (see below)
      static_assert((ORX_WAVES & (ORX_WAVES - 1)) == 0 && ORX_WAVES <= 16, "the min below runs inside one 16-lane row");
      uint32_t v = wg_seen;
      // lane 15 ends up with the min over lanes 16 - ORX_WAVES .. 15, i.e. over every wavefront's entry (row_shr; a lane
      // without a source keeps its own value)
      auto shr_min = [](uint32_t x, auto ctrl) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, decltype(ctrl)::value, 0xf, 0xf, false);
        return o < x ? o : x;
      };
      v = shr_min(v, std::integral_constant<int, 0x111>());
      if (ORX_WAVES > 2) v = shr_min(v, std::integral_constant<int, 0x112>());
      if (ORX_WAVES > 4) v = shr_min(v, std::integral_constant<int, 0x114>());
      if (ORX_WAVES > 8) v = shr_min(v, std::integral_constant<int, 0x118>());
      const uint64_t wg_bound = (uint64_t)(uint32_t)readlane((int)v, 15) << 32;
      if (wg_bound > floor) {
        floor = wg_bound;
        if (floor > tau) tau = floor;
        shared.publish_key(wg_bound, lane);
      }
    }
    // (the next step's cells are requested before this step's are looked at: the LDS round trip hides behind the work)
    ORX_STAMP(t3a);
    uint4* cell = reinterpret_cast<uint4*>(acc + 4 * threadIdx.x);  // this lane's cells of a step: [0, 4) and [4 T, 4 T + 4)
    uint4 a_next = cell[0], b_next = cell[ORX_THREADS];
    for (uint32_t i0 = 0; i0 < (uint32_t)WS && RGPU_ORX_ABL != 1; i0 += ORX_SCAN_STEP) {
      const uint4 a = a_next, b = b_next;
      uint4* const here = cell;
      if (i0 + ORX_SCAN_STEP < (uint32_t)WS) cell += ORX_SCAN_STEP / 4;  // wave-uniform
      a_next = cell[0]; b_next = cell[ORX_THREADS];
      if (__ballot((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0u)) {
        // touched docs: counted on the scalar side (one compare per cell, popcounts and adds are SALU work)
        if (RGPU_ORX_ABL != 13) hits_wave += __popcll(__ballot(a.x != 0u)) + __popcll(__ballot(a.y != 0u)) + __popcll(__ballot(a.z != 0u)) +
                     __popcll(__ballot(a.w != 0u)) + __popcll(__ballot(b.x != 0u)) + __popcll(__ballot(b.y != 0u)) +
                     __popcll(__ballot(b.z != 0u)) + __popcll(__ballot(b.w != 0u));
        if (RGPU_ORX_ABL != 12) {
          here[0] = make_uint4(0u, 0u, 0u, 0u);
          here[ORX_THREADS] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t thr = max(1u, (uint32_t)(tau >> 32));  // a key's high word is the doc's total
        if (RGPU_ORX_ABL != 11 && __ballot(max(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w))) >= thr)) {
          uint32_t nd = ~(uint32_t)(w0 + (int32_t)i0 + 4 * (int32_t)threadIdx.x);  // ~doc: smaller doc id = larger key
          auto offer = [&](uint32_t v, uint32_t ndoc) {
            const uint64_t key = v >= thr ? ((uint64_t)v << 32) | ndoc : 0ull;
            if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          };
          offer(a.x, nd); offer(a.y, nd - 1u); offer(a.z, nd - 2u); offer(a.w, nd - 3u);
          nd -= (uint32_t)(4 * ORX_THREADS);
          offer(b.x, nd); offer(b.y, nd - 1u); offer(b.z, nd - 2u); offer(b.w, nd - 3u);
        }
      }
    }
    ORX_STAMP(t4a);
    shared.publish<WIDE>(top, k, lane);
    // The workgroup's own bound on the query's k-th best: every doc is scanned by exactly one wavefront, so if each of
    // the eight lists holds m = ceil(k / 8) keys >= v there are k docs at or above v — the smallest of the eight m-th
    // best keys is such a v, and it is about the k-th best of everything the WORKGROUP has seen, where a single list's
    // k-th best is only that of an eighth of it. (Without it a list's threshold is the k-th best of 1/64 of the query's
    // docs and some 12 000 keys per query are inserted into one list or another: 1.8 of the kernel's 11.1 ms.) Only the
    // keys' high words — the totals — are exchanged: (total << 32) is a bound as valid as the key itself.
    if (lane == 0) wg_kth[wave] = (uint32_t)(readlane64(top.a, kth_m - 1) >> 32);  // 0 while the list holds fewer than m keys
    ORX_STAMP(t4);
    __syncthreads();  // the window is clear again (measured: dropping this barrier would gain 1.6 %)
    ORX_STAMP(t5);
    wg_seen = wg_kth[lane & (ORX_WAVES - 1)];  // folded in front of the next scan: the read's round trip hides behind the blocks
    // the window after next's directory words have arrived during the scan: finish it, move up, and give the new
    // current list its look-ahead entries
    cur_list = next_list;
    next_list = finish_list(after);
    append_next(cur_list, next_list);
#ifdef RGPU_ORX_TIME
    {
      const long long t6 = (long long)__builtin_readcyclecounter();
      orx_t[0] += t1 - t0; orx_t[1] += t2 - t1; orx_t[2] += t3 - t2; orx_t[3] += t4a - t3a; orx_t[6] += t3a - t3; orx_t[7] += t4 - t4a; orx_t[4] += t5 - t4; orx_t[5] += t6 - t5;
    }
#endif
  }
#ifdef RGPU_ORX_TIME
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_orx_dbg[i], (unsigned long long)orx_t[i]);
#endif
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  if (lane == 0) partial_counts[item] = hits_wave;
}

}  // namespace rgpu
