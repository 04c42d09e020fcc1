// Single-term queries (TermScorer + TopDocsCollector) on the GPU. Replaces, per leaf:
//   search/scorer/term_scorer.rs:43-67              TermScorer (postings iteration + BM25SimScorer::score)
//   search/scorer/bulk_scorer.rs:114-120            the collect loop
//   search/collector/top_docs.rs:67-94,157-172      TopDocsCollector::{collect, add_doc}
//
// Work items = (query, chunk of `blocks_per_item` FullBlocks), one wavefront each; the last chunk of a term also
// takes its VInt tail. A workgroup is TERM_WAVES consecutive items, and the waves of a workgroup that work on
// the SAME query share ONE top-k list in LDS (a "group", led by the lowest such wave): k-th best of 8 x 128
// blocks instead of 64 rises ~3x faster, and what a block costs depends on whether it can still contribute —
// see term_blocks_fast. The list is checked out into the registers of whichever wave has candidates (WaveTopK,
// under an LDS spin lock) and written back; the wave of a group that finishes last emits the list as the
// group's partial result (the other items emit empty lists), k_merge_items merges groups.
#pragma once
#include "search.hpp"

namespace rgpu {

#ifdef RGPU_EXP_COUNT  // developer instrumentation (variant builds only): [0] blocks, [1] blocks that took the doc-id path, [2] blocks looked at (not pruned)
__device__ unsigned long long g_term_dbg[4];
#endif

#ifndef RGPU_TERM_PRUNE  // 0: variant builds that measure the unpruned kernel
#define RGPU_TERM_PRUNE 1
#endif
#ifndef RGPU_TERM_ORDER
#define RGPU_TERM_ORDER 1
#endif
#ifndef RGPU_TERM_EXCHANGE
#define RGPU_TERM_EXCHANGE 2
#endif
#ifndef RGPU_TERM_EXCHANGE_MAX_ITEMS
#define RGPU_TERM_EXCHANGE_MAX_ITEMS 16
#endif
// A query's items start together, each with an empty top-k list. The same batch with every item starting from its query's FINAL
// threshold (a variant build that keeps the shared thresholds between launches): 0.041 ms against 0.072, 12 k blocks unpacked
// against 71 k; 0.131 against 0.178 ms at 100 M docs — the price of not knowing the answer. RGPU_TERM_WAIT = 1 was the attempt to
// buy some of that: the head item (the term's first blocks) publishes its k-th best key after its first 64 blocks and the query's
// other items wait for that publication before they look at a block (s_sleep between polls of the query's threshold word, at most
// RGPU_TERM_WAIT_POLLS of them: a bounded wait). Measured: 57.5 k blocks unpacked instead of 70.6 k and the SAME kernel time
// (0.0722-0.0732 against 0.0711-0.0721 ms; shorter items do not pay either: 128 blocks per item 0.080 against 0.074) — a term's
// ten best postings are spread over the whole list, the first 8192 postings say little about them. Off.
#ifndef RGPU_TERM_WAIT
#define RGPU_TERM_WAIT 0
#endif
#ifndef RGPU_TERM_WAIT_POLLS
#define RGPU_TERM_WAIT_POLLS 48
#endif
#ifndef RGPU_TERM_WAIT_SLEEP
#define RGPU_TERM_WAIT_SLEEP 16  // s_sleep argument: 64 cycles each
#endif
constexpr int TERM_EXCHANGE_MAX_ITEMS = RGPU_TERM_EXCHANGE_MAX_ITEMS;
#ifndef RGPU_TERM_SKIP_EMPTY_RING
#define RGPU_TERM_SKIP_EMPTY_RING 1
#endif
#ifndef RGPU_TERM_WAVES
#define RGPU_TERM_WAVES 8
#endif
constexpr int TERM_WAVES = RGPU_TERM_WAVES;
constexpr int TERM_THREADS = 64 * TERM_WAVES;
// per-wave LDS slice: [FullBlock staging slab 2 x 528 B | norm cache 64 f32 | score table 64 x 11 f32]; the slab
// comes first so that the extraction's ds_read2_b64 offsets stay immediates. Raw-norm mode (no ranks, no table)
// keeps its 256-entry norm cache in the last 1 KB instead.
constexpr int TERM_BLOCK_SLAB = 2 * SLAB_STREAM;
constexpr int TERM_WAVE_LDS = TERM_BLOCK_SLAB + WAVE_CACHE_FLOATS * 4;
constexpr int TERM_RAW_CACHE_AT = TERM_WAVE_LDS - 256 * 4;
static_assert(TERM_RAW_CACHE_AT >= TERM_BLOCK_SLAB, "raw-norm mode: the 256-entry cache sits behind the slab");
static_assert(TERM_WAVE_LDS % 16 == 0 && TERM_BLOCK_SLAB % 16 == 0, "16-byte aligned slices");
__host__ __device__ constexpr size_t term_lds_bytes(bool wide) {
  return (size_t)TERM_WAVES * TERM_WAVE_LDS + (size_t)TERM_WAVES * (wide ? 128 : 64) * 8 + (size_t)TERM_WAVES * 12;
}

// ---- a top-k list shared by the wavefronts of one group -------------------------------------------------------
// (Round 6's phase counters — scripts/term_timeline.py on a -DRGPU_TERM_TRACE build — put 26 % of term_blocks_fast's cycles, 40-45 % in
// the launch's longest items, into the offer below: ~1450 cycles per offered block. Two alternatives were measured and dropped:
// every wavefront keeping its OWN list in registers, the group sharing only the largest k-th best (ds_max_u64) and merging the lists
// at the end — ~1160 cycles per offer, so the lock is ~300 of them, but the looser threshold unpacks more blocks: k_search_term 0.0315
// against 0.0310 ms at 10 M docs, 0.048 against 0.043 at 100 M; and inserting the best candidate first (one or two wave maxima per
// insertion) instead of in lane order — as many cycles per offer, 0.0303 against 0.0308 ms. What an offer costs is its ~5-7 insertions.)
struct GroupList {
  uint64_t* keys;  // LDS: 64 (128 when WIDE) keys, sorted descending, 0 == empty — WaveTopK's registers at rest
  uint32_t* lock;  // LDS
};
__device__ __forceinline__ void group_lock(uint32_t* lock, int lane) {
  while (true) {
    uint32_t old = 1u;
    if (lane == 0) old = atomicCAS(lock, 0u, 1u);
    if ((uint32_t)readfirstlane((int)old) == 0u) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void group_unlock(uint32_t* lock, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the group's current k-th best (0 while the list holds fewer than k keys); a stale value is only conservative
template <bool WIDE>
__device__ __forceinline__ uint64_t group_kth(const GroupList& g, int k) {
  const uint64_t v = __hip_atomic_load(g.keys + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return ((uint64_t)(uint32_t)readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)readfirstlane((int)(uint32_t)v);
}
// Offer two keys per lane (0 == none) to the group's list; `tau` returns the entry threshold afterwards.
template <bool WIDE>
__device__ __forceinline__ void group_offer2(const GroupList& g, uint64_t key0, uint64_t key1, uint64_t& tau, int k, int lane,
                                             uint64_t floor) {
  group_lock(g.lock, lane);
  WaveTopK top;
  top.a = g.keys[lane];
  if (WIDE) top.b = g.keys[64 + lane];
  const uint64_t kth = topk_threshold<WIDE>(top, k);
  tau = kth > floor ? kth : floor;
  topk_offer<WIDE>(top, key0, tau, k, lane, floor);
  topk_offer<WIDE>(top, key1, tau, k, lane, floor);
  g.keys[lane] = top.a;
  if (WIDE) g.keys[64 + lane] = top.b;
  group_unlock(g.lock, lane);
}

// TermScorer fast path: FullBlocks of a term whose scores come from the LDS table (norm ranks, weight >= 0, no
// deleted docs). The kernel is VALU-issue bound (rocprofv3: 38 VALU + 29 SALU per block at ~80 % VALU busy), so a
// block only does what its outcome can depend on:
//   0. (PRUNE) nothing at all when its frontier word (SegView::dir_bmax: largest norm rank per freq) bounds every
//      posting's score below the entry threshold — 64 blocks are tested at once, one lane each, ten table reads;
//   1. otherwise stage the rows, unpack the FREQ stream, two table reads, one compare of the raw score bits;
//   2. the doc-delta stream is unpacked and prefix-summed only when some posting can still enter the top-k — its
//      base doc then comes from the block directory (dir_last), not from a running scan.
// Every posting is still counted (TopDocsCollector::total_hits: no deletions on this path, so a block is 128 hits
// whether it is looked at or not) and every candidate offered, so results are those of the plain loop; the reference
// itself has no such pruning (term_scorer.rs:43-67 scores every posting) — exactness is the constraint here.
// The bound is exact in f32: table[r][f] is produced by the very expression that scores a posting, and it does not
// fall as the rank r grows when the sim table's cache does not rise with the norm byte (checked on the host,
// TERM_FLAG_MONOTONE) and the weight is >= 0 — x / y with x >= 0 fixed and y = f + cache[r] > 0 shrinking, correctly
// rounded. Absent freqs carry rank 0, whose entry the bound of the block's largest freq dominates up to rounding;
// either way the maximum is taken over a superset of the block's (rank, freq) pairs.
#ifdef RGPU_TERM_TRACE  // shader-clock cycles per phase of term_blocks_fast, summed over the item (developer instrumentation)
#define TERM_PH_PARAM , uint32_t (&ph)[8]
#define TERM_PH_NOW(v) const uint64_t v = (uint64_t)clock64()
#define TERM_PH_ADD(i, since) ph[i] += (uint32_t)((uint64_t)clock64() - (since))
#else
#define TERM_PH_PARAM
#define TERM_PH_NOW(v) do {} while (0)
#define TERM_PH_ADD(i, since) do {} while (0)
#endif
template <bool LEGACY, bool WIDE>
__device__ __forceinline__ void term_blocks_fast(const SegView& seg, const DevTerm& T, int b0, int b1, uint8_t* slab,
                                                 const float* cache, float wk, int lane, const GroupList& group,
                                                 SharedTau& shared, uint64_t& floor, int k, int& count, bool prune, uint32_t& looked,
                                                 uint32_t& touched, uint64_t ceil, bool exchange, bool head TERM_PH_PARAM) {
  constexpr int DEPTH = PREFETCH_DEPTH;
  TERM_PH_NOW(ph_enter);
  const uint8_t* term_rows = seg.bstore + T.bs_base;
  const uint8_t* pn = seg.pnorm + T.pn_base;
#if RGPU_TERM_ORDER == 0
#error "the index-order variant was removed with round 5 (see git history): RGPU_TERM_ORDER must be 1"
#endif
  // Entry test on raw score bits: a posting can enter iff its key exceeds tau = (S, D), i.e. score > S, or score == S and
  // doc < D. BM25 scores of one term take few distinct values (freq <= 10 x norm rank), so ties with the threshold are the common
  // case, not the corner: a block whose docs all lie above D cannot win a tie, and the test against it is strict. Every doc of
  // block b is above `lo` = the last doc of block b - 1 (the directory), so "all above D" is lo >= D - 1 — a property of the
  // block, not of the order the blocks are visited in (round 5: they are visited best bound first, see take()).
  auto thr_of = [&](uint64_t t, int32_t lo) -> uint32_t {
    const uint32_t thi = (uint32_t)(t >> 32);
    if (!(thi & 0x80000000u)) return 0u;  // no threshold yet (or a negative one): everything is a candidate
    const uint32_t bits = thi & 0x7fffffffu;
    return key_doc(t) <= lo + 1 ? bits + 1u : bits;  // (key_doc <= lo + 1: no doc of the block is below the threshold's doc)
  };
  auto fresh_tau = [&]() -> uint64_t {
    const uint64_t kth = group_kth<WIDE>(group, k);
    return kth > floor ? kth : floor;
  };
  // `thr` lives in one scalar register between updates (kept opaque: the compiler would otherwise recompute it
  // from tau / lo — eight scalar instructions — in front of every block's compare)
  auto pin = [](uint32_t v) -> uint32_t { asm volatile("" : "+s"(v)); return v; };
  uint64_t tau = fresh_tau();
#ifdef RGPU_EXP_COUNT
  int dbg_slow = 0, dbg_looked = 0;
#endif
  // a chunk's directory entries (frontier words, the docs in front of its blocks) arrive one chunk ahead of their use — and with
  // them a fresh look at what the query's OTHER wavefronts have published meanwhile (round 5: one look per item at its start
  // left the items of a query warming up side by side, each on its own: 130 k of the headline batch's 2.2 M blocks unpacked
  // where a threshold known in advance needs a few per query)
  #ifndef RGPU_TERM_CHUNK_SUMS
#define RGPU_TERM_CHUNK_SUMS 1
#endif
#ifndef RGPU_TERM_SEEN64
#define RGPU_TERM_SEEN64 0  // 1: carry the whole published key across the chunk (its doc too: strict ties where the doc allows)
#endif
#if RGPU_TERM_SEEN64
  struct Chunk { DirChunk dir; uint64_t bmax; int32_t lo; uint32_t seen_hi; uint32_t seen_lo; };
#else
  struct Chunk { DirChunk dir; uint64_t bmax; int32_t lo; uint32_t seen_hi; };
#endif
  auto load_chunk = [&](int c0, int ci) -> Chunk {  // ci: how many chunks this item has visited before this one
    Chunk c;
    const int nb = min(64, b1 - c0);
    c.dir.load(seg.dir_row, seg.dir_hdr, T.dir_base, c0, nb, lane);
    c.bmax = (prune && lane < nb) ? seg.dir_bmax[T.dir_base + c0 + lane] : 0ull;
    const int e = c0 + lane - 1;  // the block in front of this lane's
    c.lo = (lane < nb && e >= 0) ? seg.dir_last[T.dir_base + e] : -1;
    // A look at what the query's other wavefronts published, with visited chunks 1, 2, 4, 8 ... of the item — but only for a query of few
    // items (`exchange`): every wavefront of a query reads and raises ONE word, and same-address traffic serialises. Measured
    // (k_search_term, the headline batch): 10 M docs, ~5 items per query: 0.077 ms without any exchange, 0.071 with this one,
    // 0.085 with a look per chunk; 100 M docs, ~42 items per query (305 for the longest list): 0.217 / 0.68 / 1.56 ms.
    const bool look = exchange && (RGPU_TERM_EXCHANGE == 1 || (RGPU_TERM_EXCHANGE == 2 && (ci & (ci - 1)) == 0));
    // (the score half of the published key is enough — and one register instead of two across the chunk. The doc half is then
    // taken as the LARGEST doc id: (score, INT_MAX) is at or below the published key whatever its doc, so it only ever drops
    // what the full key would drop, and a tie with it is never read as "lost" — thr_of looks at the threshold's doc)
#if RGPU_TERM_SEEN64
    const uint64_t seen = look ? shared.peek() : 0ull;
    c.seen_hi = (uint32_t)(seen >> 32);
    c.seen_lo = (uint32_t)seen;
#else
    c.seen_hi = look ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(shared.slot) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#endif
    return c;
  };
  // lane j: the best score any posting behind frontier word w can have (raw bits; scores are >= 0 here)
  auto bound_of = [&](uint64_t w) -> uint32_t {
    const uint32_t fmax = (uint32_t)w & 15u;
    uint32_t bb = 0u;
#pragma unroll
    for (int f = 1; f <= SCORE_TABLE_FREQS; ++f) {
      const uint32_t r = (uint32_t)(w >> (4 + 6 * (f - 1))) & 63u;
      const uint32_t sc = __float_as_uint(table_score(cache, r, (uint32_t)f));
      bb = ((uint32_t)f <= fmax && sc > bb) ? sc : bb;
    }
    return fmax > (uint32_t)SCORE_TABLE_FREQS ? 0xffffffffu : bb;
  };
  // Round 6: the item's chunks of 64 blocks, one LANE each, against the frontier of the chunk's 8192 postings (SegView::dir_sum:
  // the field-wise maximum of its blocks' words, so its bound IS the largest of their bounds). The per-block test above costs a
  // wavefront ~80 VALU instructions per chunk — at 100 M docs, 344 k chunks x ~700 cycles over 1024 SIMDs, that WAS the
  // kernel's 0.097 ms — and seven chunks in ten (more on longer lists) hold no block that can enter. Those are now never
  // requested at all: not their frontier words, not their directory rows. The chunks that remain are visited in index order, each
  // requested while the one in front of it is worked on, and the set is re-filtered with the threshold of the moment on the way.
  const int n_chunks = (b1 - b0 + 63) >> 6;
  uint64_t cand = n_chunks >= 64 ? ~0ull : ((1ull << n_chunks) - 1ull);
  const bool summed = RGPU_TERM_CHUNK_SUMS && prune && seg.dir_sum != nullptr && n_chunks >= 2 && n_chunks <= 64 && (b0 & 63) == 0;
  uint32_t cbest = 0xffffffffu;
  if (summed) {
    const int cj = (b0 >> 6) + lane;  // the chunk's index within the term; only whole chunks have a word
    const bool whole = lane < n_chunks && 64 * cj + 64 <= T.nblocks;
    cbest = bound_of(whole ? seg.dir_sum[((T.dir_base + 63u) >> 6) + (uint32_t)cj] : 15ull);
    touched += 8u * (uint32_t)n_chunks;  // (what this item reads of the directory is counted where it is requested)
  }
  auto still = [&](uint64_t t) {  // the non-strict test: thr_of(t, lo) is `bits` or `bits + 1` (cbest is all ones without a word)
    const uint32_t thi = (uint32_t)(t >> 32);
    if (thi & 0x80000000u) cand &= __ballot(cbest >= (thi & 0x7fffffffu));
  };
  count += 128 * (b1 - b0);
  still(tau);
  Chunk next{};
  if (cand) next = load_chunk(b0 + 64 * (int)__builtin_ctzll(cand), 0);
  TERM_PH_ADD(0, ph_enter);
  for (int visited = 0; cand != 0ull; ++visited) {
    TERM_PH_NOW(ph_chunk);
    const int c0 = b0 + 64 * (int)__builtin_ctzll(cand);
    cand &= cand - 1ull;
    const int nb = min(64, b1 - c0);
    const Chunk cur = next;
    touched += 18u * (uint32_t)nb;  // frontier word, store row, header, the doc in front: the directory entries of a visited chunk
    still(tau);  // (what the chunk before this one achieved)
    if (cand) next = load_chunk(b0 + 64 * (int)__builtin_ctzll(cand), visited + 1);
    const DirChunk& dir = cur.dir;
    {
      uint64_t t0 = 0;
#if RGPU_TERM_SEEN64
      shared.fold(cur.seen_hi != 0u ? (((uint64_t)cur.seen_hi << 32) | cur.seen_lo) : 0ull, t0, floor);
#else
      shared.fold(cur.seen_hi != 0u ? (((uint64_t)cur.seen_hi << 32) | 0x80000000ull) : 0ull, t0, floor);
#endif
    }
    // lane j: the best score any posting of block c0 + j can have
    const uint32_t best = prune ? bound_of(cur.bmax) : 0xffffffffu;
    const uint64_t in_chunk = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
    // lane j: can block c0 + j still put a posting into the top-k? (its bound against the threshold of the moment, strict when
    // none of its docs can win a tie)
    auto may_enter = [&](uint64_t t) -> bool { return best >= thr_of(t, cur.lo); };
    auto step = [&](int idx, const uint4& rows, uint32_t nn) {
      const uint32_t hdr = dir.hdr_at(idx);
      const int bf = hdr_bfreq(hdr);
      TERM_PH_NOW(ph_s0);
      stage_rows(rows, slab, lane);
      wave_sync();
      TERM_PH_ADD(2, ph_s0);
      TERM_PH_NOW(ph_s1);
      uint32_t f0, f1;
      bool in_table = true;  // wave-uniform: every freq of the block has a table column
      if (bf) {
        extract_pair<LEGACY>(slab + SLAB_STREAM, bf, lane, f0, f1);
        if (bf > 3) in_table = !__ballot((f0 > f1 ? f0 : f1) > (uint32_t)SCORE_TABLE_FREQS);
      } else {
        const uint32_t f = (uint32_t)readlane((int)rows.x, 32);  // all-equal stream: its value
        f0 = f1 = f;
        in_table = f <= (uint32_t)SCORE_TABLE_FREQS;
      }
      const uint32_t nb0 = nn & 0xffu, nb1 = nn >> 8;
      float s0, s1;
      if (in_table) {
        s0 = table_score(cache, nb0, f0);
        s1 = table_score(cache, nb1, f1);
      } else {
        s0 = bm25_score(wk, (float)(int32_t)f0, cache[nb0]);
        s1 = bm25_score(wk, (float)(int32_t)f1, cache[nb1]);
      }
#ifdef RGPU_EXP_COUNT
      ++dbg_looked;
#endif
      looked += 1u;                            // what this launch really decoded (rgpu_last_search_counters): scalar adds
      touched += encoded_block_bytes(hdr) + 128u;  // both streams' rows are requested together + the posting-order norms
      const int32_t base = readlane(cur.lo, idx) < 0 ? 0 : readlane(cur.lo, idx);  // (block 0 of the term: deltas count from doc 0)
      const uint32_t thr = pin(thr_of(tau, readlane(cur.lo, idx)));
      const uint32_t r0 = __float_as_uint(s0), r1 = __float_as_uint(s1);
      TERM_PH_ADD(3, ph_s1);
      if (__ballot((r0 > r1 ? r0 : r1) >= thr)) {
        TERM_PH_NOW(ph_s2);
#ifdef RGPU_EXP_COUNT
        ++dbg_slow;
#endif
        uint32_t e0, e1;
        staged_doc_deltas<LEGACY>(slab, rows, hdr, lane, e0, e1);
        int32_t d0, d1;
        deltas_to_docs(e0, e1, base, d0, d1);
        const uint64_t key0 = below(make_key(s0, d0), ceil), key1 = below(make_key(s1, d1), ceil);
        // most blocks that get here only tie with the threshold or trail a fresher one: look at the group's
        // current k-th best (one LDS read) before paying for the lock and the list check-out
        tau = fresh_tau();
        TERM_PH_ADD(4, ph_s2);
        TERM_PH_NOW(ph_s3);
        if (__ballot((key0 > key1 ? key0 : key1) > tau)) {
          group_offer2<WIDE>(group, key0, key1, tau, k, lane, floor);
#ifdef RGPU_TERM_TRACE
          ph[7] += 1u;
#endif
        }
        TERM_PH_ADD(5, ph_s3);
      }
      wave_sync();  // slab is free for the next block
    };
    auto norms_of = [&](int idx) -> uint32_t {
      return *reinterpret_cast<const uint16_t*>(pn + (128u * (uint32_t)(c0 + idx) + 2u * (uint32_t)lane));
    };
    // Blocks that may still matter, streamed through a DEPTH-deep ring of row (and norm) loads, the block with the LARGEST bound
    // first: its postings lift the threshold the most, and every block whose bound the new threshold exceeds is never requested
    // (in index order a dense term's first chunk unpacked block after block while the threshold crept up). `todo` is re-filtered
    // with the threshold of the moment before every round: the threshold only rises, so a stale test is merely conservative.
    // Ring slots without a block (slot[j] < 0) reload the chunk's first block instead of being guarded: a redundant load is
    // cheaper than a load behind a branch.
    tau = fresh_tau();
    uint64_t todo = in_chunk & __ballot(may_enter(tau));
    TERM_PH_ADD(1, ph_chunk);
    int slot[DEPTH];
    uint4 ring[DEPTH];
    uint32_t nring[DEPTH];
    auto take = [&]() -> int {
      if (!todo) return -1;
      int i;
      if (prune) {
        const uint32_t mine = ((todo >> lane) & 1ull) ? best : 0u;
        const uint32_t top = wave_reduce_max_u32(mine);
        i = (int)__builtin_ctzll(todo & __ballot(mine == top));
      } else {
        i = (int)__builtin_ctzll(todo);
      }
      todo &= ~(1ull << i);
      return i;
    };
#if RGPU_TERM_SKIP_EMPTY_RING
    // (round 6: a chunk none of whose blocks can enter — seven in ten on the headline batch — used to fill the ring all the same:
    // DEPTH redundant row + norm loads of its first block, "cheaper than a load behind a branch" inside a chunk, but 64 wasted
    // kilobyte loads per item across its chunks. The chunk-level test is one scalar branch around the ring — fill AND drain: with the
    // drain outside it the compiler wanted 71-73 VGPRs. Measured, same box: k_search_term 0.0379-0.0386 -> 0.0366-0.0374 ms at 10 M docs,
    // 0.1286-0.1303 -> 0.1005-0.1015 ms at 100 M.)
    if (todo != 0ull) {
#endif
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      slot[j] = take();
      const int pj = slot[j] < 0 ? 0 : slot[j];
      ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(pj)), dir.hdr_at(pj), lane);
      nring[j] = norms_of(pj);
    }
    while (slot[0] >= 0) {  // slots fill in order, so an empty slot 0 means an empty ring
      // what the group's other wavefronts achieved meanwhile: one LDS read per DEPTH blocks
      tau = fresh_tau();
      if (prune) todo &= __ballot(may_enter(tau));
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {  // static ring slot j
        const uint4 rows = ring[j];
        const uint32_t nn = nring[j];
        const int idx = slot[j];
        slot[j] = take();
        const int pj = slot[j] < 0 ? 0 : slot[j];
        ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(pj)), dir.hdr_at(pj), lane);
        nring[j] = norms_of(pj);
        if (idx >= 0) step(idx, rows, nn);
      }
    }
#if RGPU_TERM_SKIP_EMPTY_RING
    }
#endif
    // what this chunk achieved, for the query's other wavefronts (an atomic only when the group's k-th best has risen)
    {
      const int ci = visited + 1;
      // (... and the query's head item after its first 64 blocks, whatever the query's size: its other items wait for exactly
      // that — RGPU_TERM_WAIT in k_search_term)
      const bool first_of_head = RGPU_TERM_WAIT && head && c0 == b0;
      if (first_of_head || (exchange && (RGPU_TERM_EXCHANGE == 1 || (RGPU_TERM_EXCHANGE == 2 && (ci & (ci - 1)) == 0)))) shared.publish_key(group_kth<WIDE>(group, k), lane);
    }
  }
  TERM_PH_ADD(6, ph_enter);
#ifdef RGPU_EXP_COUNT
  if (lane == 0) {
    atomicAdd(&g_term_dbg[0], (unsigned long long)(b1 - b0));
    atomicAdd(&g_term_dbg[1], (unsigned long long)dbg_slow);
    atomicAdd(&g_term_dbg[2], (unsigned long long)dbg_looked);
  }
#endif
}

// ---- block-max sketches: a starting threshold for a single-term query from the block directory alone ---------------------------
// Every FullBlock holds a posting with the block's largest freq and, among those, the largest norm rank — the frontier word's
// (fmax, rank) entry (SegView::dir_bmax): a REAL posting, whose score under any query is one table read. The k-th largest of
// those scores over a term's blocks is reached by k postings of k different blocks, so the query's k-th best score cannot lie
// below it: a valid threshold before a single block is unpacked. It carries no doc id — the key is (score, largest doc id): a
// tie with it is never read as lost. Measured with the threshold computed per query in front of k_search_term (a variant build):
// 13.3 k blocks unpacked instead of 70.7 k on the headline batch (12.1 k with every query's FINAL threshold), k_search_term
// 0.038 ms instead of 0.072; 17.4 k instead of 188.6 k and 0.127 instead of 0.177 ms at 100 M docs — without it every
// workgroup of a long list finds its own k best postings first. What the k best blocks ARE does not depend on the query (a
// non-negative weight scales every score alike), only — weakly — on the similarity's table: so the (freq, rank) pairs of a term's
// TERM_SKETCH_K best blocks are kept per term (k_term_sketch, built the first time a single-term query names a term of
// TERM_SKETCH_MIN_BLOCKS blocks or more, from the table that query brings), and an item of k_search_term turns the first k of
// them into scores with ITS query's table: k table reads and a wave minimum. Under another table the k pairs are still k real
// postings of k blocks — the threshold stays valid, it is just not the tightest.
constexpr int TERM_SKETCH_K = 128;          // entries per sketch = the largest k one pass serves (RGPU_PASS_K)
#ifndef RGPU_TERM_SKETCH_MIN_BLOCKS
#define RGPU_TERM_SKETCH_MIN_BLOCKS 16
#endif
// (64 until round 6, "shorter lists are one item's work either way": they were the launch's LONGEST items — 40-60 blocks, 16 of them
// unpacked one after the other while the threshold crept up, 25 us of a 26 us launch. With a sketch from 16 blocks up: k_search_term
// 0.0306 -> 0.0278 ms at 10 M docs, 0.0424 -> 0.0421 at 100 M; cutting such lists into items of 32 or 16 blocks on top: 0.031 / 0.032.)
constexpr int TERM_SKETCH_MIN_BLOCKS = RGPU_TERM_SKETCH_MIN_BLOCKS;  // 256 bytes of sketch per term of this many blocks or more
constexpr int TERM_SKETCH_WAVES = 4;
struct SketchJob {
  uint32_t dir_base;  // the term's first directory slot
  int32_t nblocks;
  int32_t sim_table;
  uint32_t out;       // index of the sketch to write
};
__global__ __launch_bounds__(64 * TERM_SKETCH_WAVES) void k_term_sketch(SegView seg, const SketchJob* __restrict__ jobs, int n_jobs,
                                                                        uint16_t* __restrict__ sketches) {
  __shared__ float caches[TERM_SKETCH_WAVES][WAVE_CACHE_FLOATS];
  const int lane = lane_id();
  const int wave = wave_id();
  const int j = (int)blockIdx.x * TERM_SKETCH_WAVES + wave;
  if (j >= n_jobs) return;
  const SketchJob J = jobs[j];
  float* cache = caches[wave];
  float k1;
  load_sim_table(seg, J.sim_table, cache, lane, k1);
  build_score_table(cache, k1 + 1.0f, lane);  // (weight 1: any non-negative weight orders the blocks the same way)
  auto pair_of = [&](uint64_t w, uint32_t& fmax, uint32_t& r) -> bool {
    fmax = (uint32_t)w & 15u;
    const bool real = fmax >= 1u && fmax <= (uint32_t)SCORE_TABLE_FREQS;  // (15: a freq beyond the table — no entry for this block)
    r = (uint32_t)(w >> (4 + 6 * ((real ? fmax : 1u) - 1u))) & 63u;
    return real;
  };
  WaveTopK top;
  uint64_t tau = 0;
  const int kk = min(TERM_SKETCH_K, J.nblocks);
  constexpr int AHEAD = 4;  // chunks of 64 frontier words in flight
  for (int c0 = 0; c0 < J.nblocks; c0 += 64 * AHEAD) {
    uint64_t w[AHEAD];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) w[a] = seg.dir_bmax[J.dir_base + min(c0 + 64 * a + lane, J.nblocks - 1)];  // (clamped; dropped below)
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      const int b = c0 + 64 * a + lane;
      uint32_t fmax, r;
      const bool real = pair_of(w[a], fmax, r) && b < J.nblocks;
      const uint32_t sc = __float_as_uint(table_score(cache, r, real ? fmax : 1u));
      // one key per block: the block index keeps equal scores apart (earlier blocks first: ~b)
      const uint64_t key = real ? (((uint64_t)sc << 32) | (uint32_t)~(uint32_t)b) : 0ull;
      if (__ballot(key > tau)) topk_offer<true>(top, key, tau, kk, lane);
    }
  }
  uint16_t* out = sketches + (size_t)J.out * TERM_SKETCH_K;
  auto entry_of = [&](uint64_t key) -> uint16_t {
    if (key == 0ull) return (uint16_t)0;
    uint32_t fmax, r;
    (void)pair_of(seg.dir_bmax[J.dir_base + (~(uint32_t)key)], fmax, r);
    return (uint16_t)(fmax | (r << 4));
  };
  out[lane] = entry_of(top.a);
  out[64 + lane] = entry_of(top.b);
}
// the first k entries of a sketch as a threshold key under THIS query's score table (0: fewer than k entries)
template <bool WIDE>
__device__ __forceinline__ uint64_t sketch_floor(const uint16_t* __restrict__ sk, const float* cache, int k, int lane) {
  const uint32_t e0 = lane < k ? (uint32_t)sk[lane] : 0xffffu;
  const uint32_t e1 = (WIDE && lane + 64 < k) ? (uint32_t)sk[64 + lane] : 0xffffu;
  if (__ballot(e0 == 0u || e1 == 0u)) return 0ull;
  uint32_t s = 0xffffffffu;
  if (lane < k) s = __float_as_uint(table_score(cache, e0 >> 4, e0 & 15u));
  if (WIDE && lane + 64 < k) { const uint32_t s1 = __float_as_uint(table_score(cache, e1 >> 4, e1 & 15u)); s = s1 < s ? s1 : s; }
  const uint32_t m = ~wave_reduce_max_u32(~s);  // the smallest of the k scores (non-negative floats order like their bits)
  if (m == 0u || (m & 0x80000000u)) return 0ull;
  return ((uint64_t)(m | 0x80000000u) << 32) | 0x80000000ull;  // (score, largest doc id)
}

// (eight wavefronts per SIMD = 64 VGPRs hold the headline instantiation — packed blocks, k <= 64; a second top-k register pair
// (k > 64) or the legacy decode need a few more: seven wavefronts, 72 VGPRs, instead of 12 B of scratch per lane)
// Round 6: the launch bounds ASK for one wavefront less than the kernel gets — 7 for the headline instantiation (63 VGPRs: still
// eight per SIMD), 6 for the others (65: still seven). What the bound buys is the compiler's SGPR budget: at "8" it is 80 (800 / 8,
// minus 16 for the trap handler, rounded to 16), at 7 it is 96, at 6 it is 102 — and the headline instantiation's 88 SGPR spills
// (v_writelane / v_readlane into two VGPRs, 12-24 reloads in each block loop) become 29, the others' 36-44 become 14-23.
// Measured, same box: k_search_term 0.0405-0.0418 -> 0.0378-0.0382 ms at 10 M docs, 0.139 -> 0.131-0.133 ms at 100 M.
#ifdef RGPU_TERM_TRACE  // developer instrumentation (variant builds only): every item's {start, end} wall clock (100 MHz), query, chunk,
constexpr int TERM_TRACE_CAP = 1 << 17;           // blocks looked at / unpacked — the launch's timeline, read back by rgpu_debug_trace
struct TermTraceRec { unsigned long long t0, t1; int32_t q, chunk, blocks, unpacked; unsigned int d_term, d_table, d_sketch, d_pad; unsigned int ph[8]; };  // d_*: 10 ns ticks from t0; ph: shader cycles (TERM_PH_ADD)
__device__ TermTraceRec g_term_trace[TERM_TRACE_CAP];
#endif
#ifndef RGPU_TERM_OTHER_WAVES
#define RGPU_TERM_OTHER_WAVES 6
#endif
#ifndef RGPU_TERM_FAST_WAVES
#define RGPU_TERM_FAST_WAVES 6
#endif
// The launch's last step inside the launch (round 6): the wavefront that finishes a query's LAST item folds the query's item lists and
// writes the caller's row — what k_merge_items did in a launch of its own (7 us of kernel behind a 30 us one, plus the gap between
// two launches, on the headline batch). Every item writes its list and count, makes them visible to the device (release fence at agent
// scope: the lists sit in the L2 of whichever XCD ran the item) and counts itself at done[q]; the one that reads q_items - 1 there has
// seen every other item's release, takes an acquire fence and runs merge_query_items. done == nullptr: no fold, k_merge_items follows.
struct TermMerge {
  unsigned int* done = nullptr;  // per query, zeroed with the plan: items that have written their list
  const int64_t* item_prefix;   // the host's item layout (k_merge_items' head_items form)
  HitOut* hits;                 // the caller's rows, k hits each, and ...
  int64_t* totals;              // ... hit counts, row qmap[q]
  unsigned long long* ceil_out; // (nullable) per row: this pass's worst key when it filled all k slots, else 0 — as k_merge_items
  int32_t doc_base;
  int32_t out_stride, col0;     // the caller's rows are out_stride hits long and this pass fills columns [col0, col0 + k) (0, 0: k-long rows)
};
template <bool LEGACY, bool WIDE>
__global__ __launch_bounds__(TERM_THREADS, (LEGACY || WIDE) ? RGPU_TERM_OTHER_WAVES : RGPU_TERM_FAST_WAVES) void k_search_term(SegView seg, const DevQuery* __restrict__ queries,
                                                              const DevTerm* __restrict__ terms,
                                                              const int4* __restrict__ item_desc, int n_queries,
                                                              int64_t n_items, int blocks_per_item, int k,
                                                              uint64_t* __restrict__ partial_keys,
                                                              int32_t* __restrict__ partial_counts,
                                                              unsigned long long* __restrict__ tau_slots,
                                                              unsigned long long* __restrict__ work_slots,
                                                              const unsigned long long* __restrict__ ceil_slots,
                                                              const int32_t* __restrict__ qmap, TermMerge fold) {
  // ceil_slots (nullable): per CALLER row (qmap[q]) the key this pass's hits must stay below (wave.hpp `below`)
  // work_slots (nullable): [q] += encoded bytes of the FullBlocks this launch decoded for query q (+ their norms),
  // [n_queries + q] += their number — with block-max pruning a small part of the lists ("touched" vs "scan" bytes)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int LIST_N = WIDE ? 128 : 64;
  const int lane = lane_id();
  const int wave = wave_id();
  uint8_t* slice = smem + wave * TERM_WAVE_LDS;
  uint8_t* slab = slice;
  const bool ranked = seg.n_norm_ranks > 0;
  float* cache = reinterpret_cast<float*>(slice + (ranked ? TERM_BLOCK_SLAB : TERM_RAW_CACHE_AT));
  uint64_t* lists = reinterpret_cast<uint64_t*>(smem + TERM_WAVES * TERM_WAVE_LDS);
  uint32_t* locks = reinterpret_cast<uint32_t*>(lists + TERM_WAVES * LIST_N);
  uint32_t* remaining = locks + TERM_WAVES;  // waves of a group that have finished, counted at the leader's slot
  int32_t* wave_query = reinterpret_cast<int32_t*>(remaining + TERM_WAVES);

  // Item order: the first chunk of every query comes first (items 0..n_queries-1), the remaining chunks follow
  // query-major — consecutive items, i.e. the waves of a workgroup, mostly belong to one query. Workgroups
  // start in order, so by the time most chunks begin, their query's first chunk has already published a
  // threshold (SharedTau). Every wave resolves its own item and posts the query in LDS for the others.
  const int64_t item = (int64_t)blockIdx.x * TERM_WAVES + wave;
#ifdef RGPU_TERM_TRACE
  const unsigned long long trace_t0 = (unsigned long long)wall_clock64();
#endif
  // item_desc[item] = {query, chunk, the query's term (-1: absent from this leaf), the query's item count}: written by the host next to
  // the plan. Round 6's item timeline (scripts/term_timeline.py) showed every item of the launch paying ~15 us before it looked at
  // its first block — a chain of dependent round trips (two for the item -> query search over item_prefix, the query, the term, then
  // table / sketch / frontier words) on a chip where 5112 wavefronts start at once; the descriptor cuts it to item -> term -> data
  // (k_search_term 0.0391-0.0399 -> 0.0380-0.0381 ms at 10 M docs and 0.133 -> 0.129 at 100 M on one box, 0.0385-0.0388 -> 0.0384-0.0389
  // on another: at best 3 %). Requesting the item's first directory
  // chunk and its sketch entries BEFORE the score table is built, to overlap one more round trip, was measured too: slower
  // (0.0403-0.0417 ms, whatever the launch bounds: the registers it holds across the table build cost more than the trip).
  int q = -1, chunk = 0, first_term = -1, q_items = 1;
  if (item < n_items) {
    const int4 d = item_desc[item];
    q = d.x; chunk = d.y; first_term = d.z; q_items = d.w & 0xffffff;
    if (d.w >> 24) blocks_per_item = 1 << (d.w >> 24);  // this query's own item size (the host's term_query_item_blocks)
  }
  lists[wave * LIST_N + lane] = 0ull;
  if (WIDE) lists[wave * LIST_N + 64 + lane] = 0ull;
  if (lane == 0) {
    locks[wave] = 0u;
    remaining[wave] = 0u;
    wave_query[wave] = q;
  }
  __syncthreads();
  if (q < 0) return;  // past the last item
  const uint64_t same = __ballot(lane < TERM_WAVES && wave_query[lane < TERM_WAVES ? lane : 0] == q);
  const int leader = __builtin_ctzll(same);
  const int members = __popcll(same);
  const GroupList group{lists + leader * LIST_N, locks + leader};
  SharedTau shared{tau_slots + q};
  int count = 0;
  uint32_t looked = 0, touched = 0;
  const uint64_t ceil = ceil_slots != nullptr ? ceil_slots[qmap[q]] : ~0ull;

#ifdef RGPU_TERM_TRACE
  unsigned int tr_term = 0, tr_table = 0, tr_sketch = 0;
  uint32_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TERM_PH_PASS , ph
#define TERM_TR(v, dep) do { asm volatile("" :: "v"(dep)); v = (unsigned int)((unsigned long long)wall_clock64() - trace_t0); } while (0)
#else
#define TERM_TR(v, dep) do {} while (0)
#define TERM_PH_PASS
#endif
  if (first_term >= 0) {  // else: clause absent from this leaf, nothing to collect
    const DevTerm T = terms[first_term];
    TERM_TR(tr_term, T.df);
    // one look at what earlier workgroups of this query already achieved (per-block exchanges through HBM cost
    // far more in same-address atomics than they save in insertions), one publication when the group is done
    uint64_t floor = 0, tau = 0;
    const uint64_t seen = shared.peek();
    float k1;
    load_sim_table(seg, T.sim_table, cache, lane, k1);
    const float wk = T.weight * (k1 + 1.0f);
    const bool has_norms = seg.norms != nullptr;
    bool tabled = has_norms && seg.n_norm_ranks > 0;
    if (tabled) build_score_table(cache, wk, lane);
    TERM_TR(tr_table, cache[lane]);
    shared.fold(seen, tau, floor);
    // Norms of FullBlock postings arrive in posting order with the payload rows (SegView::pnorm), so scoring a
    // block needs no gather at all; only the VInt tail / singleton (< 128 postings per term) and the optional
    // live-docs test still gather. `full` (std::true_type) marks a FullBlock: two real postings per lane.
    const bool has_live = seg.live != nullptr;
    const bool nonneg = T.weight >= 0.0f;  // idf * boost; negative only with a negative boost
    auto collect = [&](auto full, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nb0, uint32_t nb1, bool v0, bool v1) {
      constexpr bool FULL = decltype(full)::value;
      if (FULL) { v0 = true; v1 = true; }
      if (has_live) {
        v0 = v0 && doc_in_segment(seg, d0) && doc_is_live(seg.live, d0);
        v1 = v1 && doc_in_segment(seg, d1) && doc_is_live(seg.live, d1);
      }
      float s0, s1;
      const uint32_t fmax = f0 > f1 ? f0 : f1;
      if (tabled && !__ballot((v0 || v1) && fmax > (uint32_t)SCORE_TABLE_FREQS)) {
        s0 = table_score(cache, nb0, v0 ? f0 : 1u);
        s1 = table_score(cache, nb1, v1 ? f1 : 1u);
      } else {
        s0 = bm25_score(wk, (float)(int32_t)f0, has_norms ? cache[nb0] : k1);
        s1 = bm25_score(wk, (float)(int32_t)f1, has_norms ? cache[nb1] : k1);
      }
      count += __popcll(__ballot(v0)) + __popcll(__ballot(v1));
      const uint64_t key0 = v0 ? below(make_key(s0, d0), ceil) : 0ull, key1 = v1 ? below(make_key(s1, d1), ceil) : 0ull;
      // `tau` may lag behind the group's list: a stale threshold only lets more keys through to the locked offer
      if (__ballot((key0 > key1 ? key0 : key1) > tau)) group_offer2<WIDE>(group, key0, key1, tau, k, lane, floor);
    };

    const int b0 = chunk * blocks_per_item;
    const int b1 = min(T.nblocks, b0 + blocks_per_item);
    int32_t base = b0 == 0 ? 0 : seg.dir_last[T.dir_base + b0 - 1];
    const uint8_t* term_rows = seg.bstore + T.bs_base;
    auto on_block = [&](int blk, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nb0, uint32_t nb1) {
      looked += 1u;  // the general path (deleted docs, raw norms, negative weights) decodes every block
      touched += encoded_block_bytes((uint32_t)seg.dir_hdr[T.dir_base + blk]) + (has_norms ? 128u : 0u);
      collect(std::true_type{}, d0, d1, f0, f1, nb0, nb1, true, true);
    };
    if (tabled && !has_live && nonneg) {
      // (the query's items: its head + the chunks behind it)
      const bool prune = RGPU_TERM_PRUNE && (T.flags & TERM_FLAG_MONOTONE) != 0u;
      // the term's block-max sketch: k real postings of k blocks, scored with this query's table — a threshold to start from
      // (first pass only: a deeper page collects below a ceiling)
      if (T.sketch != 0u && seg.sketch != nullptr && ceil == ~0ull && k <= TERM_SKETCH_K) {
        shared.fold(sketch_floor<WIDE>(seg.sketch + (size_t)(T.sketch - 1u) * TERM_SKETCH_K, cache, k, lane), tau, floor);
        touched += 2u * (uint32_t)k;
      }
      TERM_TR(tr_sketch, (uint32_t)floor);
      if (RGPU_TERM_WAIT && prune && chunk != 0) {  // (wave-uniform) the head's first publication, or the time-out
        uint64_t s2 = floor;
        for (int i = 0; i < RGPU_TERM_WAIT_POLLS && s2 == 0ull; ++i) {
          __builtin_amdgcn_s_sleep(RGPU_TERM_WAIT_SLEEP);
          const uint64_t g = shared.peek();
          s2 = ((uint64_t)(uint32_t)readfirstlane((int)(uint32_t)(g >> 32)) << 32) | (uint32_t)readfirstlane((int)(uint32_t)g);
        }
        shared.fold(s2, tau, floor);
      }
      term_blocks_fast<LEGACY, WIDE>(seg, T, b0, b1, slab, cache, wk, lane, group, shared, floor, k, count, prune, looked, touched, ceil,
                                     q_items <= TERM_EXCHANGE_MAX_ITEMS, chunk == 0 TERM_PH_PASS);
      if (b1 > b0) base = seg.dir_last[T.dir_base + b1 - 1];
    } else if (has_norms) {
      stream_blocks<LEGACY, true>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, seg.pnorm + T.pn_base, b0, b1, slab, lane, base, on_block);
    } else {
      stream_blocks<LEGACY, false>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, nullptr, b0, b1, slab, lane, base, on_block);
    }
    if (b1 == T.nblocks) {
      if (T.df == 1) {
        const bool v0 = lane == 0;
        const uint32_t nb0 = (has_norms && v0) ? norm_at(seg, T.singleton_doc) : 0u;
        collect(std::false_type{}, T.singleton_doc, 0, (uint32_t)T.singleton_freq, 1u, nb0, 0u, v0, false);
      } else if (T.tail_n > 0) {
        int32_t d0, d1;
        uint32_t f0, f1;
        tail_load(term_rows, seg.dir_row[T.dir_base + T.nblocks], lane, d0, d1, f0, f1);  // decoded and validated at prepare time
        const bool v0 = 2 * lane < T.tail_n, v1 = 2 * lane + 1 < T.tail_n;
        const uint32_t nb0 = (has_norms && v0) ? seg.norms[d0] : 0u, nb1 = (has_norms && v1) ? seg.norms[d1] : 0u;
        collect(std::false_type{}, d0, d1, f0, f1, nb0, nb1, v0, v1);
      }
    }
  }

#ifdef RGPU_TERM_TRACE
  if (lane == 0 && item < TERM_TRACE_CAP) g_term_trace[item] = TermTraceRec{trace_t0, (unsigned long long)wall_clock64(), q, chunk, (int32_t)looked, (int32_t)(touched / 256u), tr_term, tr_table, tr_sketch, 0u, {ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]}};
#endif
  // the wave of a group that finishes last emits the group's list; the other items emit empty lists
  int32_t count_seen = 0;
  if (lane == 0) {
    if (fold.done != nullptr) count_seen = __hip_atomic_exchange(partial_counts + item, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partial_counts[item] = count;
  }
  if (work_slots != nullptr && lane == 0 && (looked | touched) != 0u) {  // per query: one address for the whole launch would serialise thousands of wavefronts
    atomicAdd(work_slots + q, (unsigned long long)touched);
    atomicAdd(work_slots + n_queries + q, (unsigned long long)looked);
  }
  uint32_t done = 0;
  if (lane == 0) done = __hip_atomic_fetch_add(remaining + leader, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  done = (uint32_t)readfirstlane((int)done) + 1u;
  WaveTopK top;
  if (done == (uint32_t)members) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    top.a = group.keys[lane];
    if (WIDE) top.b = group.keys[64 + lane];
    shared.publish<WIDE>(top, k, lane);
  }
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (fold.done == nullptr) {
    if (lane < k) pk[lane] = top.a;
    if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
    return;
  }
  // (a release fence at agent scope here — buffer_wbl2: the whole L2's dirty lines, once per item — was measured: k_search_term 0.028 ->
  // 0.085 ms; agent-scope atomic STORES and a wait for their acknowledgement were built next and are not enough: the XCDs' L2s are
  // not coherent with one another inside a launch, and 5 launches in 3000 folded a stale list or count — scripts/fold_race_probe.py.
  // The list and the count are EXCHANGED in: read-modify-write atomics at agent scope are performed at the memory side, like the
  // counter below. Their returned values are waited for before the item counts itself.)
  unsigned long long seen_a = 0ull, seen_b = 0ull;
  if (lane < k) seen_a = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(pk) + lane, (unsigned long long)top.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (WIDE && lane + 64 < k) seen_b = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(pk) + lane + 64, (unsigned long long)top.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(seen_a), "v"(seen_b), "v"(count_seen) : "memory");  // (the exchanges have returned: they are performed)
  uint32_t finished = 0;
  if (lane == 0) finished = __hip_atomic_fetch_add(fold.done + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  finished = (uint32_t)readfirstlane((int)finished) + 1u;
  if (finished != (uint32_t)q_items) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // (ordering only: every read below is itself performed at the memory side)
  WaveTopK all;
  int64_t total = 0;
  merge_query_items<WIDE, true>(q, fold.item_prefix[q], (int64_t)q_items, n_queries, k, partial_keys, partial_counts, all, total, lane);
  const int row = qmap ? qmap[q] : q;
  HitOut* out = fold.hits + (size_t)row * (size_t)(fold.out_stride > 0 ? fold.out_stride : k) + fold.col0;
  if (fold.ceil_out != nullptr) {
    const uint64_t kth = topk_threshold<WIDE>(all, k);
    if (lane == 0) fold.ceil_out[row] = kth;
  }
  if (lane < k) out[lane] = all.a ? HitOut{key_doc(all.a) + fold.doc_base, key_score(all.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = all.b ? HitOut{key_doc(all.b) + fold.doc_base, key_score(all.b)} : HitOut{-1, 0.f};
  if (lane == 0) fold.totals[row] = total;
}

}  // namespace rgpu
