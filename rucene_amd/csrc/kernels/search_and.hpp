// Conjunction (AND) on the GPU, lead-driven: the data-parallel form of ConjunctionScorer's leapfrog
// (search/scorer/conjunction_scorer.rs:44-83). One wavefront owns a chunk of the cheapest clause's blocks; each
// decoded lead block yields 128 sorted candidates, two per lane. For every other clause, in cost order, the wave
//   1. keeps a 64-entry window of that clause's block directory {last doc, store row, header word} in registers
//      (three coalesced loads) and finds the block of the first pending candidate with one ballot — what
//      Lucene50SkipReader::skip_to does per probe (skip_reader.rs:554-584); every candidate up to that block's last
//      doc belongs to it, so no per-candidate search is needed;
//   2. visits only the blocks that hold at least one pending candidate, decoding each once (the next one's rows
//      are requested before the current one is unpacked: row addresses come from the register window, not from
//      dependent directory loads);
//   3. answers "is candidate d in this block" for all its candidates at once through a 2048-bit filter in LDS
//      (bit doc & 2047 per block doc); only filter hits are verified, one broadcast compare each, against the 128
//      decoded docs still in registers, which also yields the freq — the in-block scan of
//      BlockDocIterator::advance (posting_reader.rs:714-731). The freq stream is unpacked only for blocks with a hit.
// A candidate dies at the first clause that misses it, so later clauses touch fewer blocks. Scores are summed
// lead1, lead2, others... in f32 exactly as conjunction_scorer.rs:87-95 (the host sorts clauses by doc_freq).
#pragma once
#include "search.hpp"

namespace rgpu {

#ifdef RGPU_EXP_COUNT  // developer instrumentation (variant builds only): [0] lead blocks, [1] other-clause block decodes,
__device__ unsigned long long g_and_dbg[4];  // [2] probed candidates, [3] (lead block, clause) visits
#define AND_DBG(i, n) do { const unsigned long long n_ = (unsigned long long)(n); if (lane == 0) atomicAdd(&g_and_dbg[i], n_); } while (0)
#else
#define AND_DBG(i, n) do {} while (0)
#endif
#ifdef RGPU_AND_TIME  // developer instrumentation (variant builds only): wave-cycles per phase, summed over the launch's wavefronts
__device__ unsigned long long g_and_time[16];  // [0] item set-up [1] batched first probe [2] candidate loop on queued survivors
                                               // [3] candidate loop block by block [4] item epilogue [5] items [6] pops [7] block-by-block vectors
                                               // [8] queued survivors [9] walked-clause block decodes (popped) [10] ... (block by block)
#define AND_STAMP(t) const long long t = (long long)__builtin_readcyclecounter()
#define AND_TADD(i, v) and_t[i] += (long long)(v)
#else
#define AND_STAMP(t) do {} while (0)
#define AND_TADD(i, v) do {} while (0)
#endif

#ifdef RGPU_AND_TRACE  // developer instrumentation (variant builds only): every item's {start, end} wall clock (100 MHz), query and
constexpr int AND_TRACE_CAP = 1 << 17;            // chunk, survivors popped — the launch's timeline, read back by rgpu_debug_trace
struct AndTraceRec { unsigned long long t0, t1; int32_t q, chunk, lead_blocks, popped; };
__device__ AndTraceRec g_and_trace[AND_TRACE_CAP];
#endif
// Occupancy against registers (history): at 8 waves/SIMD (64 VGPRs, 80 SGPRs) the round-1 kernel spilled 90-120 bytes per lane
// to scratch; rounds 2-4 ran at 5 waves/SIMD (85-94 VGPRs, no scratch). Since round 5 (batched first probe: AND_G blocks of rows,
// 2 * AND_G gathers per lane in flight, a survivor queue in LDS) the kernel needs 99-111 VGPRs and 29.8 KB of LDS per workgroup:
// FOUR waves/SIMD (RGPU_AND_WAVES below), no scratch (scripts/kernel_resources.py). Its 102-121 "SGPR spills" are v_writelane /
// v_readlane into two VGPRs, not memory: 10-23 reloads per ~1200-instruction candidate loop.
#ifndef RGPU_AND_WAVES
#define RGPU_AND_WAVES 4
#endif
constexpr int AND_WAVES_PER_SIMD = RGPU_AND_WAVES;
// Wavefronts per WORKGROUP of k_search_and. Nothing in the kernel is shared between the wavefronts of a workgroup (every LDS
// array is indexed by the wavefront, no barrier is met), but a CU gives a workgroup's register and LDS allocation back only when
// its LAST wavefront ends: with items of 20 us at the median and 67 us at the 99th percentile (round 6's item timeline,
// scripts/and_timeline.py) three finished wavefronts wait for the fourth — the launch ran at 80 % of its 4096 wavefront slots.
#ifndef RGPU_AND_WG_WAVES
#define RGPU_AND_WG_WAVES 4
#endif
constexpr int AND_WG_WAVES = RGPU_AND_WG_WAVES;
constexpr int AND_WG_THREADS = 64 * AND_WG_WAVES;
#ifndef RGPU_AND_PREFETCH  // 1: the next block's rows are requested before the current one is unpacked (five more VGPRs)
#define RGPU_AND_PREFETCH 1
#endif
// Tried in round 3, measured on the 1024 x 3-term batch (1.51-1.53 ms before and after, every time), not kept:
//   * a "dense mode": when a directory window's pending candidates are about as many as the blocks they span, stream those
//     blocks in order and probe ONE 32768-bit filter of the candidates (built once per lead block and clause) with each
//     block's docs, confirmed hits leaving their freq in the candidate's LDS cell — 62 % of the block decodes went that
//     way, the kernel's time did not move (1.52 ms);
//   * clause descriptors in LDS, the next clause's directory window and the next lead block's rows requested a step early,
//     cursors parked at the LAST visited block: 103-123 VGPRs, four wavefronts per SIMD instead of five, 1.57-1.72 ms.
// Per (other-clause) block the kernel spends ~1000 SIMD cycles whichever way the block is chosen and probed; the ablation
// builds put 0.20 ms on the lead decode, 0.08 ms on clause set-up + directory windows, 0.77 ms on decoding the other
// clauses' blocks and 0.47 ms on the membership probe.
#ifndef RGPU_AND_ABL  // developer ablations (variant builds only; results are wrong): 1 lead decode only, 2 + clause setup
#define RGPU_AND_ABL 0  // and directory window, 3 + block decodes without the membership probe
#endif

// Wave-cooperative search (target is wave-uniform): first a coalesced look at the 64 directory entries right
// after `from` — consecutive lead blocks probe monotonically, so the answer is usually there (one load instead of
// ~14 dependent ones) — then a 64-ary search: each round the lanes probe 64 evenly spaced entries and a ballot
// picks the sub-range. Returns the first slot in [from, nblocks] whose last doc >= target.
__device__ __forceinline__ int find_block_wave(const int32_t* __restrict__ dir_last, uint32_t dir_base, int from, int nblocks,
                                               int32_t target, int lane) {
  if (from >= nblocks) return nblocks;
  {
    const int p = from + lane;
    const int32_t v = p < nblocks ? dir_last[dir_base + p] : 0x7fffffff;
    const uint64_t m = __ballot(v >= target);
    if (m) return min(from + (int)__builtin_ctzll(m), nblocks);
  }
  int lo = from + 64, hi = nblocks;  // invariant: every slot < lo is < target; answer in [lo, hi]
  while (hi - lo > 64) {
    const int stride = (hi - lo + 63) >> 6;
    const int p = min(lo + (lane + 1) * stride - 1, hi - 1);
    const uint64_t m = __ballot(dir_last[dir_base + p] >= target);
    if (m) {
      const int j = (int)__builtin_ctzll(m);
      hi = min(lo + (j + 1) * stride - 1, hi - 1);  // probe j is >= target: the answer is at or before it
      lo = lo + j * stride;                          // probes < j are < target
    } else {
      return hi;  // even slot hi - 1 is < target: the answer is hi (known >= target, or the virtual end slot)
    }
  }
  if (lo >= hi) return hi;
  const int p = lo + lane;
  const int32_t v = p < hi ? dir_last[dir_base + p] : 0x7fffffff;
  const uint64_t m = __ballot(v >= target);
  return m ? min(lo + (int)__builtin_ctzll(m), hi) : hi;
}

// A wave's register window over one clause's block directory: lane L holds entry `from - 1 + L`, so that block
// b = from + j - 1 (window slot j >= 1) finds its last doc in slot j and its base doc (the last doc of the block
// before it, 0 for block 0) in slot j - 1. Entries past the last FullBlock read as "last doc = INT_MAX": the virtual
// end slot (VInt tail, or nothing) bounds every doc.
struct DirWindow {
  int32_t last;
  uint32_t row, hdr;
  __device__ __forceinline__ void load(const SegView& seg, uint32_t dir_base, int nblocks, int from, int lane) {
    const int e = from - 1 + lane;
    const bool real = e >= 0 && e < nblocks;
    last = real ? seg.dir_last[dir_base + e] : (e < 0 ? 0 : 0x7fffffff);
    row = real ? seg.dir_row[dir_base + e] : 0u;
    hdr = real ? (uint32_t)seg.dir_hdr[dir_base + e] : 0u;
  }
};

constexpr int AND_FILTER_WORDS = 64;  // 2048-bit membership filter per wavefront

// Round 5: the batched first probe. The kernel was two to four DEPENDENT round trips per lead block (directory -> rows ->
// first clause's gather -> second clause's gather) at five wavefronts per SIMD: 65 % of its wave cycles waited. When the first
// clause behind the lead has a doc bitmap, an item now takes its lead blocks AND_G at a time: all their rows are requested
// together (row addresses from a register window over the lead's directory), each is unpacked while the next one's rows and the
// previous one's gathers are in flight, and the first clause is asked about all 128 * AND_G candidates at once — one gather
// per candidate into the list's membership bits (doc_bitmap.hpp `memb`: one bit per doc, the smallest footprint a probe can
// have; RGPU_AND_PROBE = 1 asks the four-bits-per-doc array instead — membership and freq in one gather, four times the bytes).
// What survives (a doc in five at most, usually far fewer) is appended, in doc order, to a wave-private LDS queue of {doc, norm
// byte | first-clause freq code | lead freq}; whenever 128 survivors have gathered (or the item ends) they are popped two per
// lane and take the candidate loop below — from the first clause again when their freqs are still unknown (it finds them all,
// with their freqs, on a sixth of the candidates), from the second clause on otherwise. The later clauses therefore see a tenth
// of the gathers, the BM25 divisions are only done for survivors, and the top-k list is offered full vectors. Same candidates, same f32 sums
// in the same order: bit-exact with the block-by-block path (which still serves a walked first clause, the lead's tail /
// singleton, ReqOptScorer records and lead freqs of 2^20 or more).
#ifndef RGPU_AND_FAST
#define RGPU_AND_FAST 1
#endif
#ifndef RGPU_AND_G
#define RGPU_AND_G 4
#endif
#ifndef RGPU_AND_GATHER_NT  // 1: the first probe's gathers as nontemporal loads (variant builds)
#define RGPU_AND_GATHER_NT 0
#endif
#ifndef RGPU_AND_PROBE  // what the batched first probe gathers from: 0 the membership bits, 1 the four-bits-per-doc array, 2 the {any, hi} pairs
#define RGPU_AND_PROBE 0
#endif
#ifndef RGPU_AND_PF  // 1: the next group's rows are requested before this group's gathers are waited for (4 * AND_G + AND_G more VGPRs, live across the candidate loop)
#define RGPU_AND_PF 0
#endif
constexpr int AND_G = RGPU_AND_G;
constexpr int AND_Q_CAP = 128 + 128 * AND_G;  // fewer than 128 entries wait when a group of AND_G blocks is appended
constexpr uint32_t AND_Q_FREQ_LIMIT = 1u << 20;  // a lead freq must fit the entry's 20 bits (Rucene clamps freqs to 10 at write time)

// What a MUST + SHOULD tree leaves per LEAD posting when the reference's ReqOptScorer rule is applied (k_req_opt_scan):
// the conjunction's matches in doc order (= lead posting order), each with its required and its optional sum. A lead
// posting that is no match (or a deleted doc) leaves doc = -1.
struct SeqRec {
  int32_t doc;
  float req, opt;
  int32_t pad;
};

// HAS_NOT / HAS_OPT: some query of the launch carries MUST_NOT / optional SHOULD clauses (separate instantiations keep the
// common kernel lean). Clause order on the device: [MUST x n_terms][MUST_NOT x pad][SHOULD x (op >> 16)].
template <bool LEGACY, bool WIDE, bool HAS_NOT, bool HAS_OPT>
__global__ __launch_bounds__(AND_WG_THREADS, AND_WAVES_PER_SIMD) void k_search_and(SegView seg, const DevQuery* __restrict__ queries,
                                                           const DevTerm* __restrict__ terms,
                                                           const int64_t* __restrict__ item_prefix, int n_queries,
                                                           int64_t n_items, int blocks_per_item, int k,
                                                           uint64_t* __restrict__ partial_keys,
                                                           int32_t* __restrict__ partial_counts,
                                                           unsigned long long* __restrict__ tau_slots,
                                                           unsigned long long* __restrict__ touched_slots,
                                                           const int64_t* __restrict__ emit_prefix,
                                                           unsigned long long* __restrict__ emit_count,
                                                           void* __restrict__ emit_out,
                                                           const unsigned long long* __restrict__ ceil_slots = nullptr,
                                                           const int32_t* __restrict__ qmap = nullptr,
                                                           const TermBitmap* __restrict__ bitmaps = nullptr, int xcd_chunk = 0) {
  // bitmaps (nullable, parallel to `terms`): a clause other than the lead whose term has a doc bitmap answers every candidate
  // with one bit of it (and, for a hit, the posting's rank and freq byte) instead of a walk through its blocks — a list
  // that holds a doc in five puts a candidate into nearly every one of its blocks between two lead postings. (Requesting the
  // first two bitmap clauses' words and ranks before the clause loop, so that their round trips overlap: 0.66 ms against
  // 0.51 — eight more registers, two of them spilled, and loads wasted on candidates the first clause kills. Asking the bitmap
  // clauses BEFORE the walked ones, their scores parked until the f32 sum reaches them: 0.54 — the lead blocks of a batch come
  // from queries whose lead is itself a dense list, and there every other clause has a bitmap already.)
  // emit_out != null: nothing is collected here. Without HAS_OPT (phrases): int32 doc ids appended to the query's list
  // at emit_prefix[q] in any order, emit_count[q] the cursor. With HAS_OPT (the exact ReqOptScorer rule): one SeqRec per
  // lead posting at emit_prefix[q] + the posting's ordinal — doc order, no cursor.
  int32_t* const emit_docs = HAS_OPT ? nullptr : static_cast<int32_t*>(emit_out);
  SeqRec* const seq_out = HAS_OPT ? static_cast<SeqRec*>(emit_out) : nullptr;
  __shared__ __attribute__((aligned(16))) uint8_t slabs[AND_WG_WAVES][2 * SLAB_STREAM];  // FullBlock staging only: tails arrive decoded
  __shared__ float caches[AND_WG_WAVES][256];
  __shared__ uint32_t filters[AND_WG_WAVES][AND_FILTER_WORDS];
#if RGPU_AND_FAST
  __shared__ uint2 queues[AND_WG_WAVES][AND_Q_CAP];  // survivors of the batched first probe: {doc, norm | code << 8 | lead freq << 12}
#endif
  const int lane = lane_id();
  const int wave = wave_id();
  // Workgroups go to the eight XCDs round-robin (workgroup b -> XCD b % 8), each XCD with its own 4 MB L2: with items taken in
  // launch order every XCD meets every query's bitmaps, and the probes' L2 hit rate is a third (rocprofv3: TCC_HIT 4.5 M of
  // 14.6 M requests per launch, the L1s stalled on pending misses 69 % of the time). Chunks of `xcd_chunk` consecutive
  // workgroups — neighbours in the host's order: the same query, or queries that probe the same list — are therefore dealt to the
  // XCDs whole: chunk c runs on XCD c % 8, so a list's bits are fetched into ONE L2 instead of eight. Measured on the 1024 x
  // 3-term batch at 10 M docs: 0.284 ms in launch order, 0.272 with chunks of 16, 0.259 with 64, 0.36 with 256 (the chunks'
  // costs differ: a long tail); at 100 M docs, where a list's bits are 12.5 MB and no L2 holds them: 2.28 / 2.31 / 2.49 / 2.54 —
  // so the host asks for chunks only while a list's bits fit an L2 (rgpu_api.hip and_xcd_chunk).
  int64_t wg = (int64_t)blockIdx.x;
  if (xcd_chunk > 0) {  // (the host launches whole rounds of 8 chunks)
    const int64_t r = wg >> 3, x = wg & 7;  // the r-th workgroup of XCD x
    wg = (r / xcd_chunk) * (8 * (int64_t)xcd_chunk) + x * xcd_chunk + (r % xcd_chunk);
  }
  const int64_t item = wg * AND_WG_WAVES + wave;
  if (item >= n_items) return;
#ifdef RGPU_AND_TIME
  long long and_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool and_popped = false;
#endif
  AND_STAMP(ts0);
#ifdef RGPU_AND_TRACE
  const unsigned long long trace_t0 = (unsigned long long)wall_clock64();
  int trace_popped = 0;
#endif
  const int q = upper_slot_wave(item_prefix, n_queries, item, lane);
  const int chunk = (int)(item - item_prefix[q]);
  const DevQuery Q = queries[q];
  const DevTerm L = terms[Q.first_term];
  uint8_t* slab = slabs[wave];
  float* cache = caches[wave];
  uint32_t* filt = filters[wave];
  const bool has_norms = seg.norms != nullptr;
  int cur_table = -1;
  float k1 = 0.f;
  auto use_table = [&](int id) {
    if (id != cur_table) { load_sim_table(seg, id, cache, lane, k1); cur_table = id; }
  };

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int count = 0;
  uint32_t touched = 0;  // encoded bytes of the FullBlocks this item decoded (SURVEY 8(d) "touched bytes"; scalar)
  uint32_t touched_blocks = 0;  // ... and their number
  auto block_bytes = [&](uint32_t hdr) -> uint32_t { touched_blocks += 1u; return encoded_block_bytes(hdr); };
  int cursor = 0;  // lane ti holds clause ti's directory cursor (a register array indexed by ti would spill)
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);
  const uint64_t ceil = ceil_slots != nullptr ? ceil_slots[qmap[q]] : ~0ull;  // k > 128: this pass's hits stay below it (wave.hpp)

  // nn: the candidates' norm bytes — from the lead's posting-order norms for FullBlocks, gathered for
  // its tail; every other clause scores the same docs, so no clause ever gathers norms again
  // ti_start = 2: the candidates come from the survivor queue — they ARE in the first clause, with freqs c1f0 / c1f1
  auto intersect = [&](int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nn, bool a0, bool a1, int32_t ord0, int ti_start, uint32_t c1f0,
                       uint32_t c1f1) {  // nn = norm byte 0 | norm byte 1 << 8; ord0 = ordinal of the wave's first lead posting
    // (phrase candidates keep their deleted docs — marked when they are emitted: the two-phase loop of bulk_scorer.rs counts
    // every approximation towards next_limit, live or not; the branch is scalar)
    const uint64_t* const live_here = (!HAS_OPT && emit_out != nullptr) ? nullptr : seg.live;
    a0 = a0 && doc_in_segment(seg, d0) && doc_is_live(live_here, d0);
    a1 = a1 && doc_in_segment(seg, d1) && doc_is_live(live_here, d1);
    use_table(L.sim_table);
    float wk = L.weight * (k1 + 1.0f);
    float s0 = bm25_score(wk, (float)(int32_t)f0, has_norms ? cache[nn & 0xffu] : k1);
    float s1 = bm25_score(wk, (float)(int32_t)f1, has_norms ? cache[nn >> 8] : k1);
    // clauses 1 .. n_terms-1 are required (MUST), the n_not after them prohibited (MUST_NOT: ReqNotScorer,
    // req_not_scorer.rs:47-63 — a candidate found there dies, nothing is scored), the n_opt after those optional
    // (SHOULD next to MUST: ReqOptScorer, req_opt_scorer.rs:41-66 — a candidate found there adds that clause's score to
    // a separate sum, as DisjunctionSumScorer does, which is added to the required sum at the end; a miss costs nothing.
    // The reference's sequential "skip the optional clause for low scorers after 100 docs" rule is NOT applied:
    // scores are the exact sums, >= the reference's)
    const int n_req_not = HAS_NOT ? Q.n_terms + Q.pad : Q.n_terms;
    // (HAS_OPT) behind the optional / nested group: the MUST clauses that ConjunctionScorer::score adds AFTER the nested child
    // (rgpu_api.hip search_pass: the children's stable cost order) — required like the first ones, added to (first sum + group sum)
    const int n_opt_end = HAS_OPT ? n_req_not + ((Q.op >> 16) & 0xff) : n_req_not;
    const int n_clauses = HAS_OPT ? n_opt_end + (int)((uint32_t)Q.op >> 26) : n_req_not;
    float r0 = 0.f, r1 = 0.f;  // required sums, parked while s0 / s1 collect the optional sum
    bool in_opt = false;
    // RGPU_OP_SHOULD_REQUIRED ("+a +(b c)": the SHOULD clauses are a DisjunctionSumScorer among the ConjunctionScorer's children,
    // boolean_query.rs:200-215): a candidate that none of them holds is no match. Bit 0 / 1: candidate 0 / 1 was found in one.
    const bool need_any = HAS_OPT && (Q.op & (1 << 24)) != 0;  // wave-uniform
    // RGPU_OP_NESTED_MUST ("+a +(+b +c)": the clauses behind the MUST clauses are a ConjunctionScorer of their own): a candidate that
    // one of them does not hold is no match; their sum is formed on its own, like the optional sum, and added last
    const bool need_all = HAS_OPT && (Q.op & (1 << 25)) != 0;  // wave-uniform
    uint32_t any_opt = 0u;
    if (RGPU_AND_FAST && ti_start == 2) {  // the first clause's score, added where the clause loop would have added it
      // (a one-clause prefix in front of a nested conjunction: clause 1 opens the nested group — its score starts the group's sum)
      if (HAS_OPT && need_all && n_req_not == 1) { r0 = s0; r1 = s1; s0 = 0.f; s1 = 0.f; in_opt = true; }
      const DevTerm T1 = terms[Q.first_term + 1];
      use_table(T1.sim_table);
      wk = T1.weight * (k1 + 1.0f);
      s0 += bm25_score(wk, (float)(int32_t)c1f0, has_norms ? cache[nn & 0xffu] : k1);
      s1 += bm25_score(wk, (float)(int32_t)c1f1, has_norms ? cache[nn >> 8] : k1);
    }
    for (int ti = RGPU_AND_FAST ? ti_start : 1; ti < n_clauses; ++ti) {
      if (!(__ballot(a0) | __ballot(a1))) break;
      const bool excl = HAS_NOT && ti >= Q.n_terms && ti < n_req_not;  // wave-uniform
      const bool opt = HAS_OPT && ti >= n_req_not && ti < n_opt_end;    // wave-uniform
      if (HAS_OPT && opt && !in_opt) { r0 = s0; r1 = s1; s0 = 0.f; s1 = 0.f; in_opt = true; }
      if (HAS_OPT && ti >= n_opt_end && in_opt) {  // the group's sum joins; the rest add to it
        s0 = r0 + s0; s1 = r1 + s1; in_opt = false;
        if (need_any) { a0 = a0 && (any_opt & 1u) != 0u; a1 = a1 && (any_opt & 2u) != 0u; }
      }
      const DevTerm T = terms[Q.first_term + ti];
      if (!excl) {
        use_table(T.sim_table);
        wk = T.weight * (k1 + 1.0f);
      }
      // what finding / missing a candidate in this clause means (the norm is looked up on the spot: finds are rare, and
      // two more registers held across the block loop are not)
      auto found = [&](uint32_t which, bool& alive, float& s, uint32_t fq, uint32_t nb) {
        if (excl) alive = false; else s += bm25_score(wk, (float)(int32_t)fq, has_norms ? cache[nb] : k1);
        if (HAS_OPT && opt) any_opt |= which;
      };
      const uint32_t n0 = nn & 0xffu, n1 = nn >> 8;
      auto missed = [&](bool& alive) { if (!excl && (!opt || need_all)) alive = false; };
      if (T.df == 1) {
        if (a0) { if (d0 == T.singleton_doc) found(1u, a0, s0, (uint32_t)T.singleton_freq, n0); else missed(a0); }
        if (a1) { if (d1 == T.singleton_doc) found(2u, a1, s1, (uint32_t)T.singleton_freq, n1); else missed(a1); }
        continue;
      }
      if (bitmaps != nullptr && bitmaps[Q.first_term + ti].words != nullptr) {
        const TermBitmap Bm = bitmaps[Q.first_term + ti];
        typedef const __attribute__((address_space(1))) uint32_t* gwords;
        typedef const __attribute__((address_space(1))) uint8_t* gbytes1;
        gwords words = (gwords)(uintptr_t)Bm.words;
        gwords ranks = (gwords)(uintptr_t)Bm.ranks;
        gbytes1 freqs = (gbytes1)(uintptr_t)Bm.freqs;
        if (Bm.nib != nullptr) {  // the densest lists: four bits per doc say "absent" or the freq — one gather, one round trip
          gwords nib = (gwords)(uintptr_t)Bm.nib;
          const uint32_t v0 = (nib[a0 ? (uint32_t)d0 >> 3 : 0u] >> (4 * (d0 & 7))) & 15u;
          const uint32_t v1 = (nib[a1 ? (uint32_t)d1 >> 3 : 0u] >> (4 * (d1 & 7))) & 15u;
          const bool g0 = a0 && v0 != 0u, g1 = a1 && v1 != 0u;
          touched += 4u * (uint32_t)(__popcll(__ballot(a0)) + __popcll(__ballot(a1)));
          if (!__ballot((g0 && v0 == 15u) || (g1 && v1 == 15u))) {
            if (a0) { if (g0) found(1u, a0, s0, v0, n0); else missed(a0); }
            if (a1) { if (g1) found(2u, a1, s1, v1, n1); else missed(a1); }
            continue;
          }
          // (some freq is 15 or more: the general probe below answers this clause for all of the lanes)
        }
        // (dead candidates look at word 0: unconditional loads, six in flight)
        const uint32_t i0 = a0 ? (uint32_t)d0 >> 5 : 0u, i1 = a1 ? (uint32_t)d1 >> 5 : 0u;
        const uint32_t w0 = words[2u * i0], w1 = words[2u * i1];
        const uint32_t r0w = ranks[i0], r1w = ranks[i1];
        const bool h0 = a0 && ((w0 >> (d0 & 31)) & 1u), h1 = a1 && ((w1 >> (d1 & 31)) & 1u);
        const uint32_t p0i = h0 ? r0w + (uint32_t)__popc(w0 & ((1u << (d0 & 31)) - 1u)) : 0u;
        const uint32_t p1i = h1 ? r1w + (uint32_t)__popc(w1 & ((1u << (d1 & 31)) - 1u)) : 0u;
        uint32_t fq0 = freqs[p0i], fq1 = freqs[p1i];
        if (__ballot((h0 && fq0 == 255u) || (h1 && fq1 == 255u))) {  // a clamped freq byte: the freq is in the overflow list
          for (int i = 0; i < Bm.n_ovf; ++i) {
            const uint32_t at = Bm.ovf[2 * i], f = Bm.ovf[2 * i + 1];
            if (h0 && fq0 == 255u && at == p0i) fq0 = f;
            if (h1 && fq1 == 255u && at == p1i) fq1 = f;
          }
        }
        touched += 8u * (uint32_t)(__popcll(__ballot(a0)) + __popcll(__ballot(a1))) + (uint32_t)(__popcll(__ballot(h0)) + __popcll(__ballot(h1)));
        if (a0) { if (h0) found(1u, a0, s0, fq0, n0); else missed(a0); }
        if (a1) { if (h1) found(2u, a1, s1, fq1, n1); else missed(a1); }
        continue;
      }
      AND_DBG(3, 1);
      if (RGPU_AND_ABL == 2) {
        DirWindow W2;
        W2.load(seg, T.dir_base, T.nblocks, min(readlane(cursor, ti), T.nblocks), lane);
        if (W2.last == 12345 && W2.row == 7 && W2.hdr == 9) count++;
        a0 = a1 = false;
        continue;
      }
      bool p0 = a0, p1 = a1;  // candidates this clause has not answered yet (sorted across (lane, slot))
      // doc of the first pending candidate; `any` = false when none is pending
      auto first_pending = [&](bool& any) -> int32_t {
        const uint64_t q0 = __ballot(p0), q1 = __ballot(p1);
        any = (q0 | q1) != 0;
        if (!any) return 0x7fffffff;
        const int l0 = q0 ? __builtin_ctzll(q0) : 64, l1 = q1 ? __builtin_ctzll(q1) : 64;
        return l0 <= l1 ? readlane(d0, l0 & 63) : readlane(d1, l1 & 63);
      };
      // Membership of the candidates c0 / c1 in the 128 (or, for a tail, `ev`-flagged) sorted docs e0 / e1 held two per
      // lane. `freqs()` yields the lanes' freqs and is called only when the filter reports a hit.
      auto probe = [&](int32_t e0, int32_t e1, bool ev0, bool ev1, bool c0, bool c1, auto freqs) {
        AND_DBG(2, __popcll(__ballot(c0)) + __popcll(__ballot(c1)));
        filt[lane] = 0u;
        wave_sync();
        if (ev0) atomicOr(&filt[((uint32_t)e0 >> 5) & (AND_FILTER_WORDS - 1)], 1u << (e0 & 31));
        if (ev1) atomicOr(&filt[((uint32_t)e1 >> 5) & (AND_FILTER_WORDS - 1)], 1u << (e1 & 31));
        wave_sync();
        const bool h0 = c0 && ((filt[((uint32_t)d0 >> 5) & (AND_FILTER_WORDS - 1)] >> (d0 & 31)) & 1u);
        const bool h1 = c1 && ((filt[((uint32_t)d1 >> 5) & (AND_FILTER_WORDS - 1)] >> (d1 & 31)) & 1u);
        if (c0 && !h0) missed(a0);
        if (c1 && !h1) missed(a1);
        const uint64_t k0 = __ballot(h0), k1m = __ballot(h1);
        if (k0 | k1m) {  // the filter has no false negatives; a hit is confirmed against the docs themselves
          uint32_t g0, g1;
          freqs(g0, g1);
          // One broadcast compare per hit only FETCHES the freq into the candidate's lane; the scoring (an IEEE
          // division) then runs once for all confirmed hits of the block together — doing it per hit, one lane at a
          // time, cost 100 VALU issue slots per block on the 3-term workload (rocprofv3, round 2).
          uint32_t fq0 = 0u, fq1 = 0u;
          auto collect = [&](uint64_t km, const int32_t dcand, uint32_t& fq) -> uint64_t {
            uint64_t got = 0;
            while (km) {
              const int j = __builtin_ctzll(km);
              km &= km - 1;
              const int32_t d = readlane(dcand, j);
              const uint64_t m0 = __ballot(ev0 && e0 == d), m1 = __ballot(ev1 && e1 == d);
              if (m0 | m1) {
                const uint32_t f = m0 ? (uint32_t)readlane((int)g0, __builtin_ctzll(m0)) : (uint32_t)readlane((int)g1, __builtin_ctzll(m1));
                fq = lane == j ? f : fq;
                got |= 1ull << j;
              }
            }
            return got;
          };
          const uint64_t got0 = collect(k0, d0, fq0), got1 = collect(k1m, d1, fq1);
          if (h0) { if ((got0 >> lane) & 1ull) found(1u, a0, s0, fq0, n0); else missed(a0); }
          if (h1) { if ((got1 >> lane) & 1ull) found(2u, a1, s1, fq1, n1); else missed(a1); }
        }
        wave_sync();  // the filter is rewritten by the next block
      };

      const uint8_t* term_rows = seg.bstore + T.bs_base;
      int from = min(readlane(cursor, ti), T.nblocks);
      DirWindow W;
      W.load(seg, T.dir_base, T.nblocks, from, lane);
      // window slot (>= 1) of the block that holds the first pending candidate, moving the window when that block lies
      // beyond it; -1 when nothing is pending
      auto locate = [&]() -> int {
        bool any;
        const int32_t d = first_pending(any);
        if (!any) return -1;
        while (true) {
          const uint64_t mb = __ballot(W.last >= d) & ~1ull;
          if (mb) return (int)__builtin_ctzll(mb);
          from = find_block_wave(seg.dir_last, T.dir_base, from + 63, T.nblocks, d, lane);
          W.load(seg, T.dir_base, T.nblocks, from, lane);
        }
      };
      bool first_visit = true;
      while (true) {
        int j = locate();
        if (j < 0) break;
        if (first_visit) {  // this clause's cursor only moves forward: lead blocks of an item arrive in doc order
          cursor = lane == ti ? from + j - 1 : cursor;
          first_visit = false;
        }
        if (from + j - 1 >= T.nblocks) {  // candidates past the last FullBlock: the VInt tail, or nothing
          if (T.tail_n > 0) {
            int32_t e0, e1;
            uint32_t g0, g1;
            tail_load(term_rows, seg.dir_row[T.dir_base + T.nblocks], lane, e0, e1, g0, g1);  // decoded and validated at prepare time
            probe(e0, e1, 2 * lane < T.tail_n, 2 * lane + 1 < T.tail_n, p0, p1, [&](uint32_t& x0, uint32_t& x1) { x0 = g0; x1 = g1; });
          } else {
            if (p0) missed(a0);
            if (p1) missed(a1);
          }
          break;
        }
        // FullBlocks inside the window, software-pipelined: the NEXT block's rows are requested (addresses from the
        // register window) before the current one is unpacked and probed
        struct Fetched { uint4 rows; uint32_t hdr; };
        auto fetch = [&](int slot) -> Fetched {
          Fetched f;
          f.hdr = (uint32_t)readlane((int)W.hdr, slot);
          f.rows = block_rows_load(block_rows_at(term_rows, (uint32_t)readlane((int)W.row, slot)), f.hdr, lane);
          return f;
        };
        Fetched A = fetch(j);
        while (true) {
          const int32_t vlast = readlane(W.last, j), vbase = readlane(W.last, j - 1);
          const bool c0 = p0 && d0 <= vlast, c1 = p1 && d1 <= vlast;  // every pending candidate up to the block's last doc
          p0 = p0 && !c0;
          p1 = p1 && !c1;
          int jn = 0;  // the next pending candidate's block, if it lies in this window (else: leave the pipeline)
          {
            bool any;
            const int32_t d = first_pending(any);
            const uint64_t mb = __ballot(W.last >= d) & ~1ull;
            if (any && mb) {
              jn = (int)__builtin_ctzll(mb);
              if (from + jn - 1 >= T.nblocks) jn = 0;
            }
          }
          const bool more = jn > 0;
#if RGPU_AND_PREFETCH
          const Fetched B = fetch(more ? jn : j);  // unconditional: a load behind a branch would serialise the two
#endif
          stage_rows(A.rows, slab, lane);
          wave_sync();
          touched += block_bytes(A.hdr);
          AND_DBG(1, 1);
#ifdef RGPU_AND_TIME
          and_t[and_popped ? 9 : 10] += 1;
#endif
          uint32_t x0, x1;
          staged_doc_deltas<LEGACY>(slab, A.rows, A.hdr, lane, x0, x1);
          int32_t e0, e1;
          deltas_to_docs(x0, x1, vbase, e0, e1);
          if (RGPU_AND_ABL == 3) { if (c0 && e0 == d0) count++; if (c0) a0 = false; if (c1) a1 = false; wave_sync(); }
          else
          probe(e0, e1, true, true, c0, c1, [&](uint32_t& g0, uint32_t& g1) { staged_freqs<LEGACY>(slab, A.rows, A.hdr, lane, g0, g1); });
          if (!more) break;
          j = jn;
#if RGPU_AND_PREFETCH
          A = B;
#else
          A = fetch(j);
#endif
        }
      }
    }
    if (HAS_OPT && need_any) {
      a0 = a0 && (any_opt & 1u) != 0u;
      a1 = a1 && (any_opt & 2u) != 0u;
    }
    if (HAS_OPT && seq_out != nullptr) {
      const int32_t ord = ord0 + 2 * lane;  // ordinals below the lead's doc_freq are postings (a tail fills only some lanes)
      SeqRec* rec = seq_out + emit_prefix[q] + ord;
      if (ord < L.df) rec[0] = SeqRec{a0 ? d0 : -1, in_opt ? r0 : s0, in_opt ? s0 : 0.f, 0};
      if (ord + 1 < L.df) rec[1] = SeqRec{a1 ? d1 : -1, in_opt ? r1 : s1, in_opt ? s1 : 0.f, 0};
      return;
    }
    if (HAS_OPT && in_opt) { s0 = r0 + s0; s1 = r1 + s1; }
    if (emit_docs != nullptr) {
      // Phrase queries (search_phrase.hpp): the conjunction's matches are not collected but handed on, every one of them,
      // to the position check — appended to the query's candidate list in any order (the phrase kernels are order-free).
      const uint64_t m0 = __ballot(a0), m1 = __ballot(a1);
      const int n0c = __popcll(m0), nc = n0c + __popcll(m1);
      if (nc) {
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(emit_count + q, (unsigned long long)nc);
        const int64_t base_at = emit_prefix[q] + (int64_t)readlane64(at, 0);
        // a deleted doc travels with its sign bit set (search_phrase.hpp PHRASE_DEAD): no position check, but it counts as an approximation
        if (a0) emit_docs[base_at + mbcnt(m0)] = doc_is_live(seg.live, d0) ? d0 : (d0 | (int32_t)0x80000000);
        if (a1) emit_docs[base_at + n0c + mbcnt(m1)] = doc_is_live(seg.live, d1) ? d1 : (d1 | (int32_t)0x80000000);
      }
      return;
    }
    count += __popcll(__ballot(a0)) + __popcll(__ballot(a1));
    topk_offer<WIDE>(top, a0 ? below(make_key(s0, d0), ceil) : 0ull, tau, k, lane, floor);
    topk_offer<WIDE>(top, a1 ? below(make_key(s1, d1), ceil) : 0ull, tau, k, lane, floor);
  };

  // One call site for the shapes a vector of candidates can take (FullBlock, VInt tail, singleton, 128 queued survivors): the
  // candidate loop above is large, and inlining it three times tripled the kernel's code and its register pressure.
  const int b0 = chunk * blocks_per_item;
  const int b1 = min(L.nblocks, b0 + blocks_per_item);
  const bool with_rest = b1 == L.nblocks && (L.df == 1 || L.tail_n > 0);  // this item also takes the tail / singleton
  const int b_end = b1 + (with_rest ? 1 : 0);
#if RGPU_AND_FAST
  const uint8_t* const lead_rows = seg.bstore + L.bs_base;
  // posting-order norms of the lead (any readable address when the segment has none: the load is unconditional, its value unused)
  const uint8_t* const lead_pn = has_norms ? seg.pnorm + L.pn_base : lead_rows;
  // the batched first probe needs: a doc bitmap for the first clause behind the lead, that clause required (not MUST_NOT / SHOULD),
  // a collector that takes matches in vectors (not ReqOptScorer's one-record-per-lead-posting form)
  // (... or the first clause of a nested conjunction right behind a one-clause prefix: RGPU_OP_NESTED_MUST, no MUST_NOT clause between)
  const bool c1_nested = HAS_OPT && Q.n_terms == 1 && (!HAS_NOT || Q.pad == 0) && (Q.op & (1 << 25)) != 0;
  bool fast = b1 > b0 && bitmaps != nullptr && (Q.n_terms >= 2 || c1_nested) && !(HAS_OPT && seq_out != nullptr);
  typedef const __attribute__((address_space(1))) uint32_t* gwords1;
  gwords1 probe_src = nullptr;
  int probe_nib = 0;  // 1: four bits per doc {absent, freq 1..14, 15 = look it up}; 0: one membership bit per doc
  int probe_mul = 1;  // ... in every word (the membership bits) or in every other one (the {any, hi} pairs)
  if (fast) {
    const TermBitmap B1 = bitmaps[Q.first_term + 1];
    // (words == null, memb != null: a list below the full bitmaps' density that carries its membership bits alone — the probe
    // asks them, the popped survivors find their freq through the clause's block directory in the candidate loop)
    fast = B1.words != nullptr || B1.memb != nullptr;
    // RGPU_AND_PROBE: 0 = the membership bits (one per doc; a survivor's freq is asked for when it is popped), 1 = the
    // four-bits-per-doc array where there is one (membership and freq in one gather, four times the footprint), 2 = the {any, hi} pairs
    probe_nib = (B1.nib != nullptr && RGPU_AND_PROBE == 1) ? 1 : 0;
    probe_mul = (probe_nib || ((RGPU_AND_PROBE == 0 || B1.words == nullptr) && B1.memb != nullptr)) ? 1 : 2;
    probe_src = (gwords1)(uintptr_t)(probe_nib ? (const void*)B1.nib : (probe_mul == 1 ? (const void*)B1.memb : (const void*)B1.words));
  }
  uint2* const queue = queues[wave];
  int qhead = 0, qtail = 0;
  int lw0 = b0;  // LW: a register window over the lead's directory — slot j >= 1 = block lw0 + j - 1, its base doc in slot j - 1
  DirWindow LW;
  LW.load(seg, L.dir_base, L.nblocks, lw0, lane);
#if RGPU_AND_PF
  // the NEXT group's rows and norms, requested while this group's gathers are in flight (pf_blk: the block they start at, -1: none)
  uint4 pf_rows[AND_G];
  uint32_t pf_nn[AND_G];
  int pf_blk = -1;
#endif
  int blk = b0;
  AND_STAMP(ts1);
  AND_TADD(0, ts1 - ts0);
  AND_TADD(5, 1);
  while (true) {
    AND_STAMP(tl0);
    if (blk < b1 && blk - lw0 > 63 - AND_G) { lw0 = blk; LW.load(seg, L.dir_base, L.nblocks, lw0, lane); }
    if (fast && blk < b1 && qtail - qhead < 128) {
      // ---- AND_G lead blocks: rows requested together, unpacked one after the other, 2 * AND_G gathers per lane in flight ----
      const int qn = qtail - qhead;
      if (qhead > 0) {  // what waits moves to the front (fewer than 128 entries)
        const uint2 m0 = lane < qn ? queue[qhead + lane] : make_uint2(0u, 0u);
        const uint2 m1 = lane + 64 < qn ? queue[qhead + 64 + lane] : make_uint2(0u, 0u);
        wave_sync();
        if (lane < qn) queue[lane] = m0;
        if (lane + 64 < qn) queue[lane + 64] = m1;
        wave_sync();
        qhead = 0;
        qtail = qn;
      }
      const int nb = min(AND_G, b1 - blk);
      uint4 rows[AND_G];
      uint32_t nnv[AND_G], hdrs[AND_G];
#if RGPU_AND_PF
      const bool have_pf = pf_blk == blk;  // wave-uniform
#endif
#pragma unroll
      for (int g = 0; g < AND_G; ++g) {  // (indices clamped, not guarded: a repeated block costs less than a load behind a branch)
        const int bi = min(blk + g, b1 - 1);
        const int j = bi - lw0 + 1;
        hdrs[g] = (uint32_t)readlane((int)LW.hdr, j);
#if RGPU_AND_PF
        if (have_pf) { rows[g] = pf_rows[g]; nnv[g] = pf_nn[g]; continue; }
#endif
        rows[g] = block_rows_load(block_rows_at(lead_rows, (uint32_t)readlane((int)LW.row, j)), hdrs[g], lane);
        nnv[g] = *reinterpret_cast<const uint16_t*>(lead_pn + (128u * (uint32_t)bi + 2u * (uint32_t)lane));
      }
      int32_t D[2 * AND_G];
      uint32_t P[2 * AND_G], V[2 * AND_G];
      bool too_wide = false;
      const int sh = probe_nib ? 3 : 5, mul = probe_mul;
#pragma unroll
      for (int g = 0; g < AND_G; ++g) {
        stage_rows(rows[g], slab, lane);
        wave_sync();
        uint32_t x0, x1, y0, y1;
        staged_doc_deltas<LEGACY>(slab, rows[g], hdrs[g], lane, x0, x1);
        staged_freqs<LEGACY>(slab, rows[g], hdrs[g], lane, y0, y1);
        wave_sync();  // the slab is free for the next block
        const int bi = min(blk + g, b1 - 1);
        deltas_to_docs(x0, x1, readlane(LW.last, bi - lw0), D[2 * g], D[2 * g + 1]);
        too_wide = too_wide || __ballot(y0 >= AND_Q_FREQ_LIMIT || y1 >= AND_Q_FREQ_LIMIT) != 0;
        const uint32_t n2 = has_norms ? nnv[g] : 0u;
        P[2 * g] = (n2 & 0xffu) | (y0 << 12);
        P[2 * g + 1] = (n2 >> 8) | (y1 << 12);
        uint32_t ix0 = ((uint32_t)D[2 * g] >> sh) * (uint32_t)mul, ix1 = ((uint32_t)D[2 * g + 1] >> sh) * (uint32_t)mul;
        if (RGPU_AND_ABL == 7) { ix0 &= 1023u; ix1 &= 1023u; }  // (variant builds: every probe inside one 4 KB window — what the gathers' misses cost)
#if RGPU_AND_GATHER_NT
        V[2 * g] = __builtin_nontemporal_load(probe_src + ix0);
        V[2 * g + 1] = __builtin_nontemporal_load(probe_src + ix1);
#else
        V[2 * g] = probe_src[ix0];
        V[2 * g + 1] = probe_src[ix1];
#endif
      }
      if (too_wide) {  // a lead freq that does not fit a queue entry: nothing of this group is kept, the rest of the item goes block by block
        fast = false;
        continue;
      }
#if RGPU_AND_PF
      {  // the next group's rows: they travel while this group's gathers come back and its survivors are queued (and, when 128
         // survivors are waiting, while those go through the later clauses). Blocks clamped into the directory window and the item.
        const int nb0 = blk + nb;
        const bool ok = nb0 < b1 && nb0 + AND_G - 1 - lw0 + 1 <= 63;
#pragma unroll
        for (int g = 0; g < AND_G; ++g) {
          const int bi = min(min(nb0 + g, b1 - 1), lw0 + 62);
          const int j = bi - lw0 + 1;
          const uint32_t h = (uint32_t)readlane((int)LW.hdr, j);
          pf_rows[g] = block_rows_load(block_rows_at(lead_rows, (uint32_t)readlane((int)LW.row, j)), h, lane);
          pf_nn[g] = *reinterpret_cast<const uint16_t*>(lead_pn + (128u * (uint32_t)bi + 2u * (uint32_t)lane));
        }
        pf_blk = ok ? nb0 : -1;
      }
#endif
#pragma unroll
      for (int g = 0; g < AND_G; ++g) {
        if (g < nb) {  // wave-uniform
          const int32_t e0 = D[2 * g], e1 = D[2 * g + 1];
          const uint32_t c0 = probe_nib ? (V[2 * g] >> (4 * (e0 & 7))) & 15u : ((V[2 * g] >> (e0 & 31)) & 1u) * 15u;
          const uint32_t c1 = probe_nib ? (V[2 * g + 1] >> (4 * (e1 & 7))) & 15u : ((V[2 * g + 1] >> (e1 & 31)) & 1u) * 15u;
          const uint64_t m0 = __ballot(c0 != 0u), m1 = __ballot(c1 != 0u);
          const int at = qtail + mbcnt(m0) + mbcnt(m1);  // doc order: (lane, slot 0), (lane, slot 1), (lane + 1, slot 0) ...
          if (c0 != 0u) queue[at] = make_uint2((uint32_t)e0, P[2 * g] | (c0 << 8));
          if (c1 != 0u) queue[at + (c0 != 0u ? 1 : 0)] = make_uint2((uint32_t)e1, P[2 * g + 1] | (c1 << 8));
          qtail += __popcll(m0) + __popcll(m1);
          touched += block_bytes(hdrs[g]) + 4u * 128u;
          AND_DBG(0, 1);
        }
      }
      wave_sync();
      blk += nb;
      {
        AND_STAMP(tl1);
        AND_TADD(1, tl1 - tl0);
      }
      continue;
    }
    int32_t d0, d1;
    uint32_t f0, f1, nn = 0u, c1f0 = 0u, c1f1 = 0u;
    bool a0, a1;
    int ti_start = 1;
    int32_t ord0 = 0;
    if (qtail > qhead) {  // up to 128 survivors of the first probe, two per lane in doc order
      const int n = min(128, qtail - qhead);
      a0 = 2 * lane < n;
      a1 = 2 * lane + 1 < n;
      const uint2 e0 = a0 ? queue[qhead + 2 * lane] : make_uint2(0u, 0u);
      const uint2 e1 = a1 ? queue[qhead + 2 * lane + 1] : make_uint2(0u, 0u);
      wave_sync();
      qhead += n;
#ifdef RGPU_AND_TRACE
      trace_popped += n;
#endif
      d0 = (int32_t)e0.x; d1 = (int32_t)e1.x;
      f0 = e0.y >> 12; f1 = e1.y >> 12;
      nn = (e0.y & 0xffu) | ((e1.y & 0xffu) << 8);
      c1f0 = (e0.y >> 8) & 15u; c1f1 = (e1.y >> 8) & 15u;
      // a code of 15 = "in the list, freq not in the probe's word": the clause loop asks the first clause again (it finds them all)
      ti_start = __ballot((a0 && c1f0 == 15u) || (a1 && c1f1 == 15u)) ? 1 : 2;
#ifdef RGPU_AND_TIME
      and_popped = true;
      and_t[6] += 1;
      and_t[8] += n;
#endif
    } else if (blk < b_end) {
#ifdef RGPU_AND_TIME
      and_popped = false;
      and_t[7] += 1;
#endif
      if (blk < L.nblocks) {
        const int j = blk - lw0 + 1;
        const uint32_t lhdr = (uint32_t)readlane((int)LW.hdr, j);
        if (has_norms) nn = *reinterpret_cast<const uint16_t*>(lead_pn + (128u * (uint32_t)blk + 2u * (uint32_t)lane));
        const BlockPair bp = decode_block<LEGACY>(lead_rows, (uint32_t)readlane((int)LW.row, j), lhdr, slab, lane);
        touched += block_bytes(lhdr);
        deltas_to_docs(bp.d0, bp.d1, readlane(LW.last, j - 1), d0, d1);
        f0 = bp.f0; f1 = bp.f1;
        a0 = true; a1 = true;
        ord0 = 128 * blk;
        AND_DBG(0, 1);
      } else if (L.df == 1) {
        d0 = d1 = L.singleton_doc;
        f0 = (uint32_t)L.singleton_freq; f1 = 0u;
        a0 = lane == 0; a1 = false;
        if (has_norms && a0) nn = norm_at(seg, d0);
      } else {
        tail_load(lead_rows, seg.dir_row[L.dir_base + L.nblocks], lane, d0, d1, f0, f1);
        a0 = 2 * lane < L.tail_n; a1 = 2 * lane + 1 < L.tail_n;
        if (has_norms && a0) nn = norm_at(seg, d0);
        if (has_norms && a1) nn |= norm_at(seg, d1) << 8;
        ord0 = 128 * L.nblocks;
      }
      ++blk;
    } else {
      break;
    }
    if (RGPU_AND_ABL >= 4) {  // (variant builds: what the vectors of either kind cost — results are wrong)
      const bool popped_vec = ti_start == 2 || c1f0 != 0u || c1f1 != 0u;
      if ((RGPU_AND_ABL == 4 && !popped_vec) || (RGPU_AND_ABL == 5 && popped_vec) || RGPU_AND_ABL == 6) { count += (a0 && d0 == 12345) ? 1 : 0; continue; }
    }
    intersect(d0, d1, f0, f1, nn, a0, a1, ord0, ti_start, c1f0, c1f1);
#ifdef RGPU_AND_TIME
    {
      AND_STAMP(tl2);
      and_t[and_popped ? 2 : 3] += tl2 - tl0;
    }
#endif
  }
#else
  int32_t base = b0 == 0 ? 0 : seg.dir_last[L.dir_base + b0 - 1];
  for (int blk = b0; blk < b_end; ++blk) {
    int32_t d0, d1;
    uint32_t f0, f1, nn = 0u;
    bool a0, a1;
    if (blk < L.nblocks) {
      if (has_norms) nn = *reinterpret_cast<const uint16_t*>(seg.pnorm + L.pn_base + 128 * (size_t)blk + 2 * lane);
      const uint32_t lhdr = seg.dir_hdr[L.dir_base + blk];
      const BlockPair bp = decode_block<LEGACY>(seg.bstore + L.bs_base, seg.dir_row[L.dir_base + blk], lhdr, slab, lane);
      touched += block_bytes(lhdr);
      deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
      base = readlane(d1, 63);
      f0 = bp.f0; f1 = bp.f1;
      a0 = true; a1 = true;
      AND_DBG(0, 1);
    } else if (L.df == 1) {
      d0 = d1 = L.singleton_doc;
      f0 = (uint32_t)L.singleton_freq; f1 = 0u;
      a0 = lane == 0; a1 = false;
      if (has_norms && a0) nn = norm_at(seg, d0);
    } else {
      tail_load(seg.bstore + L.bs_base, seg.dir_row[L.dir_base + L.nblocks], lane, d0, d1, f0, f1);
      a0 = 2 * lane < L.tail_n; a1 = 2 * lane + 1 < L.tail_n;
      if (has_norms && a0) nn = norm_at(seg, d0);
      if (has_norms && a1) nn |= norm_at(seg, d1) << 8;
    }
    if (RGPU_AND_ABL == 1) { if (a0 && d0 == 12345 && f0 == 77 && nn == 3) count++; continue; }
    intersect(d0, d1, f0, f1, nn, a0, a1, blk < L.nblocks ? 128 * blk : (L.df == 1 ? 0 : 128 * L.nblocks), 1, 0u, 0u);
  }
#endif
  AND_STAMP(te0);
  shared.publish<WIDE>(top, k, lane);
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
#ifdef RGPU_AND_TRACE
  if (lane == 0 && item < AND_TRACE_CAP) g_and_trace[item] = AndTraceRec{trace_t0, (unsigned long long)wall_clock64(), q, chunk, b1 - b0, trace_popped};
#endif
  if (lane == 0) {
    partial_counts[item] = count;
    atomicAdd(touched_slots + q, (unsigned long long)touched);  // ~160 items per query word: no contention to speak of
    atomicAdd(touched_slots + n_queries + q, (unsigned long long)touched_blocks);
  }
#ifdef RGPU_AND_TIME
  {
    AND_STAMP(te1);
    and_t[4] += te1 - te0;
    if (lane == 0)
      for (int i = 0; i < 11; ++i) atomicAdd(&g_and_time[i], (unsigned long long)and_t[i]);
  }
#endif
}

// ReqOptScorer::score (search/scorer/req_opt_scorer.rs:41-66) over a query's matches in doc order, one wavefront per query.
// The scorer carries state from doc to doc: the sum and the number of the required scores of the docs that took the
// optional path; once more than 100 did, a doc whose required score is under half their mean returns the required score
// alone and leaves the state untouched. An f32 running sum in iteration order cannot be re-associated, so the matches
// of a query are walked one by one (the 64 records of a step are loaded together, the decisions are scalar: one add,
// one divide, one compare per match) — a few ms per million matches, the price of the reference's exact scores for
// MUST + SHOULD trees (rgpu_config.req_opt_rule = -1 adds the optional sums everywhere instead, in the conjunction
// kernel itself). The collector part is the usual one: every match counts, keys go to the wavefront's top-k.
constexpr int REQ_OPT_THRESHOLD = 100;  // OPT_SCORE_THRESHOLD (req_opt_scorer.rs:19)
template <bool WIDE>
__global__ __launch_bounds__(WG_THREADS) void k_req_opt_scan(const SeqRec* __restrict__ seq, const int64_t* __restrict__ seq_prefix,
                                                             int n_queries, int k, int32_t doc_base, const int32_t* __restrict__ qmap,
                                                             HitOut* __restrict__ hits_out, int64_t* __restrict__ totals_out, int out_stride = 0,
                                                             int col0 = 0, const unsigned long long* __restrict__ ceil_slots = nullptr,
                                                             unsigned long long* __restrict__ ceil_out = nullptr) {
  const int lane = lane_id();
  const int q = (int)(blockIdx.x * WG_WAVES) + wave_id();
  if (q >= n_queries) return;
  const int64_t i0 = seq_prefix[q], i1 = seq_prefix[q + 1];
  const int row = qmap ? qmap[q] : q;
  const uint64_t ceil = ceil_slots != nullptr ? ceil_slots[row] : ~0ull;
  WaveTopK top;
  uint64_t tau = 0;
  int64_t total = 0;
  float scores_sum = 0.0f;  // wave-uniform
  int scores_num = 0;
  for (int64_t at = i0; at < i1; at += 64) {
    SeqRec r = SeqRec{-1, 0.f, 0.f, 0};
    if (at + lane < i1) r = seq[at + lane];
    const bool valid = r.doc >= 0;
    uint64_t m = __ballot(valid);
    total += __popcll(m);
    uint64_t skipped = 0;
    while (m) {  // in doc order
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      const float score = __int_as_float(readlane(__float_as_int(r.req), l));
      bool skip = false;
      if (scores_num > REQ_OPT_THRESHOLD) skip = 2.0f * score < scores_sum / (float)scores_num;
      if (skip) {
        skipped |= 1ull << l;
      } else {
        scores_sum += score;
        scores_num += 1;
      }
    }
    const float final_score = ((skipped >> lane) & 1ull) ? r.req : r.req + r.opt;
    topk_offer<WIDE>(top, valid ? below(make_key(final_score, r.doc), ceil) : 0ull, tau, k, lane);
  }
  HitOut* out = hits_out + (size_t)row * (size_t)(out_stride > 0 ? out_stride : k) + col0;
  if (ceil_out != nullptr) {
    const uint64_t kth = topk_threshold<WIDE>(top, k);
    if (lane == 0) ceil_out[row] = kth;
  }
  if (lane < k) out[lane] = top.a ? HitOut{key_doc(top.a) + doc_base, key_score(top.a)} : HitOut{-1, 0.f};
  if (WIDE && lane + 64 < k) out[lane + 64] = top.b ? HitOut{key_doc(top.b) + doc_base, key_score(top.b)} : HitOut{-1, 0.f};
  if (lane == 0) totals_out[row] = total;
}

}  // namespace rgpu
