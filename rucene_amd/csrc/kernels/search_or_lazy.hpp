// Disjunctions of TEN OR MORE SHOULD clauses whose densest lists are met through their doc bitmaps: k_or_lazy.
//
// k_or_wide (search_or_wide.hpp) decodes and scores every posting of every clause inside its window loop; on a Zipfian
// ten-term query ~90 % of those postings sit in two or three head terms whose scores are the smallest in the query — the lists
// that MaxScore-style pruning (Turtle & Flood) calls non-essential. This kernel never walks them:
//   * a clause whose term has a doc bitmap (doc_bitmap.hpp: a list holding >= 1 doc in 64) is LAZY: its postings are not
//     decoded. What a collector needs of it is (a) membership — TopDocs::total_hits counts every doc of the union — and
//     (b) its score on the few docs that can still enter the top-k;
//   * the other clauses are WALKED: k_score_terms (search_or.hpp) decodes and scores each distinct (term, weight) of the
//     batch once into a {doc, score} run — they hold the remaining ~10 % of the postings — and each wavefront walks the runs
//     through its own WINDOW of W docs from one cursor per clause (lane t = clause t). Walked postings are sparse in doc
//     space (<= 1/64 per clause), so a window's accumulators are indexed by RANK, not by doc: pass 1 sets one bit per
//     posting in the window's `touched` bitmap, a prefix popcount numbers the touched docs, pass 2 adds each posting's
//     fixed-point score into cell rank(doc) (the run entries of pass 1 are still in registers). A 16384-doc window costs 4 KB
//     of LDS for bitmap + numbering and 6 B per TOUCHED doc instead of the 64 KB of k_or_wide: every wavefront owns one,
//     nothing is shared inside the window loop and no barrier is met;
//   * total_hits: popcount(touched | OR of the lazy clauses' bitmap words), 2048 docs per instruction;
//   * a touched doc outside every lazy list has its exact total in its cell. One inside some lazy list is a candidate only
//     if cell + (sum of the lazy clauses' score bounds) reaches the current k-th best total; candidates queue up and are
//     evaluated 64 at a time: the lazy clauses' bitmap words say which of them hold the doc (the bound tightens to the
//     clauses actually present), ranks[] + a popcount give the posting's index, freqs[] its freq, the doc's norm the rest;
//   * a doc in lazy lists ONLY can reach the top-k only while the lazy clauses' bounds together reach the k-th best total:
//     then the docs held by enough lazy lists (a bit-sliced count of the bitmap words against the smallest m whose m
//     largest bounds reach the threshold) are candidates too.
//   * the k-th best total a pruning decision compares with is the best lower bound the QUERY has, not the wavefront: every
//     finished total at or above the wavefront's threshold is counted in a per-query histogram (128 buckets of 2^24
//     fixed-point steps, one atomic per counted doc), and a wavefront that finds k docs at or above a bucket's lower edge —
//     whoever scored them — takes that edge as its threshold. A wavefront's own k-th best is the k-th best of a 600th of
//     the query's docs; without the histogram some 50 k docs per query were evaluated on the way to a useful threshold;
// Nothing is approximated: every doc of the union is counted, and a doc is skipped only when an upper bound of its total
// is below the k-th best total seen (the argument of the TERM kernel's block-max skipping). Totals are fixed-point sums with
// k_or_wide's exponent (order-free, deterministic), handed to k_merge_items the same way.
// The host (rgpu_api.hip: search_or_lazy_group) sends a query here when it has >= 1 bitmap clause; a window that holds more
// touched docs in 2048 consecutive doc ids than it has cells flags the query, and the host runs it again through k_or_wide.
#pragma once
#include "doc_bitmap.hpp"
#include "search_or.hpp"

namespace rgpu {

constexpr int LZ_WAVES = 4;
constexpr int LZ_THREADS = 64 * LZ_WAVES;
constexpr int LZ_MAX_TERMS = 16;  // walked clauses (== RGPU_MAX_QUERY_TERMS)
constexpr int LZ_MAX_LAZY = 6;    // bitmap clauses per query (register budget: two words per clause and step are held)
#ifndef RGPU_LZ_PREFETCH
#define RGPU_LZ_PREFETCH 8
#endif
constexpr int LZ_PREFETCH = RGPU_LZ_PREFETCH;   // run heads requested at the start of a window (three VGPRs each: doc, next window's doc, score); 4 or 8
constexpr int LZ_QUEUE = 128;     // candidate queue entries (up to 63 waiting + 64 pushed at once)
constexpr int LZ_STEP_DOCS = 2048;  // one bitmap word per lane
#ifndef RGPU_LZ_HIST_SHIFT
#define RGPU_LZ_HIST_SHIFT 24
#endif
constexpr int LZ_HIST_SHIFT = RGPU_LZ_HIST_SHIFT;  // a bucket of the per-query histogram of finished totals = total >> this
constexpr int LZ_HIST = 1 << (31 - LZ_HIST_SHIFT);  // 128 buckets (a total is below 2^31)
constexpr int LZ_FLAG_BAIL = 2;   // a window did not fit (-> k_or_wide)

#ifndef RGPU_LZ_MIN_WAVES  // wavefronts per SIMD the register allocation aims at (LDS: 12 KB per wavefront at 16384-doc windows -> 3)
#define RGPU_LZ_MIN_WAVES 3
#endif
#ifndef RGPU_LZ_ABL  // developer ablations (variant builds only; results are wrong): 1 no lazy-only docs, 2 no candidate evaluation, 3 neither
#define RGPU_LZ_ABL 0
#endif
#ifdef RGPU_LZ_TIME  // developer instrumentation (variant builds only): wave-cycles per phase, summed over wavefronts
__device__ unsigned long long g_lz_dbg[16];  // [8] steps with the lazy-only test on [9] ... that had a doc in enough lists [10] per-doc bound iterations [11] steps [12] sum of `need` [13] sum of need_hi  // [0] run heads + pass 1 [1] numbering [2] bitmap words, hits, lazy-only docs [3] pass 2 [4] cell scan [5] candidate evaluation (inside 2 and 4) [6] set-up [7] windows
#define LZ_STAMP(t) const long long t = (long long)__builtin_readcyclecounter()
#define LZ_ADD(i, v) lz_t[i] += (v)
#else
#define LZ_STAMP(t) do {} while (0)
#define LZ_ADD(i, v) do {} while (0)
#endif

struct LazyClause {
  const uint2* words;    // {any, hi} per 32 docs (doc_bitmap.hpp)
  const uint32_t* ranks;
  const uint8_t* freqs;
  const uint32_t* ovf;   // {posting index, freq} pairs, n_ovf of them
  const uint32_t* nib;   // four bits per doc (0 absent, 1..14 the freq, 15 look it up), or null
  int32_t n_ovf;
  uint32_t ub;           // fixed-point upper bound of one posting's score
  uint32_t ub_lo;        // ... of a posting outside the bitmap's `hi` half (== ub when the sketch does not apply)
  float wk;              // weight * (k1 + 1)
  int32_t sim_table;
  int32_t pad[3];
};

struct LazyRun {  // one walked clause: its {doc, score} run (k_score_terms), closed by 64 sentinel entries
  int64_t base;
  int32_t len;
  int32_t pad;
};

struct LazyQuery {
  int32_t first_run, n_runs;    // walked clauses: LazyRun[first_run ..)
  int32_t first_lazy, n_lazy;   // bitmap clauses: LazyClause[first_lazy ..), by ub descending; n_lazy >= 1
  int32_t e;                    // fixed-point exponent (search_or_wide.hpp)
  uint32_t ub_sum;              // sum of the lazy clauses' ub (< 2^31)
  uint32_t ub_lo_sum;           // sum of their ub_lo
  int32_t pad;
};

__host__ __device__ constexpr size_t lz_wave_lds(int W, int C) {
  return (size_t)(W / 32) * 8 + (size_t)(W / 32) * 8 + (size_t)C * 4 + (size_t)C * 2 + (size_t)LZ_QUEUE * 8;
}
__host__ __device__ constexpr size_t lz_lds_bytes(int W, int C) { return (size_t)LZ_MAX_LAZY * 64 * 4 + (size_t)LZ_WAVES * lz_wave_lds(W, C); }

// items = (query, group of `windows_per_item` windows of W = STEPS * 2048 docs), one per wavefront; the four wavefronts of
// a workgroup work on the same query (they share its lazy clauses' norm caches); workgroup b works on query b % n_queries
// (k_or_windows: later workgroups start from the thresholds the earlier ones published; `first_item`: see below). counters[0] += candidates
// evaluated, [1] += of them docs held by lazy lists only.
template <bool WIDE, int STEPS>
__global__ __launch_bounds__(LZ_THREADS, RGPU_LZ_MIN_WAVES) void k_or_lazy(SegView seg, const LazyQuery* __restrict__ queries, const LazyRun* __restrict__ run_of,
                                                        const ScoredPosting* __restrict__ runs, const LazyClause* __restrict__ lazies,
                                                        int64_t sentinel_at, int n_queries, int windows_per_query, int windows_per_item,
                                                        int items_per_query, int first_item, int C, int k, uint64_t* __restrict__ partial_keys,
                                                        int32_t* __restrict__ partial_counts, unsigned long long* __restrict__ tau_slots,
                                                        int32_t* __restrict__ flags, unsigned long long* __restrict__ counters,
                                                        uint32_t* __restrict__ hist) {
  constexpr int W = STEPS * LZ_STEP_DOCS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = lane_id();
  const int wave = wave_id();
  LZ_STAMP(ts0);
  float* caches = reinterpret_cast<float*>(smem);  // caches[c][rank]: lazy clause c's norm cache by norm rank
  uint8_t* mine = smem + (size_t)LZ_MAX_LAZY * 256 + (size_t)wave * lz_wave_lds(W, C);
  uint2* tw = reinterpret_cast<uint2*>(mine);                  // per 32 docs of the window: {touched bits, touched docs before}
  // per 32 docs: {OR of the bitmap words of every lazy clause but the last, the last one's word}. The last lazy clause is the
  // one with the smallest bound — as a rule the query's longest list: most docs of the union are in that list only, and a
  // touched doc's bound then adds that one small bound instead of all of them
  uint2* uw = tw + W / 32;
  uint32_t* acc = reinterpret_cast<uint32_t*>(uw + W / 32);                                 // cell r: fixed-point total of the r-th touched doc (of the group)
  uint64_t* queue = reinterpret_cast<uint64_t*>(acc + C);       // candidates: (cell total << 32) | doc
  uint16_t* docoff = reinterpret_cast<uint16_t*>(queue + LZ_QUEUE);  // cell r: the doc's offset in the window

  const int q = (int)(blockIdx.x % (unsigned)n_queries);
  // first_item: this launch covers items [first_item, first_item + gridDim.x / n_queries * LZ_WAVES) of every query (a small
  // batch runs as a pilot launch over each query's first windows and a second one that starts from the pilot's thresholds)
  const int g = first_item + (int)(blockIdx.x / (unsigned)n_queries) * LZ_WAVES + wave;
  const int64_t item = (int64_t)q * items_per_query + g;
  const LazyQuery Q = queries[q];
  const int n = Q.n_runs, nl = Q.n_lazy;
  const float scale = ldexpf(1.0f, Q.e);
  auto to_fixed = [&](float score) -> uint32_t {
    const uint32_t v = (uint32_t)rintf(score * scale);
    return v > 1u ? v : 1u;
  };
  const int win0 = g * windows_per_item;
  const int win1 = min(windows_per_query, win0 + windows_per_item);
  const int32_t first_doc = win0 * W;

  // ---- lazy clauses: lane c holds clause c's constants
  uint64_t l_words = 0, l_ranks = 0, l_freqs = 0, l_nib = 0;
  uint32_t l_ub = 0u, l_ublo = 0u;
  uint32_t l_ubpre = 0u, l_dpre = 0u;  // ub, and ub - ub_lo, of lazy clauses 0 .. lane (sorted by ub descending)
  float l_wk = 0.f;
  if (lane < nl) {
    const LazyClause* L = lazies + Q.first_lazy + lane;
    l_words = (uint64_t)(uintptr_t)L->words; l_ranks = (uint64_t)(uintptr_t)L->ranks; l_freqs = (uint64_t)(uintptr_t)L->freqs;
    l_nib = (uint64_t)(uintptr_t)L->nib;
    l_ub = L->ub; l_ublo = L->ub_lo; l_wk = L->wk;
  }
  {
    uint32_t s = l_ub, d2 = l_ub - l_ublo;
    for (int d = 1; d < 8; d <<= 1) {  // inclusive scans over lanes 0 .. 7 (the host keeps the sums below 2^31)
      const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((max(lane - d, 0)) << 2, (int)s);
      const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute((max(lane - d, 0)) << 2, (int)d2);
      if (lane >= d) { s += o; d2 += o2; }
    }
    l_ubpre = s;
    l_dpre = d2;
  }
  const uint32_t ub_sum = Q.ub_sum, ub_lo_sum = Q.ub_lo_sum;
  const uint32_t ub_last = (uint32_t)readlane((int)l_ub, max(nl - 1, 0));
  const bool all_nib = !__ballot(lane < nl && l_nib == 0);  // every lazy clause has the four-bits-per-doc array (wave-uniform)
  for (int c = wave; c < nl; c += LZ_WAVES)
    caches[c * 64 + lane] = seg.sim_tables[(size_t)lazies[Q.first_lazy + c].sim_table * 257 + seg.rank_to_norm[lane]];
  for (int i = lane; i < W / 32; i += 64) { tw[i] = make_uint2(0u, 0u); uw[i] = make_uint2(0u, 0u); }
  for (int i = lane; i < C; i += 64) acc[i] = 0u;
  __syncthreads();

  // ---- walked clauses: lane t owns clause t's cursor — run base, length, the first entry with doc >= this item's first
  // window (one lane-parallel binary search over all clauses at once)
  const bool mine_run = lane < n;
  int64_t my_at = sentinel_at;  // (lanes without a clause: 64 sentinel entries — the head loads are unconditional)
  int my_len = 0;
  if (mine_run) { const LazyRun R = run_of[Q.first_run + lane]; my_at = R.base; my_len = R.len; }
  {
    const int64_t base = my_at;
    int lo = 0, hi = my_len;
    while (__ballot(lo < hi)) {
      if (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (runs[base + mid].doc < first_doc) lo = mid + 1; else hi = mid;
      }
    }
    my_at += lo;
  }
  int32_t my_next = mine_run ? runs[my_at].doc : 0x7fffffff;  // doc under the cursor (INT_MAX: the run is exhausted)

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int hits_lane = 0;
  int qn = 0;  // queued candidates (wave-uniform)
  bool bail = false;
  unsigned long long n_eval = 0, n_lazy_only = 0;
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);
  uint32_t* const my_hist = hist + (size_t)q * LZ_HIST;
  // k docs of this query (scored by any wavefront) have totals at or above the lower edge of the highest bucket b whose
  // suffix count reaches k: (b << LZ_HIST_SHIFT) bounds the k-th best from below. x[j]: the count of bucket LZ_HIST - 1 - 64 j - lane.
  constexpr int HIST_REGS = LZ_HIST / 64;
  auto hist_fold = [&](const uint32_t (&x)[HIST_REGS]) __attribute__((always_inline)) {
    int above = 0, b = -1;
#pragma unroll
    for (int j = 0; j < HIST_REGS; ++j) {
      const int sj = wave_incl_scan((int)x[j]) + above;
      const uint64_t m = __ballot(sj >= k);
      if (b < 0 && m) b = LZ_HIST - 1 - 64 * j - (int)__builtin_ctzll(m);
      above = readlane(sj, 63);
    }
    if (b > 0) {
      const uint64_t edge = (uint64_t)((uint32_t)b << LZ_HIST_SHIFT) << 32;
      if (edge > floor) floor = edge;
      if (floor > tau) tau = floor;
    }
  };
  auto hist_count = [&](bool counted, uint32_t total) __attribute__((always_inline)) {
    if (counted) atomicAdd(my_hist + (total >> LZ_HIST_SHIFT), 1u);  // (a total is below 2^31)
  };
  auto threshold = [&]() -> uint32_t { return max(1u, (uint32_t)(tau >> 32)); };
#ifdef RGPU_LZ_TIME
  long long lz_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

  // ---- evaluate up to 64 queued candidates: which lazy clauses hold the doc (the bound tightens), their freqs, the total
  typedef const __attribute__((address_space(1))) uint32_t* gwords;
  typedef const __attribute__((address_space(1))) uint8_t* gbytes1;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) u32x2* gpairs;
  auto drain = [&]() __attribute__((always_inline)) {
    LZ_STAMP(td0);
    const int take = min(qn, 64);
    const uint64_t e = lane < take ? queue[lane] : 0ull;
    const int rest = qn - take;
    const uint64_t mv = lane < rest ? queue[take + lane] : 0ull;
    wave_sync();
    if (lane < rest) queue[lane] = mv;
    wave_sync();
    qn = rest;
    n_eval += (unsigned long long)take;
    if (RGPU_LZ_ABL & 2) return;
    const bool on = lane < take;
    const uint32_t doc = (uint32_t)e;
    uint32_t total = (uint32_t)(e >> 32);
    const uint32_t dw = doc >> 5, below_mask = (1u << (doc & 31u)) - 1u, bit = 1u << (doc & 31u);
    const uint32_t nr = on ? (uint32_t)seg.norms[doc] : 0u;
    const uint32_t thr = threshold();
    bool alive = on;
    if (all_nib) {
      // one gather per clause: "absent" or the posting's freq — the exact total after ONE memory round trip. A 15 (freq 15 or
      // more) sends the whole group of candidates through the general path below.
      uint32_t exact = total;
      bool odd = false;
      for (int c0 = 0; c0 < nl; c0 += 4) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[j] = (((gwords)(uintptr_t)readlane64(l_nib, min(c0 + j, nl - 1)))[on ? doc >> 3 : 0u] >> (4u * (doc & 7u))) & 15u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c0 + j >= nl) continue;  // wave-uniform
          const int c = c0 + j;
          odd = odd || (on && v[j] == 15u);
          const float qf = (float)(int32_t)v[j];
          const float wk = __int_as_float(readlane(__float_as_int(l_wk), c));
          const uint32_t sc = to_fixed(wk * qf * __builtin_amdgcn_rcpf(qf + caches[c * 64 + nr]));
          if (v[j] != 0u) exact += sc;
        }
      }
      if (!__ballot(odd)) {
        const uint64_t key = (on && exact >= thr) ? ((uint64_t)exact << 32) | (uint32_t)~doc : 0ull;
        hist_count(key != 0ull, exact);
        if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
        LZ_STAMP(td1a);
        LZ_ADD(5, td1a - td0);
        return;
      }
    }
    for (int c0 = 0; c0 < nl; c0 += 4) {
      uint32_t wd[4], wh[4], rk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // clamped, not guarded: eight loads in flight
        const int c = min(c0 + j, nl - 1);
        const u32x2 pr = ((gpairs)(uintptr_t)readlane64(l_words, c))[dw];
        wd[j] = pr.x; wh[j] = pr.y;
        rk[j] = ((gwords)(uintptr_t)readlane64(l_ranks, c))[dw];
      }
      // the clauses of this group that hold the doc — a posting outside the `hi` half is bounded by ub_lo; a doc that cannot
      // reach the threshold even so drops out
      uint32_t bound = total;
      if (c0 + 4 < nl) bound += (uint32_t)readlane((int)l_ubpre, nl - 1) - (uint32_t)readlane((int)l_ubpre, c0 + 3);  // later groups: all assumed present
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < nl && (wd[j] & bit)) bound += (wh[j] & bit) ? (uint32_t)readlane((int)l_ub, c0 + j) : (uint32_t)readlane((int)l_ublo, c0 + j);
      alive = alive && bound >= thr;
      uint32_t fq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = min(c0 + j, nl - 1);
        const bool here = alive && c0 + j < nl && (wd[j] & bit);
        const uint32_t pidx = here ? rk[j] + (uint32_t)__popc(wd[j] & below_mask) : 0u;
        rk[j] = pidx;
        fq[j] = ((gbytes1)(uintptr_t)readlane64(l_freqs, c))[pidx];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j >= nl) continue;  // wave-uniform
        const int c = c0 + j;
        const bool here = alive && (wd[j] & bit);
        uint32_t f = fq[j];
        if (__ballot(here && f == 255u)) {  // a clamped freq byte: the posting's freq is in the clause's overflow list
          const uint32_t* ovf = lazies[Q.first_lazy + c].ovf;
          const int novf = lazies[Q.first_lazy + c].n_ovf;
          if (here && f == 255u)
            for (int i = 0; i < novf; ++i)
              if (ovf[2 * i] == rk[j]) { f = ovf[2 * i + 1]; break; }
        }
        const float qf = (float)(int32_t)f;
        const float wk = __int_as_float(readlane(__float_as_int(l_wk), c));
        const uint32_t sc = to_fixed(wk * qf * __builtin_amdgcn_rcpf(qf + caches[c * 64 + nr]));
        if (here) total += sc;
      }
    }
    const uint64_t key = (alive && total >= thr) ? ((uint64_t)total << 32) | (uint32_t)~doc : 0ull;
    hist_count(key != 0ull, total);
    if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
    LZ_STAMP(td1);
    LZ_ADD(5, td1 - td0);
  };
  auto push = [&](bool want, uint32_t cell_total, uint32_t doc) __attribute__((always_inline)) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    if (want) queue[qn + mbcnt(m)] = ((uint64_t)cell_total << 32) | doc;
    qn += __popcll(m);
    wave_sync();
    if (qn >= 64) drain();
  };
  // Run heads: the first 64 entries from every clause's cursor — doc ids requested a window ahead, scores after pass 1 (unconditionally: a run ends in 64
  // sentinel entries, so cursor + lane is always inside the buffer). Two register tuples indexed by the clause loop's scalar
  // variable — ext_vector_type: an array (of structs or of scalars) and a switch over named variables all ended up in scratch.
  typedef int32_t heads_i __attribute__((ext_vector_type(LZ_PREFETCH)));
  typedef float heads_f __attribute__((ext_vector_type(LZ_PREFETCH)));
  heads_i pd, pd_n;
  heads_f ps;
#pragma unroll
  for (int t = 0; t < LZ_PREFETCH; ++t) {
    const ScoredPosting h = runs[(int64_t)readlane64((uint64_t)my_at, t) + lane];
    pd[t] = h.doc; ps[t] = h.score;
  }
  pd_n = pd;
  LZ_STAMP(ts1);
  LZ_ADD(6, ts1 - ts0);

  u32x2 wq[4][4];  // the lazy clauses' {any, hi} words: [slot = step & 3][clause 0 .. 3]
  for (int win = win0; win < win1; ++win) {
    LZ_STAMP(t0);
    const int32_t w0 = win * W;
    const int32_t w1 = min(seg.max_doc, w0 + W);
    const uint64_t seen = shared.peek();
    uint32_t hx[HIST_REGS];
#pragma unroll
    for (int j = 0; j < HIST_REGS; ++j) hx[j] = __hip_atomic_load(my_hist + (LZ_HIST - 1 - 64 * j - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- loads first: the lazy clauses' {any, hi} bitmap words of the window's first four 2048-doc steps (clauses 0 .. 3; a
    // query with more takes the others step by step) — they arrive while pass 1 runs; a step's slot is refilled with the words
    // of the step four further on as soon as the step is done
    const uint64_t active0 = __ballot(my_next < w1);
    const uint32_t word0 = (uint32_t)(w0 >> 5) + (uint32_t)lane;
    auto words_issue = [&](int slot, int step) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c)  // (clamped: OR and the counts below ignore duplicates)
        wq[slot][c] = ((gpairs)(uintptr_t)readlane64(l_words, min(c, nl - 1)))[word0 + 64u * (uint32_t)step];
    };
    // (a window of four steps: all of its words fit the slots, and they were requested right after the previous window's steps)
    constexpr bool WORDS_AHEAD = STEPS <= 4;
    if (!WORDS_AHEAD || win == win0) {
#pragma unroll
      for (int i = 0; i < 4 && i < STEPS; ++i) words_issue(i, i);
    }

    // One clause's entries inside the window. PASS 1 sets their touched bits and leaves the clause's advance (lane t); PASS 2
    // adds the entries whose docs lie in [d_lo, d_lo + d_len) to their cells. The clause loops are unrolled over the prefetched
    // heads (static registers, no load and no wait in the common path); a stretch of more than 64 entries, and clauses beyond
    // the prefetched ones, go through walk_more.
    int my_taken = 0;
    int32_t my_next_new = my_next;
    auto visit = [&](auto pass_tag, int32_t e_doc, float e_score, int32_t d_lo, uint32_t d_len, uint32_t r0) __attribute__((always_inline)) -> int {
      constexpr bool SECOND = decltype(pass_tag)::value;
      const bool in = e_doc < w1;
      const uint32_t o = (uint32_t)(e_doc - w0);
      if (!SECOND) {
        if (in) __hip_atomic_fetch_or(&tw[o >> 5].x, 1u << (o & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (in && (uint32_t)(e_doc - d_lo) < d_len) {
        const uint2 tt = tw[o >> 5];
        const uint32_t cell = tt.y + (uint32_t)__popc(tt.x & ((1u << (o & 31u)) - 1u)) - r0;
        __hip_atomic_fetch_add(acc + cell, to_fixed(e_score), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        docoff[cell] = (uint16_t)o;
      }
      return __popcll(__ballot(in));  // runs are doc-sorted: the in-window entries are a prefix
    };
    auto walk_more = [&](auto pass_tag, int t, int taken, int32_t d_lo, uint32_t d_len, uint32_t r0) __attribute__((always_inline)) {
      constexpr bool SECOND = decltype(pass_tag)::value;
      while (true) {
        const ScoredPosting h = runs[(int64_t)readlane64((uint64_t)my_at, t) + taken + lane];
        const int cnt = visit(pass_tag, h.doc, h.score, d_lo, d_len, r0);
        taken += cnt;
        if (cnt < 64) {
          if (!SECOND) { const int32_t nx = readlane(h.doc, cnt); if (lane == t) { my_taken = taken; my_next_new = nx; } }
          break;
        }
      }
    };
    // (tried: the eight heads straight-line and predicated — spare cells for the postings outside the window, no exec-mask
    // branch, eight LDS operations in flight: 3.91 ms against 3.77 for the branches below, which skip the clauses without a
    // posting in the window)
    auto walk_all = [&](auto pass_tag, int32_t d_lo, uint32_t d_len, uint32_t r0) __attribute__((always_inline)) {
      constexpr bool SECOND = decltype(pass_tag)::value;
      uint64_t more = active0 >> LZ_PREFETCH << LZ_PREFETCH;  // clauses that go on past their 64 prefetched entries (or have none)
#pragma unroll
      for (int t = 0; t < LZ_PREFETCH; ++t) {
        if (!((active0 >> t) & 1ull)) continue;  // wave-uniform
        const int cnt = visit(pass_tag, pd[t], ps[t], d_lo, d_len, r0);
        if (cnt == 64) more |= 1ull << t;
        else if (!SECOND) { const int32_t nx = readlane(pd[t], cnt); if (lane == t) { my_taken = cnt; my_next_new = nx; } }
      }
      for (; more; more &= more - 1) {
        const int t = (int)__builtin_ctzll(more);
        walk_more(pass_tag, t, t < LZ_PREFETCH ? 64 : 0, d_lo, d_len, r0);
      }
    };
    walk_all(std::false_type{}, 0, 0u, 0u);
    wave_sync();
    const int64_t my_at_n = my_at + my_taken;  // the cursors of the next window
    // (the heads' scores are only needed in pass 2: requested now, so that the doc ids alone are held a window ahead)
#pragma unroll
    for (int t = 0; t < LZ_PREFETCH; ++t) ps[t] = runs[(int64_t)readlane64((uint64_t)my_at, t) + lane].score;
    LZ_STAMP(t1);
    // ---- number the touched docs
    uint32_t st_cnt = 0, st_base = 0;  // lane i: touched docs of step i / before it
    {
      int base = 0;
#pragma unroll
      for (int i = 0; i < STEPS; ++i) {
        const int w = 64 * i + lane;
        const uint32_t t = tw[w].x;
        const int cnt = __popc(t);
        const int incl = wave_incl_scan(cnt);
        tw[w].y = (uint32_t)(base + incl - cnt);
        const int tot = readlane(incl, 63);
        if (lane == i) { st_cnt = (uint32_t)tot; st_base = (uint32_t)base; }
        base += tot;
      }
      wave_sync();
    }
    shared.fold(seen, tau, floor);
    hist_fold(hx);
    LZ_STAMP(t2);
    // ---- the lazy clauses' bitmap words, 2048 docs at a time: total_hits, the unions for the cell scan, and the docs that lazy
    // lists alone could lift into the top-k
    uint32_t lazy_steps = 0u;  // steps that left candidate words behind (wave-uniform)
    // (NC = 4 or 8: the clause slots a query uses — most have at most four bitmap clauses, and their code is half as long)
    auto step = [&](auto nc_tag, auto i_tag) __attribute__((always_inline)) {
      constexpr int NC = decltype(nc_tag)::value;
      constexpr int i = decltype(i_tag)::value;
      uint32_t cur[NC], chi[NC];
#pragma unroll
      for (int c = 0; c < 4; ++c) { cur[c] = wq[i & 3][c].x; chi[c] = wq[i & 3][c].y; }
      if (i + 4 < STEPS) words_issue(i & 3, i + 4);
      if (NC > 4) {
#pragma unroll
        for (int c = 4; c < NC; ++c) {
          const u32x2 pr = ((gpairs)(uintptr_t)readlane64(l_words, min(c, nl - 1)))[word0 + 64u * (uint32_t)i];
          cur[c] = pr.x; chi[c] = pr.y;
        }
      }
      const int w = 64 * i + lane;
      uint32_t un_a = 0u;
#pragma unroll
      for (int c = 0; c < NC - 1; ++c) un_a |= c < nl - 1 ? cur[c] : 0u;
      const uint32_t un_b = cur[NC - 1];  // (the loads are clamped: the slots from nl - 1 on hold the last clause's word)
      const uint32_t un = un_a | un_b;
      const uint32_t t = tw[w].x;
      uw[w] = make_uint2(un_a, un_b);
      hits_lane += __popc(t | un);
      LZ_ADD(11, 1);
      if ((RGPU_LZ_ABL & 1) || ub_sum < threshold()) return;
      LZ_ADD(8, 1);
      const uint32_t thr = threshold();
      // lists a doc must be in at best: the smallest m whose m largest bounds reach the threshold (bounds are sorted: l_ubpre
      // ascends over lanes 0 .. nl-1); `hi` postings it must have at best: with every clause present at ub_lo, the h largest
      // (ub - ub_lo) — the host keeps those in the same order — must close the gap (more than there are clauses: nobody)
      const int need = __popcll(__ballot(lane < nl && l_ubpre < thr)) + 1;
      const int need_hi = ub_lo_sum >= thr ? 0 : __popcll(__ballot(lane < nl && ub_lo_sum + l_dpre < thr)) + 1;
      LZ_ADD(12, need); LZ_ADD(13, need_hi);
      if (need > nl || need_hi > nl) return;
      // docs held by at least `m` of the clauses (x = their words; clamped duplicates of the last clause are harmless to an OR
      // and an AND): an OR, an AND, or a bit-sliced count
      auto at_least = [&](const uint32_t (&x)[NC], int m) __attribute__((always_inline)) -> uint32_t {
        if (m <= 0) return 0xffffffffu;
        if (m == 1 || m == nl) {
          uint32_t v = x[0];
#pragma unroll
          for (int c = 1; c < NC; ++c) v = m == 1 ? (v | x[c]) : (v & x[c]);
          return v;
        }
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const uint32_t v = c < nl ? x[c] : 0u;
          const uint32_t k0 = p0 & v; p0 ^= v;
          const uint32_t k1 = p1 & k0; p1 ^= k0;
          if (NC > 7) { const uint32_t k2 = p2 & k1; p2 ^= k1; p3 ^= k2; }
          else p2 ^= k1;
        }
        uint32_t gt = 0u, eq = 0xffffffffu;  // plane by plane from the top
        auto plane = [&](uint32_t pv, int p) __attribute__((always_inline)) {
          const uint32_t nb = ((m >> p) & 1) ? 0xffffffffu : 0u;
          gt |= eq & pv & ~nb;
          eq &= ~(pv ^ nb);
        };
        if (NC > 7) plane(p3, 3);
        plane(p2, 2); plane(p1, 1); plane(p0, 0);
        return gt | eq;
      };
      uint32_t cand = at_least(cur, need) & ~t;
      if (!__ballot(cand != 0u)) return;  // (six steps in ten end here when a doc has to be in every list)
      LZ_ADD(9, 1);
      cand &= at_least(chi, need_hi);
      // the docs that get this far are checked one by one — do the bounds of the clauses that hold the doc (ub_lo outside the
      // `hi` halves) reach the threshold? — and the survivors' bits wait in the accumulator cells (idle until pass 2): they are
      // queued after the steps, by one copy of that code instead of one per step
      uint32_t keep = 0u;
      while (__ballot(cand != 0u)) {
        LZ_ADD(10, 1);
        const bool on = cand != 0u;
        const uint32_t b = on ? (uint32_t)__builtin_ctz(cand) : 0u;
        cand &= cand - 1u;
        uint32_t bound = 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c < nl && ((cur[c] >> b) & 1u)) bound += ((chi[c] >> b) & 1u) ? (uint32_t)readlane((int)l_ub, c) : (uint32_t)readlane((int)l_ublo, c);
        if (on && bound >= thr) keep |= 1u << b;
      }
      if (__ballot(keep != 0u)) { acc[w] = keep; lazy_steps |= 1u << i; }
    };
    auto steps_all = [&](auto nc_tag) __attribute__((always_inline)) {
      step(nc_tag, std::integral_constant<int, 0>{}); step(nc_tag, std::integral_constant<int, 1>{});
      step(nc_tag, std::integral_constant<int, 2>{}); step(nc_tag, std::integral_constant<int, 3>{});
      if (STEPS > 4) {
        step(nc_tag, std::integral_constant<int, 4 % STEPS>{}); step(nc_tag, std::integral_constant<int, 5 % STEPS>{});
        step(nc_tag, std::integral_constant<int, 6 % STEPS>{}); step(nc_tag, std::integral_constant<int, 7 % STEPS>{});
      }
    };
    static_assert(STEPS == 4 || STEPS == 8, "steps_all spells the steps out");
    if (nl <= 4) steps_all(std::integral_constant<int, 4>{});
    else steps_all(std::integral_constant<int, LZ_MAX_LAZY>{});
    wave_sync();
    if (WORDS_AHEAD) {  // the next window's words (past the item's last window: the bitmaps' zero padding, or another window's words)
      const uint32_t keep_word0 = word0;
      (void)keep_word0;
#pragma unroll
      for (int i = 0; i < STEPS && i < 4; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          wq[i][c] = ((gpairs)(uintptr_t)readlane64(l_words, min(c, nl - 1)))[word0 + (uint32_t)(W / 32) + 64u * (uint32_t)i];
      }
    }
    for (; lazy_steps; lazy_steps &= lazy_steps - 1u) {
      const int i = (int)__builtin_ctz(lazy_steps);
      const int w = 64 * i + lane;
      uint32_t cand = acc[w];
      acc[w] = 0u;
      const uint32_t doc0 = (uint32_t)w0 + 32u * (uint32_t)w;
      while (__ballot(cand != 0u)) {
        const bool on = cand != 0u;
        const uint32_t b = on ? (uint32_t)__builtin_ctz(cand) : 0u;
        cand &= cand - 1u;
        n_lazy_only += (unsigned long long)__popcll(__ballot(on));
        push(on, 0u, doc0 + b);
      }
    }
    wave_sync();
    LZ_STAMP(t3);
    LZ_ADD(0, t1 - t0); LZ_ADD(1, t2 - t1); LZ_ADD(2, t3 - t2);

    // ---- the next window's run heads are requested now: they arrive behind pass 2 and the cell scan
#pragma unroll
    for (int t = 0; t < LZ_PREFETCH; ++t) {
      pd_n[t] = runs[(int64_t)readlane64((uint64_t)my_at_n, t) + lane].doc;
    }
    // ---- groups of consecutive 2048-doc steps whose touched docs fit the cells: pass 2 over the group's docs, then its scan
    int i0 = 0;
    while (i0 < STEPS) {
      int i1 = i0, sum = 0;
      while (i1 < STEPS && sum + readlane((int)st_cnt, i1) <= C) sum += readlane((int)st_cnt, i1++);
      if (i1 == i0) {  // 2048 consecutive doc ids hold more touched docs than there are cells: not this kernel's window
        bail = true;
        i1 = i0 + 1;
        sum = 0;
      }
      if (sum > 0) {
        LZ_STAMP(t4);
        const uint32_t r0 = (uint32_t)readlane((int)st_base, i0);
        const int32_t d_lo = w0 + LZ_STEP_DOCS * i0;
        const uint32_t d_len = (uint32_t)(LZ_STEP_DOCS * (i1 - i0));
        walk_all(std::true_type{}, d_lo, d_len, r0);
        wave_sync();
        LZ_STAMP(t5);
        // the cells: one outside every lazy list is a finished total; one inside is a candidate if the lazy clauses' bounds
        // can lift it to the threshold
        for (int c0 = 0; c0 < sum; c0 += 64) {
          const bool on = c0 + lane < sum;
          uint32_t a = 0u, o = 0u;
          if (on) { a = acc[c0 + lane]; o = docoff[c0 + lane]; acc[c0 + lane] = 0u; }
          const uint2 un = uw[o >> 5];
          const bool in_a = (un.x >> (o & 31u)) & 1u, in_b = (un.y >> (o & 31u)) & 1u;
          const bool in_u = in_a || in_b;
          const uint32_t thr = threshold();
          const uint32_t doc = (uint32_t)w0 + o;
          const uint64_t key = (on && !in_u && a >= thr) ? ((uint64_t)a << 32) | (uint32_t)~doc : 0ull;
          hist_count(key != 0ull, a);
          if (__ballot(key > tau)) topk_offer<WIDE>(top, key, tau, k, lane, floor);
          push(on && in_u && a + (in_a ? ub_sum - ub_last : 0u) + (in_b ? ub_last : 0u) >= thr, a, doc);  // (all below 2^31)
        }
        LZ_STAMP(t6);
        LZ_ADD(3, t5 - t4); LZ_ADD(4, t6 - t5);
      }
      i0 = i1;
    }
    for (int i = lane; i < W / 32; i += 64) tw[i] = make_uint2(0u, 0u);
    my_at = my_at_n;
    my_next = my_next_new;
    pd = pd_n;
    wave_sync();
    shared.publish<WIDE>(top, k, lane);
    LZ_STAMP(t7);
    LZ_ADD(7, t7 - t0);
  }
  while (qn > 0) drain();
  shared.publish<WIDE>(top, k, lane);
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  const int count = wave_reduce_add(hits_lane);
#ifdef RGPU_LZ_TIME
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_lz_dbg[i], (unsigned long long)lz_t[i]);
#endif
  if (lane == 0) {
    partial_counts[item] = count;
    if (bail) atomicOr(flags + q, LZ_FLAG_BAIL);
    if (counters != nullptr) { atomicAdd(counters, n_eval); atomicAdd(counters + 1, n_lazy_only); }
#ifdef RGPU_LZ_TIME
    atomicAdd(counters + 2 + 2 * q, n_eval);
    atomicAdd(counters + 3 + 2 * q, n_lazy_only);
#endif
  }
}

}  // namespace rgpu
