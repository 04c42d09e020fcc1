// C ABI of the MI355X query-evaluation path (include/rucene_gpu.h): host-side orchestration around the
// gfx950 kernels in kernels/*.hpp. HIP runtime only — no torch types, no CPU fallback: every entry point
// either runs on the GPU or fails with a negative rgpu_status.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <tuple>
#include <vector>

#include "../../include/rucene_gpu.h"
#include "host/doc_format.hpp"
#include "host/norms_format.hpp"
#include "host/term_dict.hpp"
#include "host/field_infos_format.hpp"
#include "host/segment_infos_format.hpp"
#include "host/compound_format.hpp"
#include "kernels/prepare.hpp"
#include "kernels/search.hpp"
#include "kernels/search_and.hpp"
#include "kernels/decode_positions.hpp"
#include "kernels/search_or.hpp"
#include "kernels/search_or_wide.hpp"
#include "kernels/search_or_lazy.hpp"
#include "host/flat_fp_map.hpp"
#include "host/host_threads.hpp"
#include "host/prepared_map.hpp"
#include "kernels/search_phrase.hpp"
#include "kernels/search_term.hpp"

using namespace rgpu;

static_assert(sizeof(HitOut) == sizeof(rgpu_hit), "hit layout");
constexpr int32_t RGPU_PASS_K = 128;  // hits per pass: the widest list a wavefront's registers hold (kernels/wave.hpp WaveTopK)
static_assert(OR_MAX_TERMS >= RGPU_MAX_QUERY_TERMS, "k_or_windows keeps one cursor per clause in a lane / register slot");

static thread_local std::string g_last_error;

static int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
// A launch is described to the hardware in WORK-ITEMS, 32 bits per dimension: grid.x * block.x beyond 2^32 - 1 is cut down to
// its low 32 bits by the runtime without a word (hipGetLastError stays hipSuccess) — the kernel then runs over the first part
// of its work and nothing says so. (Round 4 found the phrase kernels, one wavefront per candidate slot, doing that from 67 M
// slots on.) Every launch goes through RGPU_LAUNCH: a grid that does not fit is not launched and the call fails
// (launch_status(), which the API functions check where they checked hipGetLastError()).
static thread_local const char* g_refused_launch = nullptr;
static inline hipError_t launch_status() {
  if (g_refused_launch != nullptr) {
    g_last_error = std::string("launch refused: the grid of ") + g_refused_launch + " exceeds 2^32 work-items";
    g_refused_launch = nullptr;
    (void)hipGetLastError();
    return hipErrorInvalidConfiguration;
  }
  return hipGetLastError();
}
// a workgroup count: saturates instead of wrapping (RGPU_LAUNCH then refuses the launch)
static inline unsigned wg_count(unsigned long long n) { return n > 0xffffffffull ? 0xffffffffu : (unsigned)n; }
#define RGPU_LAUNCH(kern, grid, block, shmem, stream, ...)                                                  \
  do {                                                                                                     \
    const dim3 g_ = (grid), b_ = (block);                                                                  \
    if ((unsigned long long)g_.x * (unsigned long long)b_.x > 0xffffffffull) g_refused_launch = #kern;     \
    else hipLaunchKernelGGL(kern, g_, b_, shmem, stream, __VA_ARGS__);                                     \
  } while (0)
#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return fail(RGPU_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
// scratch_take() may have to settle a deferred disjunction batch first (rgpu_config.or_deferred): when THAT fails, the call that
// happened to need the slot reports the deferred batch's own status and message (it used to become a generic
// "hipErrorUnknown" attributed to the unrelated later call)
static thread_local int32_t g_settle_rc = 0;
static thread_local std::string g_settle_why;
#define SCRATCH_TAKE(c)                                                                                                   \
  do {                                                                                                                    \
    hipError_t _e = scratch_take(c);                                                                                      \
    if (_e != hipSuccess) {                                                                                               \
      if (g_settle_rc != 0) {                                                                                             \
        const int32_t _rc = g_settle_rc;                                                                                  \
        g_settle_rc = 0;                                                                                                  \
        return fail(_rc, "a deferred disjunction batch failed when its flags were looked at: " + g_settle_why);          \
      }                                                                                                                   \
      return fail(RGPU_ERR_RUNTIME, std::string("scratch_take: ") + hipGetErrorString(_e));                               \
    }                                                                                                                     \
  } while (0)

namespace {

// growable device array (copy-on-grow); sizes are element counts
template <typename T>
struct DevVec {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n, size_t keep, hipStream_t s) {
    if (n <= cap) return hipSuccess;
    size_t ncap = std::max(n, cap * 2);
    ncap = std::max<size_t>(ncap, 1024);
    T* np = nullptr;
    hipError_t e = hipMalloc(&np, ncap * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) { (void)hipFree(np); return e; }
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = ncap;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostPinned {
  uint8_t* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    const size_t ncap = std::max(n, cap * 2);
    uint8_t* np = nullptr;
    hipError_t e = hipHostMalloc((void**)&np, ncap, hipHostMallocDefault);
    if (e != hipSuccess) return e;  // the old buffer stays valid
    if (p) (void)hipHostFree(p);
    p = np;
    cap = ncap;
    return hipSuccess;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct StatSlot {
  std::string name;
  int64_t launches = 0;
  double total_ms = 0;
  int64_t postings = 0;
  // every event-timed launch's own duration (at most STAT_SAMPLES_MAX: a bench pass is tens of launches): the median is what a
  // roofline is priced on — a mean over 20 launches is multiplied by one stalled launch (round 5's driver run read 1.775 ms for a
  // 0.23 ms kernel that way)
  std::vector<float> samples;
};
constexpr size_t STAT_SAMPLES_MAX = 8192;
struct PendingEvent { int slot; hipEvent_t start, stop; };

}  // namespace

constexpr int N_SCRATCH = 4;
struct Scratch {
  HostPinned h_stage;
  DevVec<uint8_t> d_stage;
  DevVec<uint64_t> d_partial_keys;
  DevVec<int32_t> d_partial_counts;
  DevVec<unsigned long long> d_tau;  // per-query shared top-k thresholds
  DevVec<unsigned long long> d_touched;  // AND: per-query encoded bytes of the blocks the kernel decoded
  DevVec<ScoredPosting> d_runs;  // OR: per-clause {doc, score} runs; ReqOptScorer records (a slot's own: no sync at the end of the group)
  HostPinned h_back;             // flags a launch set hands back to the host (fixed-point floor, windows that did not fit)
  bool settles_later = false;    // ... which the host has not looked at yet (rgpu_ctx::pending_or)
  hipEvent_t done = nullptr;
  hipEvent_t staged = nullptr;   // the slot's plan has reached d_stage (recorded on the context's upload stream)
  bool busy = false;
  void release() {
    h_stage.release(); d_stage.release(); d_partial_keys.release(); d_partial_counts.release();
    d_tau.release(); d_touched.release(); d_runs.release(); h_back.release();
    if (done) (void)hipEventDestroy(done);
    if (staged) (void)hipEventDestroy(staged);
    done = nullptr;
    staged = nullptr;
  }
};

struct rgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // (opt-in, see upload_aside) The staged plan of a batch travels on a stream of its own and the caller's stream only waits for the
  // event behind it: on ONE caller stream the copy of batch i + 1 then runs under the kernels of batch i instead of between them
  hipStream_t upload = nullptr;
  rgpu_config cfg{};
  bool blocks_per_item_auto = false;
  bool and_blocks_per_item_auto = false;
  std::mutex mu;
  DevVec<float> sim_tables;
  int n_sim_tables = 0;
  std::vector<uint8_t> sim_monotone;  // per table: cache[] finite, >= 0 and non-increasing in the norm byte
  int64_t or_wide_redone = 0;         // queries k_or_wide handed back to the f32 kernel (fixed-point floor); rgpu_kernel_stats reports it as "or_wide_redo_queries"
  std::vector<float> sim_k1;          // per table: k1
  std::vector<uint8_t> sim_nonneg;    // per table: k1 and every cache[] entry finite and >= 0 (a score is then within [0, weight * (k1 + 1)])
  std::vector<float> sim_cache_min;   // per table: the smallest cache[] entry (freq / (freq + cache) is largest there: k_or_lazy's score bounds)
  int64_t or_lazy_evals = 0, or_lazy_only = 0;
  size_t bitmap_bytes = 0;    // doc bitmaps of every segment of this context (the budget is the device's, not a segment's)
  size_t bitmap_budget = 0;   // bytes; rgpu_config.bitmap_budget_mib (0: an eighth of the device's memory)
  size_t prepared_budget = 0; // bytes per segment; rgpu_config.prepared_budget_mib (0: no ceiling)
  // search_or_lazy_group: doc_start_fp -> the batch's run of that term. Direct-mapped and stamped with the call's number, so a
  // call neither allocates nor clears it (a collision costs a second run of the same term, nothing else)
  struct UniqSlot { int64_t fp; uint32_t stamp; int32_t idx; };
  std::vector<UniqSlot> lazy_uniq;
  uint32_t lazy_stamp = 0;  // k_or_lazy: candidates evaluated / of them docs held by lazy lists only (last launch)
  // Per-call scratch, in rotating slots: a search call only enqueues work (staging copy + kernels) on its stream
  // and marks its slot with an event; the slot is waited for when its turn comes again, so the host prepares batch
  // i+1 while the GPU runs batch i and a caller synchronizes the stream once, when it wants the results.
  Scratch scr[N_SCRATCH];
  Scratch* S = &scr[0];
  int scr_next = 0;
  DevVec<ScoredPosting> d_runs;  // scratch of calls that end with a stream sync (bitmap builds; OR groups whose runs exceed RUNS_PER_SLOT_MAX)
  // The fixed-point disjunction kernels hand flags back (a top-k that reaches below the fixed-point floor, a window that did not
  // fit): the host then runs those queries again through another kernel. Looking at the flags needs the launch set to have
  // finished — a stream sync at the end of every OR group (rounds 2-4), with the host's 0.5 ms of partitioning for the next batch
  // waiting behind it. Instead the look can wait (`defer_or`): the group records an event, the flags travel to pinned memory,
  // and what is left to do is a closure that runs when the NEXT call needs the slot or the results are about to be consumed
  // (rgpu_synchronize, the collective of a sharded batch, rgpu_search_batch's copy to the host).
  std::vector<std::function<int32_t()>> pending_or;
  bool defer_or = false;  // set by the entry point for the duration of one call
  // staged plans travel on `upload` when RGPU_UPLOAD_ASIDE=1 is in the environment (default: on the caller's stream). Measured in round 6
  // (1024-query TERM batches): ONE caller stream gains — 0.049 ms per step against 0.059 in a 200-step loop, 0.062 against 0.071 in 20-step
  // regions — but the cross-stream event wait costs 15-30 us of latency whenever the pipeline is shallow: two alternating streams
  // 0.063 against 0.047 in 20-step regions, the 3-term AND batch 0.290 against 0.258. Off by default.
  bool upload_aside = false;
  bool host_time = false;         // RGPU_HOST_TIME=1: term_batch_fast prints where the calling thread's time goes
  bool stage_by_kernel = true;    // staged plans reach the device through k_stage_copy (RGPU_STAGE_COPY=dma: hipMemcpyAsync)
  bool term_fold = true;          // plan + search in one call, single-term batches: k_search_term folds the item lists itself (RGPU_TERM_FOLD=0: k_merge_items)
  int term_min_item_blocks = 64;  // ... and none of its items shorter than this (RGPU_TERM_MIN_ITEM_BLOCKS)
  int term_target_items = 3000;  // single-term launches: items of the launch's size, at most about this many (RGPU_TERM_TARGET_ITEMS in the environment)
  int term_split = 8;         // single-term queries: at least this many items per query, of 64 blocks or more each (RGPU_TERM_SPLIT in the environment; 1: off)
  bool memb_only_on = true;   // membership-only bits for sparse first clauses of conjunctions (RGPU_AND_MEMB_ONLY=0 in the environment: off; A/B, tests)
  bool term_sketches = true;  // block-max sketches for single-term queries (search_term.hpp); RGPU_TERM_SKETCH=0 in the environment turns them off (A/B, tests)
  DevVec<uint32_t> pos_counts;             // rgpu_decode_positions: positions per directory slot -> their exclusive prefix sums
  DevVec<unsigned long long> pos_tiles;    // ... the scan's tile sums (+ [0]: unused, [1]: the call's total)
  DevVec<int32_t> phrase_docs;             // phrase search: the conjunctions' matches (candidates), per query
  DevVec<uint64_t> phrase_keys;            // ... and their keys (0 = phrase freq 0)
  DevVec<int64_t> phrase_redo;             // ... and the slots the 64-candidate kernel left for the one-candidate kernel (PHRASE_REDO_LIST_CAP)
  DevVec<unsigned long long> phrase_count;  // ... how many each query's conjunction produced
  DevVec<HitOut> host_api_hits;  // rgpu_search_batch (blocking, host outputs): device-side result rows
  DevVec<int64_t> host_api_totals;
  HostPinned host_api_rows;  // ... of a SMALL batch: pinned host memory the merge kernels write straight into (no copy operations behind them)
  int* d_err = nullptr;
  // k > 128: the search runs in passes of up to 128 hits; a pass writes columns [col0, col0 + k_pass) of rows `stride` hits
  // long and keeps below the previous pass's worst key per caller row (d_ceil; null in the first pass)
  struct Pass { int stride = 0, col0 = 0; const unsigned long long* ceil_in = nullptr; unsigned long long* ceil_out = nullptr; } pass;
  // (one set of ceiling arrays per call in flight: rotating like the scratch slots, so a deep-page call stays enqueue-only)
  struct CeilSlot { DevVec<unsigned long long> d; hipEvent_t done = nullptr; bool busy = false; } ceil_slots[N_SCRATCH];
  int ceil_next = 0;
  Scratch* last_and = nullptr;  // the slot whose d_touched the most recent AND launch filled
  int last_and_queries = 0;
  // rgpu_last_search_counters: the most recent TERM / AND / wide-OR launch
  Scratch* last_counted = nullptr;
  const unsigned long long* last_counted_words = nullptr;  // [touched bytes per query][blocks per query] of that launch (the slot's d_touched, or inside its stage)
  int last_counted_op = -1, last_counted_queries = 0;
  int64_t last_counted_dir_blocks = 0;   // TERM: blocks of the launch's lists (what it READ of their directory the kernel counts itself, since round 6)
  int64_t last_counted_loose = 0;        // postings outside FullBlocks (prepared tails, singletons) of the (lead) clauses
  int64_t last_counted_postings = 0;     // sum of doc_freq over the launch's clauses
  int64_t last_counted_algo_bytes = 0;   // wide OR: encoded bytes of every clause + 1 B norm per posting
  int64_t last_counted_or_decoded = -1;  // k_or_lazy: postings of the walked clauses (-1: the launch walked every clause)
  int64_t last_counted_or_bytes = 0;     // k_or_lazy: encoded bytes + norms of the walked clauses + the lazy clauses' bitmap words
  // profiling
  std::vector<StatSlot> stats;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> free_events;
  char name[128] = {0};
};

using rucene::TermInfo;
using rucene::PreparedEntry;
using rucene::PreparedBulk;
using rucene::PreparedMap;
// a term's doc bitmap (kernels/doc_bitmap.hpp): one allocation [words | ranks | ovf | stats | freqs]
struct BitmapInfo { uint2* words; uint32_t* ranks; uint8_t* freqs; uint32_t* ovf; uint32_t* nib; uint32_t* memb; int32_t n_ovf; int32_t max_freq; int32_t df; int32_t sim_table; bool usable; };

static std::atomic<uint64_t> g_segment_uid{1};
struct rgpu_segment {
  rgpu_ctx* ctx = nullptr;
  const uint64_t uid = g_segment_uid.fetch_add(1, std::memory_order_relaxed);  // never reused (a freed segment's address may be): memo keys
  uint8_t* d_doc = nullptr;
  size_t doc_len = 0;
  uint8_t* d_norms = nullptr;       // raw norm bytes, or norm ranks when n_norm_ranks > 0
  uint8_t* d_rank_to_norm = nullptr;
  int32_t n_norm_ranks = 0;
  uint64_t* d_live = nullptr;
  int32_t max_doc = 0, doc_base = 0, version = 1;
  bool has_freqs = true;  // false: IndexOptions::Docs (no freq blocks, plain-delta tails, every freq 1)
  DevVec<int32_t> dir_last;
  DevVec<uint32_t> dir_off;
  DevVec<uint32_t> dir_row;
  DevVec<uint16_t> dir_hdr;
  DevVec<uint64_t> dir_bmax;  // per block: (freq, norm rank) frontier word (SegView::dir_bmax)
  DevVec<uint64_t> dir_sum;   // per whole chunk of 64 blocks of a term: the chunk's frontier word (SegView::dir_sum)
  bool has_positions = false;  // IndexOptions::DocsAndFreqsAndPositions: skip entries carry position pointers
  bool has_offsets = false;    // ...AndOffsets / FieldInfo::has_store_payloads: the skip entries also carry .pay words, the
  bool has_payloads = false;   // trailing VInt position block also payload bytes / offset words (kernels: read past)
  bool pay_checked = false;    // rgpu_segment_attach_payloads went through
  int skip_vals = 2;           // values per level-0 skip entry: 2, 4 (positions), 5 (+ offsets), 6 (+ payloads) — prepare.hpp A1
  DevVec<uint64_t> dir_pos;    // per block: position-stream state at the block's start (SegView::dir_pos)
  uint8_t* d_pos = nullptr;    // raw .pos bytes (rgpu_segment_attach_positions)
  size_t pos_len = 0;
  size_t dir_used = 0;
  DevVec<uint8_t> bstore;  // 16-byte aligned FullBlock payload rows of every prepared term (SegView::bstore)
  size_t bstore_used = 0;
  DevVec<uint8_t> pnorm;  // posting-order norms of every prepared term's FullBlocks and tail
  size_t pnorm_used = 0;
  mutable PreparedMap prepared;  // (a look-up may move a term from the bulk array into the table)
  rucene::FlatFpMap<BitmapInfo> bitmaps;  // doc_start_fp -> the term's doc bitmap (terms holding >= 1 doc in cfg.or_bitmaps)
  // membership bits alone (k_bitmap_memb) of terms below the full bitmaps' density that stand right behind the lead of a
  // conjunction: doc_start_fp -> {bits, doc_freq}; bits == null: the list was tried and is unusable (its clause is walked)
  struct MembOnly { uint32_t* memb; int32_t df; };
  rucene::FlatFpMap<MembOnly> memb_only;
  std::vector<void*> bitmap_allocs;
  size_t bitmap_bytes = 0;
  int64_t bitmap_terms = 0, bitmap_refused = 0;  // terms that hold a bitmap / that were filed as "walk it" (budget, allocator, unusable list)
  uint8_t* empty_bitmap = nullptr;  // all-zero {any, hi} words + ranks: the one lazy clause of a query that has no dense term (k_or_lazy wants one)
  DevVec<uint8_t> prep_scratch;  // k_skip_dir's chunk aggregates + ticket, the prefix sum's tile sums
  DevVec<uint16_t> sketch;       // block-max sketches of long terms (SegView::sketch), TERM_SKETCH_K entries each
  size_t sketch_used = 0;        // sketches in use (TermInfo::sketch = 1 + index); dropped with the prepared terms
  DevVec<SketchJob> sketch_jobs;
};

// ---- profiling helpers -----------------------------------------------------------------------------------------
static int stat_slot(rgpu_ctx* c, const char* name) {
  for (size_t i = 0; i < c->stats.size(); ++i) if (c->stats[i].name == name) return (int)i;
  c->stats.push_back(StatSlot{name, 0, 0.0, 0, {}});
  return (int)c->stats.size() - 1;
}
static hipEvent_t take_event(rgpu_ctx* c) {
  if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct TimedLaunch {
  rgpu_ctx* c;
  hipStream_t s;
  int slot;
  hipEvent_t stop = nullptr;
  bool on;
  TimedLaunch(rgpu_ctx* c_, hipStream_t s_, const char* name, int64_t postings) : c(c_), s(s_), on(c_->cfg.profile_kernels != 0) {
    slot = stat_slot(c, name);
    c->stats[(size_t)slot].launches++;
    c->stats[(size_t)slot].postings += postings;
    if (on) {
      hipEvent_t start = take_event(c);
      stop = take_event(c);
      (void)hipEventRecord(start, s);
      c->pending.push_back(PendingEvent{slot, start, stop});
    }
  }
  ~TimedLaunch() { if (on) (void)hipEventRecord(stop, s); }
};
static void drain_events(rgpu_ctx* c) {
  for (auto& pe : c->pending) {
    (void)hipEventSynchronize(pe.stop);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
      StatSlot& st = c->stats[(size_t)pe.slot];
      st.total_ms += ms;
      if (st.samples.size() < STAT_SAMPLES_MAX) st.samples.push_back(ms);
    }
    c->free_events.push_back(pe.start);
    c->free_events.push_back(pe.stop);
  }
  c->pending.clear();
}

// A deferred group's closure runs under the multi-pass state (rgpu_ctx::pass: row stride, first column, ceiling arrays) of the
// call that ENQUEUED it, whatever call happens to be running when the flags are finally looked at (a later k > 128 search whose
// scratch_take settles the earlier batch would otherwise lend the redo its own stride and ceilings: ADVICE r5, medium).
struct PassScope {
  rgpu_ctx* c;
  rgpu_ctx::Pass saved;
  PassScope(rgpu_ctx* c_, const rgpu_ctx::Pass& at_enqueue) : c(c_), saved(c_->pass) { c->pass = at_enqueue; }
  ~PassScope() { c->pass = saved; }
};

// ---- launch sets whose hand-back flags are still to be looked at (rgpu_ctx::pending_or) ------------------------------------
constexpr size_t RUNS_PER_SLOT_MAX = (size_t)1 << 28;  // ScoredPostings (2 GiB): a group with longer runs uses the context's buffer and syncs
static int32_t settle_pending(rgpu_ctx* c) {
  int32_t first = RGPU_OK;
  std::string why;
  while (!c->pending_or.empty()) {
    std::function<int32_t()> fin = std::move(c->pending_or.front());
    c->pending_or.erase(c->pending_or.begin());
    const bool was = c->defer_or;
    c->defer_or = false;  // what a settle launches (clause-order redo) finishes inside it
    const int32_t rc = fin();
    c->defer_or = was;
    if (rc != RGPU_OK && first == RGPU_OK) { first = rc; why = g_last_error; }
  }
  if (first != RGPU_OK) return fail(first, why);
  return RGPU_OK;
}

// ---- scratch slots -------------------------------------------------------------------------------------------------
// take the next slot (waiting for the work that used it N_SCRATCH calls ago) ...
static hipError_t scratch_take(rgpu_ctx* c) {
  // a slot whose flags (pinned read-back, event) are still to be looked at: that happens before it is reused (what the look
  // launches takes slots itself: the next slot is chosen afterwards)
  while (c->scr[c->scr_next].settles_later) {
    const int32_t rc = settle_pending(c);
    if (rc != RGPU_OK) { g_settle_rc = rc; g_settle_why = g_last_error; return hipErrorUnknown; }
  }
  Scratch* sc = &c->scr[c->scr_next];
  c->scr_next = (c->scr_next + 1) % N_SCRATCH;
  if (sc->busy) {
    hipError_t e = hipEventSynchronize(sc->done);
    if (e != hipSuccess) return e;
    sc->busy = false;
  }
  c->S = sc;
  if (c->last_counted == sc) { c->last_counted = nullptr; c->last_counted_words = nullptr; }  // (its counters may live in the slot's stage: gone with the reuse)
  return hipSuccess;
}
// ... and mark it in flight once everything that reads it has been enqueued on `s`
static hipError_t scratch_mark(rgpu_ctx* c, hipStream_t s) {
  Scratch* sc = c->S;
  if (!sc->done) {
    hipError_t e = hipEventCreateWithFlags(&sc->done, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipEventRecord(sc->done, s);
  if (e == hipSuccess) sc->busy = true;
  return e;
}

// the slot's staged plan -> d_stage on the upload stream; `s` (the stream the kernels go to) waits for it
static hipError_t stage_upload(rgpu_ctx* c, size_t bytes, hipStream_t s) {
  Scratch* sc = c->S;
  if (!sc->staged) {
    hipError_t e = hipEventCreateWithFlags(&sc->staged, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipMemcpyAsync(sc->d_stage.p, sc->h_stage.p, bytes, hipMemcpyHostToDevice, c->upload);
  if (e == hipSuccess) e = hipEventRecord(sc->staged, c->upload);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, sc->staged, 0);
  return e;
}

// The slot's staged plan -> d_stage, ordered on `s`. Round 6: a copy KERNEL reading the pinned buffer over PCIe, not hipMemcpyAsync: the
// DMA engine's copy of the headline TERM batch's ~210 KB plan took 11 us, and handing over from the DMA queue to the compute queue and
// back cost 8-10 us each way (rocprofv3 --kernel-trace --memory-copy-trace, scripts/step_timeline.sh: copy 11.2 | gap 8.5 | k_search_term
// 28.5 | gap 9.5 us per step on one stream) — 29 us of a 58 us step in which the kernel ran for 28.5. A kernel behind a kernel on one
// queue starts within 1-2 us. RGPU_STAGE_COPY=dma in the environment: hipMemcpyAsync as before; plans beyond 8 MB take it too.
__global__ __launch_bounds__(256) void k_stage_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size) instead of in front of every launch: it is a
// runtime call of ~1 us on the calling thread of a 30 us step
static hipError_t set_dynamic_lds_once(rgpu_ctx* c, const void* kern, size_t lds) {
  static std::mutex mu;
  static std::vector<std::tuple<const void*, int, size_t>> seen;
  std::lock_guard<std::mutex> g(mu);
  for (const auto& t : seen) if (std::get<0>(t) == kern && std::get<1>(t) == c->device && std::get<2>(t) >= lds) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) seen.emplace_back(kern, c->device, lds);
  return e;
}
// ... and for the fused single-term call the same kernel finishes the plan on the way: the per-query thresholds / counters / finished-item
// counts are ZEROED in place (not read over PCIe), and k_search_term's item descriptors — 16 bytes per item, 5.8 k of them on the headline
// batch: 93 KB the host used to write and the copy to carry — are written by one thread per query from the query, its item range and its
// item size (fill_term_item_desc's loop, on the device; it reads the HOST copy of the plan, so it does not wait for the copy next to it).
// The calling thread's share of a headline step, RGPU_HOST_TIME=1: item loop 5.5 + descriptors and zeroes 6.8 us -> 1.4 + 0.3.
struct TermPlanLayout {
  uint32_t o_q, o_p, o_sh, o_id;   // DevQuery[nq], item_prefix[nq + 1], log2 item size per query (bytes), int4 descriptors out
  uint32_t zero_from, zero_to;     // [from, to): zero-filled in d_stage instead of copied
  int32_t nq;
};
__global__ __launch_bounds__(256) void k_stage_term_plan(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n16, TermPlanLayout L) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  const uint4* s16 = reinterpret_cast<const uint4*>(src);
  uint4* d16 = reinterpret_cast<uint4*>(dst);
  const DevQuery* hq = reinterpret_cast<const DevQuery*>(src + L.o_q);
  const int64_t* hp = reinterpret_cast<const int64_t*>(src + L.o_p);
  int4* out = reinterpret_cast<int4*>(dst + L.o_id);
  // a thread's query (the grid has at least one thread per query up to 262144 of them): its words are requested BEFORE the copy loop,
  // so that the two reads of host memory — a PCIe round trip each — are in flight together instead of one behind the other
  const bool mine = tid < (size_t)L.nq;
  const size_t q0 = mine ? tid : 0;
  const DevQuery Q0 = hq[q0];
  const int64_t p00 = hp[q0], p01 = hp[q0 + 1];
  const int sh0 = (int)src[L.o_sh + q0];
  for (size_t i = tid; i < n16; i += stride) {
    const size_t off = i * 16;
    d16[i] = (off >= L.zero_from && off < L.zero_to) ? make_uint4(0u, 0u, 0u, 0u) : s16[i];
  }
  auto describe = [&](size_t q, const DevQuery& Q, int64_t p0, int64_t p1, int sh) {
    const int n_mine = 1 + (int)(p1 - p0);
    const int ft = Q.n_terms >= 1 ? Q.first_term : -1;
    const int w = n_mine | (sh << 24);
    out[q] = make_int4((int)q, 0, ft, w);
    int4* rest = out + L.nq + p0;
    for (int ch = 1; ch < n_mine; ++ch) rest[ch - 1] = make_int4((int)q, ch, ft, w);
  };
  if (mine) describe(q0, Q0, p00, p01, sh0);
  for (size_t q = tid + stride; q < (size_t)L.nq; q += stride) describe(q, hq[q], hp[q], hp[q + 1], (int)src[L.o_sh + q]);
}
static hipError_t stage_h2d(rgpu_ctx* c, size_t bytes, hipStream_t s) {
  Scratch* sc = c->S;
  if (c->upload_aside) return stage_upload(c, bytes, s);
  const size_t n16 = (bytes + 15) / 16;
  if (!c->stage_by_kernel || bytes > ((size_t)8 << 20) || n16 * 16 > sc->h_stage.cap || n16 * 16 > sc->d_stage.cap)
    return hipMemcpyAsync(sc->d_stage.p, sc->h_stage.p, bytes, hipMemcpyHostToDevice, s);
  TimedLaunch tl(c, s, "k_stage_copy", 0);
  const unsigned grid = (unsigned)std::min<size_t>(1024, (n16 + 255) / 256);
  RGPU_LAUNCH(k_stage_copy, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint4*>(sc->h_stage.p), reinterpret_cast<uint4*>(sc->d_stage.p), n16);
  return hipSuccess;
}

// ---- staging: pack several host arrays into one pinned buffer, one H2D copy ---------------------------------
struct Stager {
  rgpu_ctx* c;
  size_t used = 0;
  std::vector<std::pair<size_t, size_t>> parts;  // offset, bytes
  explicit Stager(rgpu_ctx* c_) : c(c_) {}
  size_t add(size_t bytes) {
    used = (used + 255) & ~size_t(255);
    size_t off = used;
    used += bytes;
    return off;
  }
};

static SegView seg_view(const rgpu_segment* s) {
  SegView v;
  v.doc = s->d_doc;
  v.norms = s->d_norms;
  v.rank_to_norm = s->d_rank_to_norm;
  v.pnorm = s->d_norms ? s->pnorm.p : nullptr;
  v.n_norm_ranks = s->n_norm_ranks;
  v.live = s->d_live;
  v.dir_last = s->dir_last.p;
  v.dir_off = s->dir_off.p;
  v.dir_row = s->dir_row.p;
  v.bstore = s->bstore.p;
  v.dir_hdr = s->dir_hdr.p;
  v.dir_bmax = s->dir_bmax.p;
  v.dir_sum = s->dir_sum.p;
  v.pos = s->d_pos;
  v.dir_pos = s->has_positions ? s->dir_pos.p : nullptr;
  v.sketch = s->sketch_used ? s->sketch.p : nullptr;
  v.pos_tail_flags = (s->has_payloads ? POS_TAIL_PAYLOADS : 0) | (s->has_offsets ? POS_TAIL_OFFSETS : 0);
  v.pad_ = 0;
  v.sim_tables = s->ctx->sim_tables.p;
  v.max_doc = s->max_doc;
  v.doc_base = s->doc_base;
  v.has_freqs = s->has_freqs ? 1 : 0;
  return v;
}

static int ilog8_levels(int32_t df) {  // skip_reader.rs:307-313 trim + :461-472
  int32_t t = (df % 128 == 0) ? df - 1 : df;
  if (t <= 128) return 1;
  int levels = 1;
  for (int64_t x = t / 128; x >= 8; x /= 8) levels++;
  return std::min(levels, 10);
}

// developer knob: RGPU_HOST_TIMING=1 prints where the host side of a term preparation / decode call spends its time (stderr)
struct HostClock {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  static bool on() { static const bool v = std::getenv("RGPU_HOST_TIMING") != nullptr; return v; }
  long long lap() { const auto n = std::chrono::steady_clock::now(); const long long us = (long long)std::chrono::duration_cast<std::chrono::microseconds>(n - t).count(); t = n; return us; }
};

// 0 = a term state the preparation can take (side-effect free: the bulk planner's threads ask it; validate_state words the rest)
static int state_defect(const rgpu_segment* seg, const rgpu_term_state& st) {
  if (st.doc_freq < 0) return 1;
  if (st.doc_freq <= 1) return 0;
  if (st.doc_start_fp < 0 || (size_t)st.doc_start_fp >= seg->doc_len) return 2;
  if (st.doc_freq > 128) {
    if (st.skip_offset <= 0 || (size_t)(st.doc_start_fp + st.skip_offset) >= seg->doc_len) return 3;
    if (st.skip_offset > (int64_t)0xffffffffLL) return 4;
  }
  return 0;
}
static int32_t validate_state(const rgpu_segment* seg, const rgpu_term_state& st) {
  switch (state_defect(seg, st)) {
    case 0: return RGPU_OK;
    case 1: return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative doc_freq");
    case 2: return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "doc_start_fp outside the .doc file");
    case 3: return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "skip_offset outside the .doc file");
    default: return fail(RGPU_ERR_UNSUPPORTED, "a single term's postings exceed 4 GiB");
  }
}

// Stage A of term preparation (kernels/prepare.hpp) for every not-yet-seen term with df >= 2 (ctx mutex held by the caller):
// block directory, aligned block store, decoded tail — what a decode needs. `with_norms` adds stage B (posting-order norms +
// block-max frontier words: what scoring needs) for every named term that lacks it.
// A materialising decode that is waiting for (some of) the terms: term i of the call goes to docs / freqs + out_base[i];
// fused[i] = 1 on return for the terms whose postings the preparation itself wrote there (first occurrences of terms that
// were not prepared before)
struct DecodeSink {
  const int64_t* out_base;
  int32_t *docs, *freqs;
  std::vector<uint8_t>* fused;
};
static int32_t prepare_terms_attempt(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n, bool wide, const DecodeSink* sink);
static int32_t prepare_norms_locked(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n);
// -101 from the device: a term holds EF / BITSET doc blocks whose re-packed deltas need more block-store rows than its
// file bytes suggest — plan the same call again with worst-case rows (64 per block). Nothing was committed.
// bytes of HBM the prepared-term store of a segment holds (what rgpu_segment_get_footprint reports as directory + block store +
// posting-order norms)
static size_t prepared_store_bytes(const rgpu_segment* seg) {
  const size_t per_slot = 4 + 4 + 4 + 2 + 8 + (seg->has_positions ? 8 : 0);  // dir_last, dir_off, dir_row, dir_hdr, dir_bmax (, dir_pos)
  return seg->dir_used * per_slot + seg->bstore_used + seg->pnorm_used + seg->sketch_used * (size_t)TERM_SKETCH_K * 2 + (seg->dir_used ? (seg->dir_used / 64 + 2) * 8 : 0);
}
// rgpu_config.prepared_budget_mib: a store over its ceiling is dropped as a whole BEFORE the arriving batch is planned — the
// batch then prepares what it names, like a first touch (a few hundred microseconds per thousand terms; k_prepare_blocks moves
// ~3 TB/s). Everything enqueued earlier may still read the store: the device is drained first, so this is the one place where
// an otherwise enqueue-only call waits — once per refill of the budget, not per batch. Doc bitmaps keep their own budget.
static int32_t enforce_prepared_budget(rgpu_segment* seg) {
  rgpu_ctx* c = seg->ctx;
  if (c->prepared_budget == 0 || prepared_store_bytes(seg) <= c->prepared_budget) return RGPU_OK;
  const int32_t rc_settle = settle_pending(c);  // (what an earlier batch still has to run again reads the store)
  if (rc_settle != RGPU_OK) return rc_settle;
  HIP_TRY(hipDeviceSynchronize());
  for (auto& sc : c->scr) sc.busy = false;
  for (auto& cs : c->ceil_slots) cs.busy = false;
  seg->prepared.clear();
  seg->dir_used = seg->bstore_used = seg->pnorm_used = seg->sketch_used = 0;  // the arrays keep their capacity and are refilled from the start
  c->stats[(size_t)stat_slot(c, "prepared_store_evictions")].launches += 1;  // (read by tests / callers through rgpu_kernel_stats)
  return RGPU_OK;
}
static int32_t prepare_terms_locked(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n, bool with_norms = true,
                                    const DecodeSink* sink = nullptr) {
  int32_t rc = enforce_prepared_budget(seg);
  if (rc != RGPU_OK) return rc;
  rc = prepare_terms_attempt(seg, sts, n, false, sink);
  if (rc == -101) rc = prepare_terms_attempt(seg, sts, n, true, sink);
  if (rc == -101) rc = fail(RGPU_ERR_CORRUPT_INDEX, "corrupt block framing in .doc (prepare.hpp check #12)");
  if (rc == RGPU_OK && with_norms) rc = prepare_norms_locked(seg, sts, n);
  return rc;
}
static void prep_items(const std::vector<PrepTerm>& work, std::vector<int64_t>* item_prefix, int64_t* n_items, int64_t* postings) {
  item_prefix->assign(work.size() + 1, 0);
  *n_items = 0;
  *postings = 0;
  for (size_t i = 0; i < work.size(); ++i) {
    (*item_prefix)[i] = *n_items;
    *n_items += std::max(1, (work[i].nblocks + PREP_BLOCKS_PER_ITEM - 1) / PREP_BLOCKS_PER_ITEM);  // the last item takes the tail
    *postings += work[i].df;
  }
  (*item_prefix)[work.size()] = *n_items;
}
// what one new term (df >= 2) adds to a preparation call
struct PlanOne {
  int32_t nblocks, n_entries, n_levels;
  int64_t skip_fp;
  uint64_t rows;                  // block-store rows reserved for it
  int64_t items, chunks, groups;  // work items of the block kernels; 1 KB chunks of level-0 skip bytes; groups of SKIP_GROUP chunks
};
static inline PlanOne plan_one(const rgpu_segment* seg, const rgpu_term_state& st, bool wide) {
  PlanOne o;
  o.nblocks = st.doc_freq / 128;
  o.n_entries = (st.doc_freq + 127) / 128 - 1;
  o.n_levels = ilog8_levels(st.doc_freq);
  o.skip_fp = st.doc_freq > 128 ? st.doc_start_fp + st.skip_offset : -1;
  // block store rows: a block's aligned copy is at most 28 bytes longer than its framing in the file (two
  // header bytes dropped, each all-equal VInt padded to a 16-byte row); the FullBlocks end before the skip data
  const uint64_t span = st.doc_freq > 128 ? (uint64_t)st.skip_offset : (o.nblocks ? 1026u : 0u);
  // (a docs-only field: one header byte dropped, a synthetic 16-byte freq row added per block);
  // + the decoded tail: 64 cells {doc, doc, freq, freq} behind the block rows
  o.rows = (wide ? 64u * (uint64_t)o.nblocks : (span + (seg->has_freqs ? 28u : 32u) * (uint64_t)o.nblocks + 15u) / 16u) +
           ((st.doc_freq % 128) ? (uint64_t)TAIL_STORE_ROWS : 0u);
  o.items = std::max(1, (o.nblocks + PREP_BLOCKS_PER_ITEM - 1) / PREP_BLOCKS_PER_ITEM);  // the last item takes the tail
  o.chunks = skip_chunks(o.n_entries, seg->skip_vals);
  o.groups = o.chunks > SKIP_GROUP ? skip_groups(o.chunks) : 0;
  return o;
}
static inline PrepTerm prep_term_of(const rgpu_term_state& st, const PlanOne& o, size_t dir_base, size_t batch_bs, int64_t out_base) {
  PrepTerm p;
  p.start_fp = (uint64_t)st.doc_start_fp;
  p.df = st.doc_freq;
  p.nblocks = o.nblocks;
  p.n_entries = o.n_entries;
  p.n_levels = o.n_levels;
  p.skip_fp = o.skip_fp;
  p.dir_base = (uint32_t)dir_base;
  p.pn_base = 0;                   // assigned when (if) the term's norms are prepared
  p.bs_base = (uint64_t)batch_bs;  // dir_row counts from the call's first row for every one of its terms
  p.bs_rows = (uint32_t)o.rows;
  p.out_base = out_base;
  return p;
}
// A bulk first touch planned by several host threads (host/host_threads.hpp): nothing of the segment is prepared yet and the call
// names a dictionary's worth of terms in file order (strictly ascending doc_start_fp: no repeats, so neither of the sequential
// loop's tables is needed). Pass 1 counts per contiguous range of the call's terms; the ranges' sums give every range its bases;
// pass 2 (after the staging buffer is sized) writes descriptors and prefix arrays straight into the pinned staging memory.
constexpr size_t BULK_PLAN_MIN = 32768;
struct BulkPart {
  size_t cnt = 0, slots = 0;
  uint64_t rows = 0;
  int64_t items = 0, postings = 0, chunks = 0, groups = 0, first_fp = -1, last_fp = -1;
  bool odd = false;  // a defect, a repeat or a step back: the sequential loop takes the call (and words the error, if it is one)
};
static int32_t prepare_terms_attempt(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n, bool wide, const DecodeSink* sink) {
  rgpu_ctx* c = seg->ctx;
  HostClock hc;
  long long t_plan = 0, t_reserve = 0, t_stage = 0, t_enqueue = 0, t_sync = 0, t_commit = 0;
  std::vector<PrepTerm> work;
  if (sink) sink->fused->assign(n, 0);
  size_t need_slots = seg->dir_used;
  const size_t batch_bs = (seg->bstore_used + 15) & ~size_t(15);  // this call's rows form one dense region from here
  uint64_t cap_rows = 0;
  // (empty; with the memory of the segment's last bulk call when none of that call's terms is still waiting in the array)
  PreparedBulk added = seg->prepared.bulk_live == 0 ? seg->prepared.take_array() : PreparedBulk();
  size_t n_work = 0;
  int64_t n_items = 0, postings = 0, n_chunks = 0, n_groups = 0;
  std::vector<int64_t> item_prefix, chunk_prefix, group_prefix;
  const bool none_prepared = seg->prepared.size() == 0;
  bool ascending = true;
  constexpr size_t AHEAD = 16;  // look-ups of a bulk call are asked for this many terms ahead (flat_fp_map.hpp prefetch)

  // the step both plans share: the store's arrays grown to what the call adds, the staging buffer sized
  size_t n_slots = 0, o_work = 0, o_items = 0, o_chunks = 0, o_groups = 0;
  int64_t n_tiles = 0;
  Stager st(c);
  auto size_and_reserve = [&]() -> int32_t {
    t_plan = hc.lap();
    SCRATCH_TAKE(c);  // staging below; this function ends with a stream sync, so the slot is free again on return
    HIP_TRY(seg->dir_last.reserve(need_slots, seg->dir_used, c->stream));
    HIP_TRY(seg->dir_off.reserve(need_slots, seg->dir_used, c->stream));
    HIP_TRY(seg->dir_row.reserve(need_slots, seg->dir_used, c->stream));
    HIP_TRY(seg->bstore.reserve(batch_bs + (size_t)cap_rows * 16 + 1024, seg->bstore_used, c->stream));  // + over-read padding of the row loads
    HIP_TRY(seg->dir_hdr.reserve(need_slots, seg->dir_used, c->stream));
    HIP_TRY(seg->dir_bmax.reserve(need_slots, seg->dir_used, c->stream));
    HIP_TRY(seg->dir_sum.reserve(need_slots / 64 + 2, seg->dir_used / 64 + 2, c->stream));
    if (seg->has_positions) HIP_TRY(seg->dir_pos.reserve(need_slots, seg->dir_used, c->stream));
    t_reserve = hc.lap();
    n_slots = need_slots - seg->dir_used;
    n_tiles = (int64_t)((n_slots + SCAN_TILE - 1) / SCAN_TILE);
    o_work = st.add(n_work * sizeof(PrepTerm));
    o_items = st.add((n_work + 1) * 8);
    o_chunks = st.add((n_work + 1) * 8);
    o_groups = st.add((n_work + 1) * 8);
    HIP_TRY(c->S->h_stage.reserve(st.used));
    HIP_TRY(c->S->d_stage.reserve(st.used, 0, c->stream));
    return RGPU_OK;
  };

  const int n_parts = rucene::host_threads();
  bool bulk = none_prepared && n >= BULK_PLAN_MIN && n_parts > 1;
  if (bulk) {
    std::vector<BulkPart> parts((size_t)n_parts), base((size_t)n_parts);  // a range's sums; what precedes a range
    PrepTerm* h_work = nullptr;
    int64_t *h_items = nullptr, *h_chunks = nullptr, *h_groups = nullptr;
    int32_t rc_mid = RGPU_OK;
    const size_t dir_used = seg->dir_used;
    rucene::two_pass_run(n_parts,
      [&](int t) {  // pass 1: count
        const auto range = rucene::part_range(n, n_parts, t);
        BulkPart P;
        for (size_t i = range.first; i < range.second; ++i) {
          const rgpu_term_state& ts = *sts[i];
          if (ts.doc_freq < 2) {
            if (ts.doc_freq < 0) { P.odd = true; break; }
            continue;
          }
          if (state_defect(seg, ts) != 0 || ts.doc_start_fp <= P.last_fp) { P.odd = true; break; }
          const PlanOne o = plan_one(seg, ts, wide);
          if (o.rows > 0xffffffffull) { P.odd = true; break; }
          if (P.cnt == 0) P.first_fp = ts.doc_start_fp;
          P.last_fp = ts.doc_start_fp;
          P.cnt += 1;
          P.slots += (size_t)o.nblocks + 1;
          P.rows += o.rows;
          P.items += o.items;
          P.postings += ts.doc_freq;
          P.chunks += o.chunks;
          P.groups += o.groups;
        }
        parts[(size_t)t] = P;
      },
      [&]() -> bool {  // between the passes, on this thread: totals, bases, the buffers pass 2 writes into
        int64_t last = -1;
        BulkPart acc;
        for (int t = 0; t < n_parts && bulk; ++t) {
          const BulkPart& P = parts[(size_t)t];
          if (P.odd || (P.cnt != 0 && P.first_fp <= last)) { bulk = false; break; }
          if (P.cnt != 0) last = P.last_fp;
          base[(size_t)t] = acc;
          acc.cnt += P.cnt; acc.slots += P.slots; acc.rows += P.rows; acc.items += P.items; acc.postings += P.postings;
          acc.chunks += P.chunks; acc.groups += P.groups;
        }
        // (a defect, a repeat, a step back, a store past its limits, nothing to do: the sequential loop takes the call and words it)
        if (bulk && (acc.cnt == 0 || seg->dir_used + acc.slots > 0xfffffff0ull || acc.rows > 0xfffffff0ull)) bulk = false;
        if (!bulk) return false;
        n_work = acc.cnt; need_slots = seg->dir_used + acc.slots; cap_rows = acc.rows;
        n_items = acc.items; postings = acc.postings; n_chunks = acc.chunks; n_groups = acc.groups;
        rc_mid = size_and_reserve();
        if (rc_mid != RGPU_OK) return false;
        h_work = reinterpret_cast<PrepTerm*>(c->S->h_stage.p + o_work);
        h_items = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_items);
        h_chunks = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_chunks);
        h_groups = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_groups);
        added.resize(n_work);  // (left uninitialised: pass 2 writes every entry)
        return true;
      },
      [&](int t) {  // pass 2: fill — descriptors and prefix arrays straight into the pinned staging memory
        const auto range = rucene::part_range(n, n_parts, t);
        const BulkPart& B = base[(size_t)t];
        size_t j = B.cnt, slots = dir_used + B.slots;
        int64_t items = B.items, chunks = B.chunks, groups = B.groups;
        for (size_t i = range.first; i < range.second; ++i) {
          const rgpu_term_state& ts = *sts[i];
          if (ts.doc_freq < 2) continue;
          const PlanOne o = plan_one(seg, ts, wide);
          const PrepTerm p = prep_term_of(ts, o, slots, batch_bs, sink ? sink->out_base[i] : -1);
          h_work[j] = p;
          h_items[j] = items;
          h_chunks[j] = chunks;
          h_groups[j] = groups;
          added[j] = PreparedEntry{ts.doc_start_fp, p.dir_base, p.df};
          if (sink) (*sink->fused)[i] = 1;
          ++j;
          slots += (size_t)o.nblocks + 1;
          items += o.items;
          chunks += o.chunks;
          groups += o.groups;
        }
      });
    if (rc_mid != RGPU_OK) return rc_mid;
    if (bulk) {
      h_items[n_work] = n_items;
      h_chunks[n_work] = n_chunks;
      h_groups[n_work] = n_groups;
      c->stats[(size_t)stat_slot(c, "prepare_bulk_plans")].launches += 1;  // (tests ask whether this path ran: rgpu_kernel_stats)
    }
  }
  if (!bulk) {
    // Two tables stand between a term and the work list: "prepared already" and "named earlier in this call". A first touch in
    // file order pays two DRAM round trips per term for answers that are known in advance — nothing is prepared yet, and while
    // the doc_start_fps of the call ascend strictly no term can repeat: the second table is only built (from the work list) once
    // one does not.
    rucene::FlatFpMap<int> in_batch;
    int64_t last_fp = -1;
    if (n > 4096) { work.reserve(n); added.reserve(n); }
    for (size_t i = 0; i < n; ++i) {
      if (i + AHEAD < n) {
        if (!none_prepared) seg->prepared.prefetch(sts[i + AHEAD]->doc_start_fp);
        if (!ascending) in_batch.prefetch(sts[i + AHEAD]->doc_start_fp);
      }
      const rgpu_term_state& st = *sts[i];
      if (st.doc_freq < 2) {  // a singleton lives in the term dictionary; everything else has blocks and / or a tail
        if (st.doc_freq < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative doc_freq");
        continue;
      }
      if (!none_prepared) {
        if (const TermInfo* known = seg->prepared.find(st.doc_start_fp)) {
          if (known->df != st.doc_freq) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term state changed doc_freq for a known doc_start_fp");
          continue;
        }
      }
      int32_t rc = validate_state(seg, st);
      if (rc != RGPU_OK) return rc;
      if (ascending && st.doc_start_fp > last_fp) {
        last_fp = st.doc_start_fp;
      } else {
        if (ascending) {  // the order broke: from here on (and for what came before) the table decides
          ascending = false;
          in_batch.reserve_more(n);
          for (const auto& a : added) in_batch.put(a.first, 1);
        }
        if (in_batch.find(st.doc_start_fp)) continue;
        in_batch.put(st.doc_start_fp, 1);
      }
      const PlanOne o = plan_one(seg, st, wide);
      if (o.rows > 0xffffffffull) return fail(RGPU_ERR_UNSUPPORTED, "a single term's postings exceed 64 GiB");
      const PrepTerm p = prep_term_of(st, o, need_slots, batch_bs, sink ? sink->out_base[i] : -1);
      if (sink) (*sink->fused)[i] = 1;
      cap_rows += o.rows;
      need_slots += (size_t)p.nblocks + 1;
      if (need_slots > 0xfffffff0ull) return fail(RGPU_ERR_UNSUPPORTED, "block directory exceeds 2^32 slots");
      work.push_back(p);
      added.push_back(PreparedEntry{st.doc_start_fp, p.dir_base, p.df});
    }
    if (work.empty()) return RGPU_OK;
    if (cap_rows > 0xfffffff0ull) return fail(RGPU_ERR_UNSUPPORTED, "one call prepares more than 64 GiB of block store: split the term list");
    n_work = work.size();
    // plans: (term, 1 KB chunk of level-0 skip bytes) for k_skip_dir; (term, chunk of blocks) for the block kernels
    // (+ (term, SKIP_GROUP chunks) for k_skip_groups: terms of more than SKIP_GROUP chunks only, the others are zero-width in group_prefix)
    chunk_prefix.resize(n_work + 1);
    group_prefix.resize(n_work + 1);
    prep_items(work, &item_prefix, &n_items, &postings);
    for (size_t i = 0; i < n_work; ++i) {
      const int64_t mine = skip_chunks(work[i].n_entries, seg->skip_vals);
      chunk_prefix[i] = n_chunks;
      group_prefix[i] = n_groups;
      n_chunks += mine;
      if (mine > SKIP_GROUP) n_groups += skip_groups(mine);
    }
    chunk_prefix[n_work] = n_chunks;
    group_prefix[n_work] = n_groups;
    const int32_t rc_size = size_and_reserve();
    if (rc_size != RGPU_OK) return rc_size;
    std::memcpy(c->S->h_stage.p + o_work, work.data(), n_work * sizeof(PrepTerm));
    std::memcpy(c->S->h_stage.p + o_items, item_prefix.data(), item_prefix.size() * 8);
    std::memcpy(c->S->h_stage.p + o_chunks, chunk_prefix.data(), chunk_prefix.size() * 8);
    std::memcpy(c->S->h_stage.p + o_groups, group_prefix.data(), group_prefix.size() * 8);
  }
  HIP_TRY(hipMemcpyAsync(c->S->d_stage.p, c->S->h_stage.p, st.used, hipMemcpyHostToDevice, c->stream));
  // device scratch: [-, total rows][chunk aggregates][group aggregates][level-0 starts][tile sums]
  const size_t o_aggs = 64, o_gaggs = o_aggs + (size_t)n_chunks * sizeof(SkipAgg), o_l0 = o_gaggs + (size_t)n_groups * sizeof(SkipAgg),
               o_tiles = o_l0 + n_work * 8;
  HIP_TRY(seg->prep_scratch.reserve(o_tiles + (size_t)n_tiles * 8 + 64, 0, c->stream));
  HIP_TRY(hipMemsetAsync(seg->prep_scratch.p, 0, 64, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_err, 0, 4 * sizeof(int), c->stream));
  t_stage = hc.lap();
  unsigned long long* d_ticket = reinterpret_cast<unsigned long long*>(seg->prep_scratch.p);
  unsigned long long* d_total = d_ticket + 1;
  SkipAgg* d_aggs = reinterpret_cast<SkipAgg*>(seg->prep_scratch.p + o_aggs);
  SkipAgg* d_gaggs = reinterpret_cast<SkipAgg*>(seg->prep_scratch.p + o_gaggs);
  int64_t* d_l0 = reinterpret_cast<int64_t*>(seg->prep_scratch.p + o_l0);
  unsigned long long* d_tiles = reinterpret_cast<unsigned long long*>(seg->prep_scratch.p + o_tiles);
  const PrepTerm* d_work = reinterpret_cast<const PrepTerm*>(c->S->d_stage.p + o_work);
  const int64_t* d_items = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_items);
  const int64_t* d_chunks = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_chunks);
  const int64_t* d_groups = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_groups);
  const bool legacy = seg->version < 1;
  const unsigned item_grid = wg_count((n_items + PREP_WAVES - 1) / PREP_WAVES);
  {
    TimedLaunch tl(c, c->stream, "k_skip_dir", postings);
    const dim3 grid(wg_count((n_chunks + PREP_WAVES - 1) / PREP_WAVES));
    RGPU_LAUNCH(k_skip_terms, dim3(wg_count((n_work + PREP_THREADS - 1) / PREP_THREADS)), dim3(PREP_THREADS), 0, c->stream, seg->d_doc,
                       (int64_t)seg->doc_len, d_work, (int)n_work, d_l0, seg->dir_last.p, seg->dir_off.p,
                       seg->has_positions ? seg->dir_pos.p : nullptr, seg->skip_vals, c->d_err);
    if (n_chunks > 0) {  // the terms whose level 0 one lane does not finish
      auto pass = [&](auto kern) {
        RGPU_LAUNCH(kern, grid, dim3(PREP_THREADS), 0, c->stream, seg->d_doc, (int64_t)seg->doc_len, (int64_t)seg->doc_len + 8192,
                           d_work, d_chunks, d_l0, (int)n_work, n_chunks, d_aggs, d_groups, d_gaggs, seg->dir_last.p, seg->dir_off.p,
                           seg->has_positions ? seg->dir_pos.p : nullptr, c->d_err);
      };
      auto groups = [&](auto kern) {
        if (n_groups > 0)
          RGPU_LAUNCH(kern, dim3(wg_count((n_groups + PREP_WAVES - 1) / PREP_WAVES)), dim3(PREP_THREADS), 0, c->stream, d_chunks,
                             d_groups, (int)n_work, n_groups, d_aggs, d_gaggs);
      };
      switch (seg->skip_vals) {  // values per level-0 skip entry (prepare.hpp, A1)
        case 2: pass(k_skip_dir<1, 2>); groups(k_skip_groups<2>); pass(k_skip_dir<2, 2>); break;
        case 4: pass(k_skip_dir<1, 4>); groups(k_skip_groups<4>); pass(k_skip_dir<2, 4>); break;
        case 5: pass(k_skip_dir<1, 5>); groups(k_skip_groups<5>); pass(k_skip_dir<2, 5>); break;
        default: pass(k_skip_dir<1, 6>); groups(k_skip_groups<6>); pass(k_skip_dir<2, 6>); break;
      }
    }
  }
  {
    TimedLaunch tl(c, c->stream, "k_block_headers", postings);
    auto go = [&](auto kern) {
      RGPU_LAUNCH(kern, dim3(wg_count((n_slots + PREP_THREADS - 1) / PREP_THREADS)), dim3(PREP_THREADS), 0, c->stream, seg->d_doc,
                         (int64_t)seg->doc_len, d_work, (int)n_work, (uint32_t)seg->dir_used, (int64_t)n_slots, seg->dir_last.p, seg->dir_off.p,
                         seg->dir_row.p, seg->dir_hdr.p, seg->has_freqs ? 1 : 0, c->d_err);
    };
    if (legacy) go(k_block_headers<true>); else go(k_block_headers<false>);
  }
  {
    TimedLaunch tl(c, c->stream, "k_scan_rows", 0);
    uint32_t* rows = seg->dir_row.p + seg->dir_used;
    RGPU_LAUNCH(k_scan_reduce, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, c->stream, rows, (int64_t)n_slots, d_tiles);
    RGPU_LAUNCH(k_scan_tiles, dim3(1), dim3(PREP_THREADS), 0, c->stream, d_tiles, n_tiles, (unsigned long long)cap_rows, d_total, c->d_err);
    RGPU_LAUNCH(k_scan_down, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, c->stream, rows, (int64_t)n_slots, d_tiles);
  }
  {
    TimedLaunch tl(c, c->stream, "k_prepare_blocks", postings);
    auto go = [&](auto kern) {
      RGPU_LAUNCH(kern, dim3(item_grid), dim3(PREP_THREADS), 0, c->stream, seg->d_doc, d_work, d_items, (int)n_work, n_items,
                         seg->dir_last.p, seg->dir_off.p, seg->dir_row.p, seg->dir_hdr.p, seg->bstore.p, seg->dir_bmax.p, seg->has_freqs ? 1 : 0,
                         seg->max_doc, c->d_err, sink ? sink->docs : (int32_t*)nullptr, sink ? sink->freqs : (int32_t*)nullptr);
    };
    if (legacy) go(k_prepare_blocks<true>); else go(k_prepare_blocks<false>);
  }
  int err4[4] = {0, 0, 0, 0};
  unsigned long long total_rows = 0;
  t_enqueue = hc.lap();
  // The terms are filed as prepared WHILE the kernels run (the context's mutex is held: nobody looks before this call returns):
  // 152 k insertions into a table that is grown and first touched here — 2 to 6 ms of host time that used to follow the
  // 1.3 ms of kernels instead of hiding behind them. A failing call takes them back (remove_keys: a rebuild, the rare path).
  const bool as_bulk = ascending && added.size() >= 4096;  // file order, no repeats: the array itself is the index (PreparedMap)
  std::vector<int64_t> added_keys;
  if (as_bulk) {
    seg->prepared.adopt_sorted(std::move(added), (uint64_t)batch_bs, seg->d_norms == nullptr);
  } else {
    added_keys.resize(added.size());
    seg->prepared.reserve_more(added.size());
    for (size_t i = 0; i < added.size(); ++i) {
      if (i + AHEAD < added.size()) seg->prepared.prefetch(added[i + AHEAD].first);
      seg->prepared.put(added[i].first, PreparedMap::expand(added[i], (uint64_t)batch_bs, seg->d_norms == nullptr));
      added_keys[i] = added[i].first;
    }
  }
  t_commit = hc.lap();
  // (behind the commit: a copy into pageable host memory returns when it is done, i.e. after the kernels)
  hipError_t e_copy = hipMemcpyAsync(err4, c->d_err, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (e_copy == hipSuccess) e_copy = hipMemcpyAsync(&total_rows, d_total, 8, hipMemcpyDeviceToHost, c->stream);
  // The commit above is provisional until the kernels have been verified: EVERY exit from here on takes the terms back unless it
  // disarms the guard (the rollback used to be a call each failing branch had to remember).
  struct Rollback {
    rgpu_segment* seg; bool as_bulk; const std::vector<int64_t>* keys; const DecodeSink* sink; size_t n; bool armed;
    ~Rollback() {
      if (!armed) return;
      if (as_bulk) seg->prepared.drop_bulk(); else seg->prepared.remove_keys(keys->data(), keys->size());
      if (sink) sink->fused->assign(n, 0);
    }
  } provisional{seg, as_bulk, &added_keys, sink, n, true};
  hipError_t e_sync = e_copy == hipSuccess ? hipStreamSynchronize(c->stream) : e_copy;
  if (e_sync == hipSuccess) e_sync = launch_status();
  if (e_sync != hipSuccess) {
    (void)hipStreamSynchronize(c->stream);
    return fail(RGPU_ERR_RUNTIME, std::string("term preparation: ") + hipGetErrorString(e_sync));
  }
  t_sync = hc.lap();
  const int err = err4[0];
  if (err == -101) return -101;  // see prepare_terms_locked
  if (err != 0) {
    return fail(err, err == RGPU_ERR_UNSUPPORTED ? std::string("FULL-encoded doc block (unimplemented in Rucene itself), or an EF / BITSET block in a legacy (.doc version 0) file")
                                                 : "corrupt skip data or block framing in .doc (prepare.hpp check #" + std::to_string(err4[1]) + ")");
  }
  provisional.armed = false;  // verified: the terms stay
  seg->dir_used = need_slots;
  seg->bstore_used = batch_bs + (size_t)total_rows * 16;
  if (HostClock::on())
    std::fprintf(stderr, "[prepare host] %zu terms (%s): plan %lld us, reserve (hipMalloc / grow) %lld, plan chunks + stage + H2D %lld, enqueue %lld, commit (under the kernels) %lld, kernels + sync %lld\n",
                 n_work, bulk ? "bulk plan, host threads" : "one thread", t_plan, t_reserve, t_stage, t_enqueue, t_commit, t_sync);
  return RGPU_OK;
}

// Stage B for the named terms that are prepared but have no posting-order norms yet (segments without norms have none to prepare)
static int32_t prepare_norms_locked(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n) {
  rgpu_ctx* c = seg->ctx;
  if (!seg->d_norms) return RGPU_OK;
  std::vector<PrepTerm> work;
  std::vector<int64_t> fps;
  size_t need_pn = seg->pnorm_used;
  rucene::FlatFpMap<int> in_batch;
  for (size_t i = 0; i < n; ++i) {
    const rgpu_term_state& st = *sts[i];
    if (st.doc_freq < 2) continue;
    const TermInfo* info = seg->prepared.find(st.doc_start_fp);
    if (!info || info->norms || in_batch.find(st.doc_start_fp)) continue;
    in_batch.put(st.doc_start_fp, 1);
    PrepTerm p{};
    p.start_fp = (uint64_t)st.doc_start_fp;
    p.df = info->df;
    p.nblocks = info->nblocks;
    p.dir_base = info->dir_base;
    p.bs_base = info->bs_base;
    p.pn_base = (uint64_t)need_pn;
    need_pn += ((size_t)p.nblocks + ((p.df % 128) ? 1u : 0u)) * 128;  // the tail's norms follow the FullBlocks'
    work.push_back(p);
    fps.push_back(st.doc_start_fp);
  }
  if (work.empty()) return RGPU_OK;
  SCRATCH_TAKE(c);
  HIP_TRY(seg->pnorm.reserve(need_pn + 64, seg->pnorm_used, c->stream));
  std::vector<int64_t> item_prefix;
  int64_t n_items = 0, postings = 0;
  prep_items(work, &item_prefix, &n_items, &postings);
  Stager st(c);
  const size_t o_work = st.add(work.size() * sizeof(PrepTerm));
  const size_t o_items = st.add(item_prefix.size() * 8);
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, c->stream));
  std::memcpy(c->S->h_stage.p + o_work, work.data(), work.size() * sizeof(PrepTerm));
  std::memcpy(c->S->h_stage.p + o_items, item_prefix.data(), item_prefix.size() * 8);
  HIP_TRY(hipMemcpyAsync(c->S->d_stage.p, c->S->h_stage.p, st.used, hipMemcpyHostToDevice, c->stream));
  {
    TimedLaunch tl(c, c->stream, "k_prepare_norms", postings);
    const unsigned grid = wg_count((n_items + PREP_WAVES - 1) / PREP_WAVES);
    auto go = [&](auto kern) {
      RGPU_LAUNCH(kern, dim3(grid), dim3(PREP_THREADS), 0, c->stream, seg_view(seg), reinterpret_cast<const PrepTerm*>(c->S->d_stage.p + o_work),
                         reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_items), (int)work.size(), n_items, seg->pnorm.p, seg->dir_bmax.p,
                         seg->n_norm_ranks > 0 ? 1 : 0);
    };
    if (seg->version < 1) go(k_prepare_norms<true>); else go(k_prepare_norms<false>);
  }
  {  // ... and one level up: the frontier of every whole chunk of 64 blocks (what k_search_term tests first)
    TimedLaunch tl(c, c->stream, "k_chunk_frontiers", postings / 128);
    const unsigned grid = wg_count((n_items + PREP_WAVES - 1) / PREP_WAVES);
    RGPU_LAUNCH(k_chunk_frontiers, dim3(grid), dim3(PREP_THREADS), 0, c->stream, reinterpret_cast<const PrepTerm*>(c->S->d_stage.p + o_work),
                reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_items), (int)work.size(), n_items, seg->dir_bmax.p, seg->dir_sum.p);
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(launch_status());
  seg->pnorm_used = need_pn;
  for (size_t i = 0; i < work.size(); ++i) {
    TermInfo info = *seg->prepared.find(fps[i]);
    info.pn_base = work[i].pn_base;
    info.norms = true;
    seg->prepared.put(fps[i], info);
  }
  return RGPU_OK;
}

// `known`: the term's TermInfo when the caller has looked it up already (search_pass: ONE table look-up per clause and call — the
// validation, the norms check and this function used to take one each; at 3072 clauses per conjunction batch the host's share of a
// step was 0.17 ms next to 0.25 ms of kernels)
static int32_t make_dev_term(const rgpu_segment* seg, const rgpu_term_state& st, float weight, int32_t sim_table, DevTerm* out, bool need_norms = true,
                             const TermInfo* known = nullptr) {
  DevTerm t;
  std::memset(&t, 0, sizeof t);
  t.start_fp = (uint64_t)std::max<int64_t>(0, st.doc_start_fp);
  t.df = st.doc_freq;
  t.weight = weight;
  t.sim_table = sim_table;
  t.flags = (sim_table >= 0 && (size_t)sim_table < seg->ctx->sim_monotone.size() && seg->ctx->sim_monotone[(size_t)sim_table]) ? TERM_FLAG_MONOTONE : 0u;
  t.singleton_doc = st.singleton_doc_id;
  t.singleton_freq = seg->has_freqs ? (int32_t)st.total_term_freq : 1;  // posting_reader.rs:483: total_term_freq = doc_freq without freqs
  if (st.doc_freq == 1 && (st.singleton_doc_id < 0 || st.singleton_doc_id >= seg->max_doc))
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "singleton_doc_id out of range");
  if (st.doc_freq >= 2) {
    const TermInfo* info = known ? known : seg->prepared.find(st.doc_start_fp);
    if (!info) return fail(RGPU_ERR_ILLEGAL_STATE, "term not prepared");
    if (need_norms && !info->norms) return fail(RGPU_ERR_ILLEGAL_STATE, "term's posting-order norms not prepared");
    t.dir_base = info->dir_base;
    t.nblocks = info->nblocks;
    t.pn_base = info->pn_base;
    t.bs_base = info->bs_base;
    t.sketch = info->sketch;
  }
  t.tail_n = st.doc_freq > 1 ? st.doc_freq % 128 : 0;
  *out = t;
  return RGPU_OK;
}

// ---- context ---------------------------------------------------------------------------------------------------
extern "C" int32_t rgpu_abi_version(void) { return RGPU_ABI_VERSION; }

extern "C" const char* rgpu_last_error(rgpu_ctx*) { return g_last_error.c_str(); }

extern "C" int32_t rgpu_init(int32_t device_ordinal, const rgpu_config* cfg, rgpu_ctx** out_ctx) {
  if (!out_ctx) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out_ctx is null");
  *out_ctx = nullptr;
  if (cfg && cfg->abi_version != RGPU_ABI_VERSION) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rgpu_config.abi_version mismatch");
  if (cfg && (cfg->prepared_budget_mib < 0 || cfg->or_deferred < 0 || cfg->or_deferred > 1)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rgpu_config.prepared_budget_mib must be >= 0, or_deferred 0 or 1");
  if (cfg && (cfg->comm_force_gather < 0 || cfg->comm_force_gather > 1)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rgpu_config.comm_force_gather must be 0 or 1");
  if (cfg && cfg->bitmap_budget_mib < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rgpu_config.bitmap_budget_mib must be >= 0 (or_bitmaps / and_bitmaps = -1 turn bitmaps off)");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(RGPU_ERR_RUNTIME, "no HIP device present: this library has no CPU fallback");
  if (device_ordinal < 0 || device_ordinal >= count) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RGPU_ERR_RUNTIME, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  rgpu_ctx* c = new rgpu_ctx();
  c->device = device_ordinal;
  if (cfg) c->cfg = *cfg;
  c->cfg.abi_version = RGPU_ABI_VERSION;
  if (c->cfg.blocks_per_item <= 0) { c->cfg.blocks_per_item = 32; c->blocks_per_item_auto = true; }
  // lead blocks per item. With the dense clauses answered through bitmaps a lead block costs a fraction of what it did when
  // every clause was walked, and an item's fixed cost (descriptors, directory windows, one more list for the merge) weighs
  // more: on the 1024 x 3-term batch k_search_and + k_merge_items take 0.73 + 0.09 ms at 2, 0.475 + 0.053 at 4 (round 2's
  // choice), 0.412 + 0.040 at 6, 0.390 + 0.034 at 8, 0.398 + 0.028 at 12, 0.436 + 0.024 at 16, 0.60 + 0.02 at 32
  // Round 5 (batched first probe): the items of a launch are sized from its lead blocks — and_item_blocks() — unless the caller
  // pins them: on one box k_search_and took 0.357 ms at 8, 0.285 at 12, 0.304 at 16, 0.339 at 20, 0.39 at 24 (10 M docs: 320 k
  // lead blocks) and 2.76 at 32, 2.54 at 48, 2.35 at 60 (100 M docs: 3.2 M lead blocks)
  if (c->cfg.and_blocks_per_item <= 0) { c->cfg.and_blocks_per_item = 8; c->and_blocks_per_item_auto = true; }
  // doc bitmaps are an optional accelerator: they get a byte budget (default: an eighth of the device's memory — 36 GB of an
  // MI355X's 288), and a term past it stays a walked clause (ensure_bitmaps_locked)
  c->bitmap_budget = c->cfg.bitmap_budget_mib > 0 ? (size_t)c->cfg.bitmap_budget_mib << 20 : (size_t)prop.totalGlobalMem / 8;
  c->prepared_budget = c->cfg.prepared_budget_mib > 0 ? (size_t)c->cfg.prepared_budget_mib << 20 : 0;
  if (const char* e = std::getenv("RGPU_TERM_SKETCH")) c->term_sketches = std::atoi(e) != 0;
  if (const char* e = std::getenv("RGPU_UPLOAD_ASIDE")) c->upload_aside = std::atoi(e) != 0;
  if (const char* e = std::getenv("RGPU_AND_MEMB_ONLY")) c->memb_only_on = std::atoi(e) != 0;
  if (const char* e = std::getenv("RGPU_TERM_TARGET_ITEMS")) c->term_target_items = std::max(256, std::atoi(e));
  if (const char* e = std::getenv("RGPU_HOST_TIME")) c->host_time = std::atoi(e) != 0;
  if (const char* e = std::getenv("RGPU_STAGE_COPY")) c->stage_by_kernel = std::strcmp(e, "dma") != 0;
  if (const char* e = std::getenv("RGPU_TERM_FOLD")) c->term_fold = std::atoi(e) != 0;
  if (const char* e = std::getenv("RGPU_TERM_MIN_ITEM_BLOCKS")) c->term_min_item_blocks = std::max(8, std::min(4096, std::atoi(e)));
  if (const char* e = std::getenv("RGPU_TERM_SPLIT")) c->term_split = std::max(1, std::min(64, std::atoi(e)));
  if (const char* e = std::getenv("RGPU_COMM_FORCE_GATHER")) { if (std::atoi(e) != 0) c->cfg.comm_force_gather = 1; }
  std::snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
  // (the upload stream only exists when it is asked for: HIP multiplexes streams onto four hardware queues, and a fifth stream in the
  // process — two caller streams + the context's + torch's + this one — is a suspect for the two caller streams sharing one)
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      (c->upload_aside && hipStreamCreateWithFlags(&c->upload, hipStreamNonBlocking) != hipSuccess) ||
      hipMalloc(&c->d_err, 4 * sizeof(int)) != hipSuccess) {
    delete c;
    return fail(RGPU_ERR_RUNTIME, "failed to create stream / error word");
  }
  *out_ctx = c;
  return RGPU_OK;
}

extern "C" void rgpu_shutdown(rgpu_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->pending_or.clear();  // (nobody will read those batches any more)
  (void)hipStreamSynchronize(c->stream);
  drain_events(c);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  c->sim_tables.release(); for (auto& cs : c->ceil_slots) { cs.d.release(); if (cs.done) (void)hipEventDestroy(cs.done); } c->d_runs.release(); c->pos_counts.release(); c->pos_tiles.release(); c->phrase_docs.release(); c->phrase_keys.release(); c->phrase_redo.release(); c->phrase_count.release(); c->host_api_hits.release(); c->host_api_totals.release(); c->host_api_rows.release();
  for (auto& sc : c->scr) sc.release();
  if (c->d_err) (void)hipFree(c->d_err);
  if (c->upload) { (void)hipStreamSynchronize(c->upload); (void)hipStreamDestroy(c->upload); }
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int32_t rgpu_device_name(rgpu_ctx* c, char* buf, size_t len) {
  if (!c || !buf || len == 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::snprintf(buf, len, "%s", c->name);
  return RGPU_OK;
}

extern "C" int32_t rgpu_synchronize(rgpu_ctx* c) {
  if (!c) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "ctx is null");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  const int32_t rc = settle_pending(c);  // (rgpu_config.or_deferred: flags of OR batches nobody has looked at yet; waits for those batches)
  if (rc != RGPU_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RGPU_OK;
}

extern "C" int32_t rgpu_kernel_stats(rgpu_ctx* c, rgpu_kernel_stat* out, int32_t max_out) {
  if (!c) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "ctx is null");
  std::lock_guard<std::mutex> g(c->mu);
  (void)hipSetDevice(c->device);
  drain_events(c);
  int32_t n = 0;
  for (auto& s : c->stats) {
    if (n >= max_out) break;
    std::memset(&out[n], 0, sizeof(rgpu_kernel_stat));
    std::snprintf(out[n].name, sizeof out[n].name, "%s", s.name.c_str());
    out[n].launches = s.launches;
    out[n].total_ms = s.total_ms;
    out[n].postings = s.postings;
    out[n].timed_launches = (int64_t)s.samples.size();
    if (!s.samples.empty()) {
      std::vector<float> v(s.samples);
      std::sort(v.begin(), v.end());
      out[n].min_ms = v.front();
      out[n].max_ms = v.back();
      out[n].median_ms = (v.size() & 1) ? v[v.size() / 2] : 0.5 * ((double)v[v.size() / 2 - 1] + (double)v[v.size() / 2]);
    }
    ++n;
  }
  return n;
}

extern "C" int32_t rgpu_set_profiling(rgpu_ctx* c, int32_t on) {
  if (!c) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "ctx is null");
  std::lock_guard<std::mutex> g(c->mu);
  (void)hipSetDevice(c->device);
  drain_events(c);
  c->cfg.profile_kernels = on ? 1 : 0;
  return RGPU_OK;
}

extern "C" int32_t rgpu_and_touched_bytes(rgpu_ctx* c, int64_t* bytes_out) {
  if (!c || !bytes_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  *bytes_out = 0;
  if (!c->last_and || c->last_and_queries <= 0) return RGPU_OK;
  if (c->last_and->busy) { HIP_TRY(hipEventSynchronize(c->last_and->done)); c->last_and->busy = false; }
  std::vector<unsigned long long> h((size_t)c->last_and_queries);
  HIP_TRY(hipMemcpy(h.data(), c->last_and->d_touched.p, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long sum = 0;
  for (auto v : h) sum += v;
  *bytes_out = (int64_t)sum;
  return RGPU_OK;
}

// What the most recent TERM / AND / (>= 10 clause) OR launch on this context really did (waits for it): FullBlocks decoded,
// their encoded bytes (+ 1 norm byte per posting of a TERM block, + 14 directory bytes per block a pruned TERM launch looked
// at: row, header and frontier words), postings decoded = 128 per decoded block + the prepared tails / singletons read.
extern "C" int32_t rgpu_last_search_counters(rgpu_ctx* c, rgpu_search_counters* out) {
  if (!c || !out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  std::memset(out, 0, sizeof *out);
  out->op = c->last_counted_op;
  out->postings_covered = c->last_counted_postings;
  if (c->last_counted_op == RGPU_OP_OR) {  // k_or_wide: every posting decoded (blocks that straddle windows more than once)
    out->postings_decoded = c->last_counted_or_decoded >= 0 ? c->last_counted_or_decoded : c->last_counted_postings;
    if (c->last_counted_or_decoded >= 0) out->touched_bytes = c->last_counted_or_bytes;  // k_or_lazy: walked clauses + bitmap words
    return RGPU_OK;
  }
  if (!c->last_counted || c->last_counted_queries <= 0) return RGPU_OK;
  if (c->last_counted->busy) { HIP_TRY(hipEventSynchronize(c->last_counted->done)); c->last_counted->busy = false; }
  const size_t nq = (size_t)c->last_counted_queries;
  std::vector<unsigned long long> h(nq * 2);
  HIP_TRY(hipMemcpy(h.data(), c->last_counted_words ? c->last_counted_words : c->last_counted->d_touched.p, nq * 16, hipMemcpyDeviceToHost));
  for (size_t q = 0; q < nq; ++q) { out->touched_bytes += (int64_t)h[q]; out->blocks_decoded += (int64_t)h[nq + q]; }
  out->postings_decoded = 128 * out->blocks_decoded + c->last_counted_loose;
  return RGPU_OK;
}

extern "C" void rgpu_kernel_stats_reset(rgpu_ctx* c) {
  if (!c) return;
  std::lock_guard<std::mutex> g(c->mu);
  (void)hipSetDevice(c->device);
  drain_events(c);
  c->stats.clear();
}

// ---- similarity tables ---------------------------------------------------------------------------------------
extern "C" int32_t rgpu_sim_table_upload(rgpu_ctx* c, const float cache[256], float k1) {
  if (!c || !cache) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(c->sim_tables.reserve((size_t)(c->n_sim_tables + 1) * 257, (size_t)c->n_sim_tables * 257, c->stream));
  float tmp[257];
  std::memcpy(tmp, cache, 256 * sizeof(float));
  tmp[256] = k1;
  HIP_TRY(hipMemcpyAsync(c->sim_tables.p + (size_t)c->n_sim_tables * 257, tmp, sizeof tmp, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // BM25's cache falls as the norm byte grows (shorter doc): then a score never falls as the norm rank grows, which is
  // what the per-block (freq, rank) frontier bound of the TERM kernel relies on. Any other table simply runs unpruned.
  bool mono = k1 >= 0.0f;
  for (int i = 0; i < 256 && mono; ++i) mono = cache[i] >= 0.0f && cache[i] <= 3.0e38f && (i == 0 || cache[i] <= cache[i - 1]);
  c->sim_monotone.push_back(mono ? 1 : 0);
  bool nonneg = k1 >= 0.0f && k1 <= 3.0e38f;
  for (int i = 0; i < 256 && nonneg; ++i) nonneg = cache[i] >= 0.0f && cache[i] <= 3.0e38f;
  c->sim_k1.push_back(k1);
  c->sim_nonneg.push_back(nonneg ? 1 : 0);
  float cmin = cache[0];
  for (int i = 1; i < 256; ++i) cmin = std::min(cmin, cache[i]);
  c->sim_cache_min.push_back(cmin);
  return c->n_sim_tables++;
}

// ---- segment ---------------------------------------------------------------------------------------------------
extern "C" int32_t rgpu_segment_upload(rgpu_ctx* c, const uint8_t* doc_file, size_t doc_len, const uint8_t* norms,
                                       int32_t max_doc, int32_t doc_base, const uint64_t* live_docs, rgpu_segment** out_seg) {
  return rgpu_segment_upload_field(c, doc_file, doc_len, norms, max_doc, doc_base, live_docs, 2 /* DocsAndFreqs */, out_seg);
}

extern "C" int32_t rgpu_segment_upload_field(rgpu_ctx* c, const uint8_t* doc_file, size_t doc_len, const uint8_t* norms,
                                             int32_t max_doc, int32_t doc_base, const uint64_t* live_docs, int32_t index_options,
                                             rgpu_segment** out_seg) {
  if (!c || !doc_file || !out_seg) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  *out_seg = nullptr;
  const bool payloads = (index_options & RGPU_FIELD_STORES_PAYLOADS) != 0;
  index_options &= ~RGPU_FIELD_STORES_PAYLOADS;
  if (index_options < 1 || index_options > 4)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "index_options must be 1 (Docs), 2 (DocsAndFreqs), 3 (DocsAndFreqsAndPositions) or 4 (...AndOffsets), "
                                           "optionally | RGPU_FIELD_STORES_PAYLOADS");
  if (payloads && index_options < 3)  // FieldInfo::check_consistency (field_infos/mod.rs): payloads need positions
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "a field that stores payloads is indexed with positions (index_options >= 3)");
  if (max_doc < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative max_doc");
  rucene::DocFileInfo info;
  std::string why;
  int rc = rucene::parse_doc_file(doc_file, doc_len, &info, &why);
  if (rc != 0) return fail(rc, why);
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  rgpu_segment* s = new rgpu_segment();
  s->ctx = c;
  s->doc_len = doc_len;
  s->max_doc = max_doc;
  s->doc_base = doc_base;
  s->version = info.version;
  s->has_freqs = index_options >= 2;
  s->has_positions = index_options >= 3;
  s->has_offsets = index_options >= 4;
  s->has_payloads = payloads;
  s->skip_vals = !s->has_positions ? 2 : payloads ? 6 : s->has_offsets ? 5 : 4;  // skip_writer.rs:261-289
  const size_t pad = 8192;  // speculative row / tail loads may run past the last posting byte
  auto bail = [&](hipError_t e, const char* what) { rgpu_segment_free(s); return fail(RGPU_ERR_RUNTIME, std::string(what) + ": " + hipGetErrorString(e)); };
  hipError_t e;
  if ((e = hipMalloc(&s->d_doc, doc_len + pad)) != hipSuccess) return bail(e, "hipMalloc(.doc)");
  if ((e = hipMemsetAsync(s->d_doc + doc_len, 0, pad, c->stream)) != hipSuccess) return bail(e, "memset");
  if ((e = hipMemcpyAsync(s->d_doc, doc_file, doc_len, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return bail(e, "copy .doc");
  if (norms && max_doc > 0) {
    if ((e = hipMalloc(&s->d_norms, (size_t)max_doc + 64)) != hipSuccess) return bail(e, "hipMalloc(norms)");
    // <= 64 distinct norm bytes (the usual SmallFloat length spectrum): store ranks so that a clause's whole
    // (norm, freq) score table fits in LDS; otherwise keep the raw bytes
    size_t hist[256] = {0};
    for (int32_t d = 0; d < max_doc; ++d) hist[norms[d]]++;
    uint8_t rank_of[256] = {0}, rank_to_norm[64] = {0};
    int used = 0;
    for (int b = 0; b < 256; ++b) if (hist[b]) { if (used < 64) { rank_of[b] = (uint8_t)used; rank_to_norm[used] = (uint8_t)b; } ++used; }
    if (used <= 64 && !c->cfg.raw_norms) {
      std::vector<uint8_t> ranks((size_t)max_doc);
      for (int32_t d = 0; d < max_doc; ++d) ranks[(size_t)d] = rank_of[norms[d]];
      if ((e = hipMalloc(&s->d_rank_to_norm, 64)) != hipSuccess) return bail(e, "hipMalloc(rank_to_norm)");
      if ((e = hipMemcpy(s->d_norms, ranks.data(), (size_t)max_doc, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy norm ranks");
      if ((e = hipMemcpy(s->d_rank_to_norm, rank_to_norm, 64, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "copy rank_to_norm");
      s->n_norm_ranks = used;
    } else if ((e = hipMemcpyAsync(s->d_norms, norms, (size_t)max_doc, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return bail(e, "copy norms");
  }
  if (live_docs && max_doc > 0) {
    const size_t words = ((size_t)max_doc + 63) / 64;
    if ((e = hipMalloc(&s->d_live, words * 8)) != hipSuccess) return bail(e, "hipMalloc(live docs)");
    if ((e = hipMemcpyAsync(s->d_live, live_docs, words * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return bail(e, "copy live docs");
  }
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "sync");
  // A large segment's prepared-term store is given its room here rather than by the first call that prepares terms: the first
  // touch of a 100 M-doc segment's whole dictionary reserves ~0.8 GB of block store and seven directory arrays, and those
  // hipMallocs were measured at 0.3-0.5 ms on most boxes and 15-20 ms on others — inside what a caller sees as the latency of its
  // first decode. Best effort (a refused allocation is not an error: the arrays grow on demand as before), sized from the file:
  // the aligned copy of a block is at most 28 bytes longer than its framing (prepare_terms_attempt: ~1.1 x the file in all), a
  // directory slot per ~330 file bytes on the corpora seen; under a byte budget (prepared_budget_mib) the ceiling bounds it.
  if (doc_len >= ((size_t)64 << 20)) {
    size_t rows_bytes = doc_len + doc_len / 4;
    size_t slots = doc_len / 256;
    if (c->prepared_budget != 0) { rows_bytes = std::min(rows_bytes, c->prepared_budget); slots = std::min(slots, c->prepared_budget / 32); }
    const bool roomy = s->bstore.reserve(rows_bytes, 0, c->stream) == hipSuccess && s->dir_last.reserve(slots, 0, c->stream) == hipSuccess &&
                       s->dir_off.reserve(slots, 0, c->stream) == hipSuccess && s->dir_row.reserve(slots, 0, c->stream) == hipSuccess &&
                       s->dir_hdr.reserve(slots, 0, c->stream) == hipSuccess && s->dir_bmax.reserve(slots, 0, c->stream) == hipSuccess &&
                       s->dir_sum.reserve(slots / 64 + 2, 0, c->stream) == hipSuccess &&
                       (!s->has_positions || s->dir_pos.reserve(slots, 0, c->stream) == hipSuccess);
    if (!roomy) (void)hipGetLastError();  // (out of memory is not sticky)
  }
  *out_seg = s;
  return RGPU_OK;
}

extern "C" void rgpu_segment_free(rgpu_segment* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  { std::lock_guard<std::mutex> g(s->ctx->mu); (void)settle_pending(s->ctx); }  // (closures of deferred OR batches name the segment)
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->d_doc) (void)hipFree(s->d_doc);
  if (s->d_norms) (void)hipFree(s->d_norms);
  if (s->d_rank_to_norm) (void)hipFree(s->d_rank_to_norm);
  if (s->d_live) (void)hipFree(s->d_live);
  if (s->d_pos) (void)hipFree(s->d_pos);
  s->dir_last.release(); s->dir_off.release(); s->dir_row.release(); s->dir_hdr.release(); s->dir_bmax.release(); s->dir_sum.release(); s->dir_pos.release(); s->pnorm.release(); s->bstore.release(); s->prep_scratch.release(); s->sketch.release(); s->sketch_jobs.release();
  for (void* b : s->bitmap_allocs) (void)hipFree(b);
  s->ctx->bitmap_bytes -= std::min(s->ctx->bitmap_bytes, s->bitmap_bytes);
  if (s->empty_bitmap) (void)hipFree(s->empty_bitmap);
  delete s;
}

extern "C" int32_t rgpu_segment_version(const rgpu_segment* s) { return s ? s->version : RGPU_ERR_ILLEGAL_ARGUMENT; }

// HBM held for one segment, by part (used bytes, not capacity)
extern "C" int32_t rgpu_segment_get_footprint(rgpu_segment* seg, rgpu_segment_footprint* out) {
  if (!seg || !out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  std::memset(out, 0, sizeof *out);
  out->doc_file_bytes = (int64_t)seg->doc_len;
  out->norms_bytes = seg->d_norms ? (int64_t)seg->max_doc : 0;
  out->live_docs_bytes = seg->d_live ? (int64_t)(((size_t)seg->max_doc + 63) / 64 * 8) : 0;
  out->positions_file_bytes = (int64_t)seg->pos_len;
  out->directory_bytes = (int64_t)seg->dir_used * (4 + 4 + 4 + 2 + 8 + (seg->has_positions ? 8 : 0)) + (int64_t)seg->sketch_used * TERM_SKETCH_K * 2 +
                         (int64_t)(seg->dir_used ? seg->dir_used / 64 + 2 : 0) * 8;
  out->block_store_bytes = (int64_t)seg->bstore_used;
  out->posting_norms_bytes = seg->d_norms ? (int64_t)seg->pnorm_used : 0;
  out->prepared_terms = (int64_t)seg->prepared.size();
  out->doc_bitmap_bytes = (int64_t)seg->bitmap_bytes;
  out->doc_bitmap_terms = seg->bitmap_terms;
  out->doc_bitmap_refused = seg->bitmap_refused;
  return RGPU_OK;
}

extern "C" int32_t rgpu_segment_release_prepared_terms(rgpu_segment* seg) {
  if (!seg) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "seg is null");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  // (a deferred batch whose redo fails here: the store is released all the same — nothing may keep reading it — and the
  // batch's own status is what this call returns)
  const int32_t settled = settle_pending(c);
  const std::string settled_why = settled == RGPU_OK ? std::string() : g_last_error;
  HIP_TRY(hipDeviceSynchronize());  // batches in flight on any stream still read the directories
  for (auto& sc : c->scr) sc.busy = false;
  for (auto& cs : c->ceil_slots) cs.busy = false;
  seg->prepared.clear();
  seg->dir_used = seg->bstore_used = seg->pnorm_used = seg->sketch_used = 0;  // the arrays keep their capacity and are refilled from the start
  for (void* b : seg->bitmap_allocs) (void)hipFree(b);
  seg->bitmap_allocs.clear();
  seg->bitmaps.clear();
  seg->memb_only.clear();
  c->bitmap_bytes -= std::min(c->bitmap_bytes, seg->bitmap_bytes);
  seg->bitmap_bytes = 0;
  seg->bitmap_terms = seg->bitmap_refused = 0;
  if (settled != RGPU_OK) return fail(settled, "a deferred disjunction batch failed when its flags were looked at: " + settled_why);
  return RGPU_OK;
}

extern "C" int32_t rgpu_segment_prepare_terms(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms) {
  if (!seg || (!terms && n_terms > 0) || n_terms < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  std::vector<const rgpu_term_state*> ptrs((size_t)n_terms);
  for (int64_t i = 0; i < n_terms; ++i) ptrs[(size_t)i] = &terms[i];
  return prepare_terms_locked(seg, ptrs.data(), ptrs.size());
}

// ---- decode ----------------------------------------------------------------------------------------------------
static int32_t decode_terms_impl(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, int32_t* docs_dev,
                                 int32_t* freqs_dev, hipStream_t stream, int64_t* total_out) {
  rgpu_ctx* c = seg->ctx;
  std::vector<const rgpu_term_state*> ptrs((size_t)n_terms);
  std::vector<int64_t> out_of((size_t)n_terms + 1);
  int64_t out = 0;
  for (int64_t i = 0; i < n_terms; ++i) {
    ptrs[(size_t)i] = &terms[i];
    if (terms[i].doc_freq < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative doc_freq");
    out_of[(size_t)i] = out;
    out += terms[i].doc_freq;
  }
  out_of[(size_t)n_terms] = out;
  if (total_out) *total_out = out;
  // A decode needs no norms: stage A only. Terms met for the first time are unpacked by the preparation anyway (every block
  // is validated once): it writes their postings straight to the caller's arrays; k_decode_terms serves the rest.
  std::vector<uint8_t> fused;
  const DecodeSink sink{out_of.data(), docs_dev, freqs_dev, &fused};
  int32_t rc = prepare_terms_locked(seg, ptrs.data(), ptrs.size(), false, &sink);
  if (rc != RGPU_OK) return rc;
  std::vector<int64_t> rest;
  for (int64_t i = 0; i < n_terms; ++i)
    if (terms[i].doc_freq > 0 && !(i < (int64_t)fused.size() && fused[(size_t)i])) rest.push_back(i);
  if (rest.empty()) return RGPU_OK;
  const int64_t nr = (int64_t)rest.size();
  SCRATCH_TAKE(c);  // callers synchronize the stream before they return
  Stager st(c);
  const size_t o_terms = st.add((size_t)nr * sizeof(DevTerm));
  const size_t o_items = st.add((size_t)(nr + 1) * 8);
  const size_t o_out = st.add((size_t)(nr + 1) * 8);
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, c->stream));
  DevTerm* ht = reinterpret_cast<DevTerm*>(c->S->h_stage.p + o_terms);
  int64_t* hitems = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_items);
  int64_t* hout = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_out);
  int64_t items = 0, postings = 0;
#ifndef RGPU_DEC_BPI
#define RGPU_DEC_BPI 16
#endif
  const int dec_blocks_per_item = RGPU_DEC_BPI;
  for (int64_t j = 0; j < nr; ++j) {
    if (j + 16 < nr) seg->prepared.prefetch(terms[rest[(size_t)(j + 16)]].doc_start_fp);
    const int64_t i = rest[(size_t)j];
    rc = make_dev_term(seg, terms[i], 0.f, 0, &ht[j], false);
    if (rc != RGPU_OK) return rc;
    hitems[j] = items;
    hout[j] = out_of[(size_t)i];
    items += ht[j].nblocks == 0 ? 1 : (ht[j].nblocks + dec_blocks_per_item - 1) / dec_blocks_per_item;
    postings += terms[i].doc_freq;
  }
  hitems[nr] = items;
  hout[nr] = out;
  HIP_TRY(stage_h2d(c, st.used, stream));
  const unsigned grid = wg_count((items + WG_WAVES - 1) / WG_WAVES);
  {
    TimedLaunch tl(c, stream, "k_decode_terms", postings);
    auto args = [&](auto kern) {
      RGPU_LAUNCH(kern, dim3(grid), dim3(WG_THREADS), 0, stream, seg_view(seg),
                         reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_terms),
                         reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_items),
                         reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_out), (int)nr, items, dec_blocks_per_item, docs_dev,
                         freqs_dev);
    };
    if (seg->version >= 1) args(k_decode_terms<false>); else args(k_decode_terms<true>);
  }
  HIP_TRY(launch_status());
  return RGPU_OK;
}

extern "C" int32_t rgpu_decode_terms_device(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, void* docs_dev,
                                            void* freqs_dev, void* hip_stream) {
  if (!seg || !terms || n_terms <= 0 || !docs_dev || !freqs_dev) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : seg->ctx->stream;
  // the staging buffer is reused by the next call: finish this launch's H2D before returning control
  int32_t rc = decode_terms_impl(seg, terms, n_terms, (int32_t*)docs_dev, (int32_t*)freqs_dev, s, nullptr);
  if (rc != RGPU_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return RGPU_OK;
}

extern "C" int32_t rgpu_decode_terms(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, int32_t* docs_out,
                                     int32_t* freqs_out) {
  if (!seg || !terms || n_terms <= 0 || !docs_out || !freqs_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  int64_t total = 0;
  for (int64_t i = 0; i < n_terms; ++i) { if (terms[i].doc_freq < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative doc_freq"); total += terms[i].doc_freq; }
  if (total == 0) return RGPU_OK;
  int32_t *d_docs = nullptr, *d_freqs = nullptr;
  HIP_TRY(hipMalloc(&d_docs, (size_t)total * 4));
  if (hipMalloc(&d_freqs, (size_t)total * 4) != hipSuccess) { (void)hipFree(d_docs); return fail(RGPU_ERR_RUNTIME, "out of device memory"); }
  int32_t rc = decode_terms_impl(seg, terms, n_terms, d_docs, d_freqs, c->stream, nullptr);
  if (rc == RGPU_OK) {
    hipError_t e1 = hipMemcpyAsync(docs_out, d_docs, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipMemcpyAsync(freqs_out, d_freqs, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(RGPU_ERR_RUNTIME, "device to host copy failed");
  }
  (void)hipFree(d_docs);
  (void)hipFree(d_freqs);
  return rc;
}

extern "C" int32_t rgpu_advance_batch(rgpu_segment* seg, const rgpu_term_state* term, const int32_t* targets, int64_t n,
                                      int32_t* out_docs, int32_t* out_freqs) {
  if (!seg || !term || !targets || n <= 0 || !out_docs || !out_freqs) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  if (term->doc_freq <= 0) { for (int64_t i = 0; i < n; ++i) { out_docs[i] = RGPU_NO_MORE_DOCS; out_freqs[i] = 0; } return RGPU_OK; }
  const rgpu_term_state* p = term;
  int32_t rc = prepare_terms_locked(seg, &p, 1, false);
  if (rc != RGPU_OK) return rc;
  DevTerm T;
  rc = make_dev_term(seg, *term, 0.f, 0, &T, false);
  if (rc != RGPU_OK) return rc;
  int32_t* d = nullptr;
  HIP_TRY(hipMalloc(&d, (size_t)n * 12));
  hipError_t e = hipMemcpyAsync(d, targets, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    TimedLaunch tl(c, c->stream, "k_advance", n);
    const unsigned grid = wg_count((n + WG_WAVES - 1) / WG_WAVES);
    if (seg->version >= 1)
      RGPU_LAUNCH(k_advance<false>, dim3(grid), dim3(WG_THREADS), 0, c->stream, seg_view(seg), T, d, n, d + n, d + 2 * n);
    else
      RGPU_LAUNCH(k_advance<true>, dim3(grid), dim3(WG_THREADS), 0, c->stream, seg_view(seg), T, d, n, d + n, d + 2 * n);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_docs, d + n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_freqs, d + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RGPU_ERR_RUNTIME, hipGetErrorString(e));
  return RGPU_OK;
}


// ---- doc bitmaps (kernels/doc_bitmap.hpp) --------------------------------------------------------------------------
static int32_t decode_terms_impl(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, int32_t* docs_dev,
                                 int32_t* freqs_dev, hipStream_t stream, int64_t* total_out);
// `or_bitmaps`: a term holding at least one doc in this many gets a bitmap (0 = default 64, < 0: none)
static int64_t bitmap_min_df(const rgpu_segment* seg) {
  const int32_t d = seg->ctx->cfg.or_bitmaps;
  if (d < 0) return INT64_MAX;
  const int64_t den = d == 0 ? 64 : d;
  return std::max<int64_t>(1024, ((int64_t)seg->max_doc + den - 1) / den);
}
// ... and for the clauses of a conjunction (`and_bitmaps`: 0 = default 256)
static int64_t bitmap_min_df_and(const rgpu_segment* seg) {
  const int32_t d = seg->ctx->cfg.and_bitmaps;
  if (d < 0) return INT64_MAX;
  const int64_t den = d == 0 ? 256 : d;
  return std::max<int64_t>(1024, ((int64_t)seg->max_doc + den - 1) / den);
}
// builds the bitmaps of the given terms that lack one (ctx mutex held; ends synchronised). One decode of the list into scratch
// (the context's run buffer), one thread per posting, a prefix popcount: a one-off per term, ~1 ms for a 2 M-posting list.
static int32_t ensure_bitmaps_locked(rgpu_segment* seg, const rgpu_term_state* const* sts, const int32_t* sim_tables, size_t n) {
  rgpu_ctx* c = seg->ctx;
  const int64_t n_words = ((int64_t)seg->max_doc + 31) / 32;
  const size_t nw_pad = ((size_t)n_words + 1 + BITMAP_PAD_WORDS + 63) & ~size_t(63);
  // A bitmap is an accelerator, never a requirement: a term the budget has no room for, whose allocation fails or whose build
  // fails is filed as "no bitmap" (usable = false, nothing held) and its clause is walked — the search itself never fails here.
  auto refuse = [&](const rgpu_term_state& st) {
    BitmapInfo none{};
    none.df = st.doc_freq;
    none.usable = false;
    seg->bitmaps.put(st.doc_start_fp, none);
    seg->bitmap_refused++;
  };
  for (size_t i = 0; i < n; ++i) {
    const rgpu_term_state& st = *sts[i];
    if (st.doc_freq < 2 || seg->bitmaps.find(st.doc_start_fp)) continue;
    const size_t df = (size_t)st.doc_freq;
    const size_t o_ranks = nw_pad * 8, o_ovf = o_ranks + nw_pad * 4, o_stats = o_ovf + (size_t)BITMAP_OVF_CAP * 8;
    // (+ the plain membership bits, one per doc: what the conjunction kernel's batched first probe gathers from — a quarter of
    // the four-bits-per-doc array's footprint, half of the {any, hi} pairs')
    const size_t o_freqs = o_stats + 64, o_memb = o_freqs + ((df + 127) & ~size_t(63)), o_nib = o_memb + nw_pad * 4;
    // (the four-bits-per-doc array of the densest terms: conjunctions answer a candidate with ONE gather from it)
    bool with_nib = (int64_t)st.doc_freq * BITMAP_NIBBLE_DENSITY >= (int64_t)seg->max_doc;
    size_t total = o_nib + (with_nib ? (nw_pad * 4 + 64) * 4 : 0);
    if (with_nib && c->bitmap_bytes + total > c->bitmap_budget) { with_nib = false; total = o_nib; }  // the part a clause can do without goes first
    if (c->bitmap_bytes + total > c->bitmap_budget) { refuse(st); continue; }
    uint8_t* block = nullptr;
    if (hipMalloc(&block, total) != hipSuccess) {
      (void)hipGetLastError();  // out of memory is not sticky; the clause is walked
      refuse(st);
      continue;
    }
    BitmapInfo info{};
    info.words = reinterpret_cast<uint2*>(block);
    info.sim_table = sim_tables[i];
    info.ranks = reinterpret_cast<uint32_t*>(block + o_ranks);
    info.ovf = reinterpret_cast<uint32_t*>(block + o_ovf);
    info.freqs = block + o_freqs;
    info.nib = with_nib ? reinterpret_cast<uint32_t*>(block + o_nib) : nullptr;
    info.memb = reinterpret_cast<uint32_t*>(block + o_memb);
    info.df = st.doc_freq;
    BitmapStats* d_stats = reinterpret_cast<BitmapStats*>(block + o_stats);
    BitmapStats hs{};
    uint32_t listed = 0;
    // the build; the block is accounted for only once it succeeded (a failure frees it: no retry leaks a block)
    auto build = [&]() -> int32_t {
      HIP_TRY(hipMemsetAsync(block, 0, total, c->stream));
      static_assert(sizeof(ScoredPosting) == 8, "the run buffer doubles as {docs, freqs} scratch");
      HIP_TRY(c->d_runs.reserve(df + 64, 0, c->stream));
      int32_t* docs = reinterpret_cast<int32_t*>(c->d_runs.p);
      int32_t* freqs = docs + df;
      int32_t rc = decode_terms_impl(seg, &st, 1, docs, freqs, c->stream, nullptr);
      if (rc != RGPU_OK) return rc;
      {
        TimedLaunch tl(c, c->stream, "k_bitmap_build", (int64_t)df);
        RGPU_LAUNCH(k_bitmap_fill, dim3(wg_count((df + 255) / 256)), dim3(256), 0, c->stream, docs, freqs, (int64_t)df, seg->max_doc,
                           (const uint8_t*)seg->d_norms, (const float*)(c->sim_tables.p + (size_t)sim_tables[i] * 257),
                           seg->n_norm_ranks > 0 ? (const uint8_t*)seg->d_rank_to_norm : (const uint8_t*)nullptr, info.words, info.freqs, info.ovf, d_stats, info.nib, info.memb);
        const int64_t n_scan = n_words + 1;  // ranks[n_words] = the list's size
        RGPU_LAUNCH(k_bitmap_popc, dim3(wg_count((n_scan + 255) / 256)), dim3(256), 0, c->stream, info.words, n_scan, info.ranks);
        const int64_t n_tiles = (n_scan + SCAN_TILE - 1) / SCAN_TILE;
        HIP_TRY(seg->prep_scratch.reserve(64 + (size_t)n_tiles * 8 + 64, 0, c->stream));
        unsigned long long* d_total = reinterpret_cast<unsigned long long*>(seg->prep_scratch.p);
        unsigned long long* d_tiles = d_total + 8;
        RGPU_LAUNCH(k_scan_reduce, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, c->stream, info.ranks, n_scan, d_tiles);
        RGPU_LAUNCH(k_scan_tiles, dim3(1), dim3(PREP_THREADS), 0, c->stream, d_tiles, n_tiles, 0xfffffff0ull, d_total, c->d_err);
        RGPU_LAUNCH(k_scan_down, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, c->stream, info.ranks, n_scan, d_tiles);
      }
      HIP_TRY(hipMemcpyAsync(&hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(&listed, info.ranks + n_words, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(launch_status());
      return RGPU_OK;
    };
    const int32_t rc = build();
    if (rc != RGPU_OK) {
      (void)hipStreamSynchronize(c->stream);  // nothing may still write the block
      (void)hipFree(block);
      if (rc == RGPU_ERR_RUNTIME) { (void)hipGetLastError(); refuse(st); continue; }  // scratch allocation, launch: a walked clause serves
      return rc;  // the term itself is bad (corrupt postings): walking it would fail the same way — report it
    }
    seg->bitmap_allocs.push_back(block);
    seg->bitmap_bytes += total;
    c->bitmap_bytes += total;
    info.n_ovf = (int32_t)std::min<unsigned>(hs.n_ovf, (unsigned)BITMAP_OVF_CAP);
    info.max_freq = (int32_t)std::min<unsigned>(hs.max_freq, 0x7fffffffu);
    // (a list with a repeated or out-of-range doc id — a corrupt tail — or too many huge freqs simply stays a walked clause)
    info.usable = hs.n_ovf <= (unsigned)BITMAP_OVF_CAP && hs.bad_docs == 0 && listed == (uint32_t)st.doc_freq && hs.max_freq >= 1;
    if (!info.usable) {  // ... and holds no HBM for it
      (void)hipFree(block);
      seg->bitmap_allocs.pop_back();
      seg->bitmap_bytes -= total;
      c->bitmap_bytes -= total;
      info.words = nullptr; info.ranks = nullptr; info.ovf = nullptr; info.freqs = nullptr; info.nib = nullptr; info.memb = nullptr;
      seg->bitmap_refused++;
    } else {
      seg->bitmap_terms++;
    }
    seg->bitmaps.put(st.doc_start_fp, info);
  }
  return RGPU_OK;
}

// Membership bits for conjunction clauses too sparse for a full bitmap (ctx mutex held; ends synchronised): see k_bitmap_memb.
// Round 6 measured where k_search_and's cycles go on the 1024 x 3-term batch (-DRGPU_AND_TIME): the 6 % of the lead blocks whose
// first clause behind the lead had NO bitmap (a list below 1 doc in 256: walked through its block directory, ~6 block decodes per
// lead block) took 29 % of the kernel's wave cycles — 13 x what a lead block costs on the batched-probe path. The probe itself
// only needs one bit per doc: such a clause now gets the bits alone (max_doc / 8 bytes, no ranks / freqs / nibbles), every lead
// block takes the batched probe, and only the survivors (docs that ARE in the list) walk its directory for their freq.
constexpr int64_t MEMB_ONLY_MIN_DF = 512;   // shorter lists: the lead-driven walk meets at most four of their blocks anyway
static void ensure_memb_only_locked(rgpu_segment* seg, const rgpu_term_state* const* sts, size_t n) {
  rgpu_ctx* c = seg->ctx;
  const int64_t n_words = ((int64_t)seg->max_doc + 31) / 32;
  const size_t nw_pad = ((size_t)n_words + 1 + BITMAP_PAD_WORDS + 63) & ~size_t(63);
  const size_t total = nw_pad * 4 + 64;
  for (size_t i = 0; i < n; ++i) {
    const rgpu_term_state& st = *sts[i];
    if (seg->memb_only.find(st.doc_start_fp)) continue;
    auto refuse = [&]() { seg->memb_only.put(st.doc_start_fp, rgpu_segment::MembOnly{nullptr, st.doc_freq}); };
    if (c->bitmap_bytes + total > c->bitmap_budget) { refuse(); continue; }
    uint8_t* block = nullptr;
    if (hipMalloc(&block, total) != hipSuccess) { (void)hipGetLastError(); refuse(); continue; }
    BitmapStats hs{};
    auto build = [&]() -> bool {
      const size_t df = (size_t)st.doc_freq;
      if (hipMemsetAsync(block, 0, total, c->stream) != hipSuccess) return false;
      if (c->d_runs.reserve(df + 64, 0, c->stream) != hipSuccess) return false;
      int32_t* docs = reinterpret_cast<int32_t*>(c->d_runs.p);
      if (decode_terms_impl(seg, &st, 1, docs, docs + df, c->stream, nullptr) != RGPU_OK) return false;
      {
        TimedLaunch tl(c, c->stream, "k_bitmap_memb", (int64_t)df);
        RGPU_LAUNCH(k_bitmap_memb, dim3(wg_count((df + 255) / 256)), dim3(256), 0, c->stream, docs, (int64_t)df, seg->max_doc,
                           reinterpret_cast<uint32_t*>(block), reinterpret_cast<BitmapStats*>(block + nw_pad * 4));
      }
      if (hipMemcpyAsync(&hs, block + nw_pad * 4, sizeof hs, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
      if (hipStreamSynchronize(c->stream) != hipSuccess || launch_status() != hipSuccess) return false;
      return hs.bad_docs == 0;
    };
    if (!build()) {  // an accelerator, never a requirement: the clause is walked, and whatever is wrong with the list shows there
      (void)hipStreamSynchronize(c->stream);
      (void)hipGetLastError();
      (void)hipFree(block);
      refuse();
      continue;
    }
    seg->bitmap_allocs.push_back(block);
    seg->bitmap_bytes += total;
    c->bitmap_bytes += total;
    seg->memb_only.put(st.doc_start_fp, rgpu_segment::MembOnly{reinterpret_cast<uint32_t*>(block), st.doc_freq});
  }
}

// ---- search ----------------------------------------------------------------------------------------------------
namespace {
struct Group {  // queries of one op, in their original order
  int op = 0;
  bool or_wide = false;         // OR queries of >= 10 clauses: the order-free workgroup-window kernel
  bool no_lazy = false;         // ... that k_or_lazy already declined (no bitmap clause) or handed back
  bool after_lazy = false;      // ... following a k_or_lazy launch of the same call (rgpu_last_search_counters adds them up)
  std::vector<int64_t> term_bytes;  // or_wide: per clause, the encoded bytes of its postings (rgpu_last_search_counters)
  bool req_opt = false;         // MUST + SHOULD trees under the reference's ReqOptScorer rule: conjunction records + sequential scan
  std::vector<int32_t> qmap;    // original query index
  std::vector<DevQuery> queries;
  std::vector<DevTerm> terms;
  std::vector<int64_t> item_prefix;
  int64_t postings = 0;
};
}  // namespace

template <bool WIDE>
static void launch_merge(rgpu_ctx* c, hipStream_t s, int n_queries, int k, const int64_t* d_prefix, int32_t doc_base, HitOut* hits,
                         int64_t* totals, int head_items = 0, const int2* fixed_info = nullptr, int32_t* low_flags = nullptr,
                         const int32_t* qmap = nullptr) {
  TimedLaunch tl(c, s, "k_merge_items", 0);
  const unsigned grid = wg_count((n_queries + WG_WAVES - 1) / WG_WAVES);
  RGPU_LAUNCH(k_merge_items<WIDE>, dim3(grid), dim3(WG_THREADS), 0, s, d_prefix, n_queries, k, c->S->d_partial_keys.p,
                     c->S->d_partial_counts.p, doc_base, head_items, hits, totals, fixed_info, low_flags, qmap, c->pass.stride, c->pass.col0,
                     c->pass.ceil_out);
}

// OR: score every clause once into {doc, score} runs, then accumulate per doc-id window (kernels/search_or.hpp)
static int32_t search_or_group(rgpu_segment* seg, Group& G, int32_t k, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  const int nq = (int)G.queries.size();
  const int nt = (int)G.terms.size();
  const bool wide = k > 64;
  const bool legacy = seg->version < 1;
  if (nt == 0) return RGPU_OK;  // every clause absent from this leaf: rows keep their {-1, 0} / 0 defaults
  SCRATCH_TAKE(c);
  // Which clauses does the window kernel decode itself? Per query the (up to) `or_dense_clauses` longest SHOULD lists
  // whose 128-posting blocks span at most two windows on average (df * W >= 64 * max_doc): a block is unpacked again
  // in every window it reaches into, so a sparser list is cheaper through a run. On a Zipfian query those few lists
  // hold most of the postings; every other clause (and every tail) goes through a run. Needs the per-clause LDS score
  // table, i.e. norms held as ranks.
  const int W = std::min(4096, std::max(256, c->cfg.or_window_docs > 0 ? (c->cfg.or_window_docs + 255) / 256 * 256 : 1024));
  const int dense_max = c->cfg.or_dense_clauses < 0 ? 0 : (c->cfg.or_dense_clauses == 0 ? OR_DENSE_MAX : std::min(c->cfg.or_dense_clauses, OR_DENSE_MAX));
  if (dense_max > 0 && seg->d_norms && seg->n_norm_ranks > 0) {
    for (DevQuery& dq : G.queries) {
      uint32_t mask = 0;
      for (int pick = 0; pick < dense_max; ++pick) {
        int best = -1;
        for (int i = 0; i < std::min(dq.n_terms, 16); ++i) {  // (the mask travels in 16 bits of DevQuery::op)
          const DevTerm& t = G.terms[(size_t)(dq.first_term + i)];
          if (((mask >> i) & 1u) || t.nblocks < 1 || (int64_t)t.df * W < 64 * (int64_t)seg->max_doc) continue;
          if (best < 0 || t.df > G.terms[(size_t)(dq.first_term + best)].df) best = i;
        }
        if (best < 0) break;
        mask |= 1u << best;
        G.terms[(size_t)(dq.first_term + best)].flags |= TERM_FLAG_OR_DENSE;
      }
      dq.op = (dq.op & 0xffff) | (int32_t)(mask << 16);
    }
  }
  // phase 1 plan: items = (clause, chunk of blocks); a dense clause is one item (its tail)
  int blocks_per_item = 32;
  std::vector<int64_t> item_prefix((size_t)nt + 1), run_prefix((size_t)nt + 1);
  int64_t items1 = 0, postings = 0;
  while (true) {
    items1 = 0;
    for (int j = 0; j < nt; ++j) {
      item_prefix[(size_t)j] = items1;
      const DevTerm& t = G.terms[(size_t)j];
      const bool dense = (t.flags & TERM_FLAG_OR_DENSE) != 0u;
      items1 += (t.nblocks == 0 || dense) ? 1 : (t.nblocks + blocks_per_item - 1) / blocks_per_item;
    }
    item_prefix[(size_t)nt] = items1;
    if (items1 <= 1048576 || blocks_per_item >= (1 << 17)) break;
    blocks_per_item *= 2;
  }
  for (int j = 0; j < nt; ++j) {
    const DevTerm& t = G.terms[(size_t)j];
    run_prefix[(size_t)j] = postings;
    postings += (int64_t)((t.flags & TERM_FLAG_OR_DENSE) ? t.tail_n : t.df) + OR_RUN_PAD;
  }
  run_prefix[(size_t)nt] = postings;  // run lengths include the sentinel padding
  // phase 2 plan: items = (query, group of windows), one per wavefront; a workgroup's wavefronts share one query
  const int wpq = std::max(1, (seg->max_doc + W - 1) / W);
  const int wpi = (int)std::max<int64_t>(1, ((int64_t)nq * wpq + 131071) / 131072);
  const int ipq = ((wpq + wpi - 1) / wpi + OR_WAVES - 1) / OR_WAVES * OR_WAVES;
  const int64_t items2 = (int64_t)nq * ipq;
  std::vector<int64_t> merge_prefix((size_t)nq + 1);
  for (int q = 0; q <= nq; ++q) merge_prefix[(size_t)q] = (int64_t)q * ipq;

  Stager st(c);
  const size_t o_q = st.add((size_t)nq * sizeof(DevQuery));
  const size_t o_t = st.add((size_t)nt * sizeof(DevTerm));
  const size_t o_ip = st.add((size_t)(nt + 1) * 8);
  const size_t o_rp = st.add((size_t)(nt + 1) * 8);
  const size_t o_mp = st.add((size_t)(nq + 1) * 8);
  const size_t o_m = st.add((size_t)nq * 4);
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
  std::memcpy(c->S->h_stage.p + o_q, G.queries.data(), (size_t)nq * sizeof(DevQuery));
  std::memcpy(c->S->h_stage.p + o_t, G.terms.data(), (size_t)nt * sizeof(DevTerm));
  std::memcpy(c->S->h_stage.p + o_ip, item_prefix.data(), (size_t)(nt + 1) * 8);
  std::memcpy(c->S->h_stage.p + o_rp, run_prefix.data(), (size_t)(nt + 1) * 8);
  std::memcpy(c->S->h_stage.p + o_mp, merge_prefix.data(), (size_t)(nq + 1) * 8);
  std::memcpy(c->S->h_stage.p + o_m, G.qmap.data(), (size_t)nq * 4);
  HIP_TRY(stage_h2d(c, st.used, stream));
  // the scored runs: the slot's own buffer (the group then only marks its slot, like a TERM / AND group), or — a group whose runs
  // would pin gigabytes in every slot — the context's, with a stream sync at the end
  const bool own_runs = (size_t)postings + 64 <= RUNS_PER_SLOT_MAX;
  DevVec<ScoredPosting>& runs_buf = own_runs ? c->S->d_runs : c->d_runs;
  HIP_TRY(runs_buf.reserve((size_t)postings + 64, 0, stream));
  HIP_TRY(c->S->d_tau.reserve((size_t)nq, 0, stream));
  HIP_TRY(hipMemsetAsync(c->S->d_tau.p, 0, (size_t)nq * 8, stream));
  HIP_TRY(c->S->d_partial_keys.reserve((size_t)items2 * (size_t)k, 0, stream));
  HIP_TRY(c->S->d_partial_counts.reserve((size_t)items2, 0, stream));
  const DevQuery* dq = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
  const DevTerm* dt = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
  const int64_t* dip = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_ip);
  const int64_t* drp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_rp);
  const int64_t* dmp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_mp);
  const int32_t* dm = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_m);
  const SegView sv = seg_view(seg);
  {
    TimedLaunch tl(c, stream, "k_score_terms", G.postings);
    const unsigned grid = wg_count((items1 + WG_WAVES - 1) / WG_WAVES);
    if (legacy)
      RGPU_LAUNCH(k_score_terms<true>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, dt, dip, drp, nt, items1, blocks_per_item, runs_buf.p);
    else
      RGPU_LAUNCH(k_score_terms<false>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, dt, dip, drp, nt, items1, blocks_per_item, runs_buf.p);
  }
  {
    TimedLaunch tl(c, stream, "k_or_windows", G.postings);
    bool has_not = false, has_msm = false;
    for (const DevQuery& dq : G.queries) { has_not = has_not || dq.pad != 0; has_msm = has_msm || ((dq.op >> 8) & 0xff) > 1; }
    const size_t lds = or_lds_bytes(W, has_msm);
    const unsigned grid = (unsigned)(items2 / OR_WAVES);  // exact: items_per_query is a multiple of OR_WAVES
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_dynamic_lds_once(c, reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      RGPU_LAUNCH(kern, dim3(grid), dim3(OR_THREADS), lds, stream, sv, dq, dt, drp, runs_buf.p, nq, wpq, wpi, ipq, W, (int)k,
                         c->S->d_partial_keys.p, c->S->d_partial_counts.p, c->S->d_tau.p, c->pass.ceil_in, dm);
      return hipSuccess;
    };
    auto pick = [&](auto legacy_tag) -> hipError_t {
      constexpr bool LG = decltype(legacy_tag)::value;
      if (has_msm)  // min_should_match > 1 somewhere: the general instantiation (it also handles MUST_NOT clauses)
        return wide ? go(k_or_windows<LG, true, true, true>) : go(k_or_windows<LG, false, true, true>);
      if (has_not) return wide ? go(k_or_windows<LG, true, true, false>) : go(k_or_windows<LG, false, true, false>);
      return wide ? go(k_or_windows<LG, true, false, false>) : go(k_or_windows<LG, false, false, false>);
    };
    HIP_TRY(legacy ? pick(std::true_type{}) : pick(std::false_type{}));
  }
  if (wide) launch_merge<true>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, nullptr, nullptr, dm);
  else launch_merge<false>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, nullptr, nullptr, dm);
  HIP_TRY(launch_status());
  if (own_runs) HIP_TRY(scratch_mark(c, stream));  // no stream sync: the slot is waited for when it is taken again
  else HIP_TRY(hipStreamSynchronize(stream));      // the context's run buffer is the next group's too
  return RGPU_OK;
}


static int32_t search_or_wide_group(rgpu_segment* seg, Group& G, int32_t k, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream);

// OR with >= 10 SHOULD clauses of which at least one has a doc bitmap: k_or_lazy (kernels/search_or_lazy.hpp). The bitmap
// clauses are not walked at all; the others are decoded and scored once per distinct (term, weight) of the batch by
// k_score_terms into {doc, score} runs. Queries without a bitmap clause, and queries the kernel hands back, go through k_or_wide.
#ifdef RGPU_LZ_TIME
#include <chrono>
#define HOST_STAMP(t) const auto t = std::chrono::steady_clock::now()
#define HOST_US(a, b) (long long)std::chrono::duration_cast<std::chrono::microseconds>((b) - (a)).count()
#else
#define HOST_STAMP(t) do {} while (0)
#endif
static int32_t search_or_lazy_group(rgpu_segment* seg, Group& G, int32_t k, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  HOST_STAMP(h0);
  const bool wide = k > 64;
  const bool legacy = seg->version < 1;
  const int64_t min_df = bitmap_min_df(seg);
#ifndef RGPU_LZ_STEPS
#define RGPU_LZ_STEPS 8
#endif
  constexpr int STEPS = RGPU_LZ_STEPS;  // 2048-doc steps per window (4 or 8)
  constexpr int W = STEPS * LZ_STEP_DOCS;
  // (at least one cell per 32 docs of a window: between the passes the cells double as the window's candidate words)
  int C = c->cfg.or_lazy_cells > 0 ? std::min(4096, std::max(W / 32, (c->cfg.or_lazy_cells + 63) / 64 * 64)) : 512;
  while (lz_lds_bytes(W, C) > 160u * 1024u && C > W / 32) C -= 64;

  if (!seg->empty_bitmap) {
    const size_t bytes = ((((size_t)seg->max_doc + 31) / 32 + 1 + BITMAP_PAD_WORDS + 63) & ~size_t(63)) * 16 + 256;  // (as a four-bits-per-doc array too)
    HIP_TRY(hipMalloc(&seg->empty_bitmap, bytes));
    HIP_TRY(hipMemsetAsync(seg->empty_bitmap, 0, bytes, stream));
  }
  if (G.queries.size() > 2) {
    // Queries that stream the same bitmaps sit next to each other in the launch (workgroup b works on query b % n_queries, one
    // group of windows after the other): the group's queries are ordered by their two densest terms. Measured on the
    // 1024 x 10-term batch: k_or_lazy 3.45 ms against 3.66 in the caller's order (26.3 against 28.9 at 100 M docs). Giving
    // every XCD (b % 8) a contiguous eighth of that order instead — its L2 would hold one neighbourhood's bitmaps — costs
    // more than it gains: 4.0 ms, the heavy neighbourhoods make their XCD the launch's tail. Rows are found through qmap.
    const int nq0 = (int)G.queries.size();
    std::vector<std::pair<uint64_t, int>> keyed((size_t)nq0);
    for (int i = 0; i < nq0; ++i) {
      const DevQuery& q0 = G.queries[(size_t)i];
      int32_t d1 = 0, d2 = 0;
      uint64_t f1 = 0, f2 = 0;  // the two densest clauses' terms
      for (int j = 0; j < q0.n_terms; ++j) {
        const DevTerm& t = G.terms[(size_t)(q0.first_term + j)];
        if (t.df > d1) { d2 = d1; f2 = f1; d1 = t.df; f1 = t.start_fp; } else if (t.df > d2) { d2 = t.df; f2 = t.start_fp; }
      }
      keyed[(size_t)i] = {(f1 & 0xffffffffull) << 32 | (f2 & 0xffffffffull), i};
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return a.first < b.first; });
    std::vector<DevQuery> qs((size_t)nq0);
    std::vector<int32_t> qm((size_t)nq0);
    for (int i = 0; i < nq0; ++i) { qs[(size_t)i] = G.queries[(size_t)keyed[(size_t)i].second]; qm[(size_t)i] = G.qmap[(size_t)keyed[(size_t)i].second]; }
    G.queries.swap(qs);
    G.qmap.swap(qm);
  }
  std::vector<LazyQuery> lq;
  std::vector<LazyRun> run_of;       // per walked clause instance
  std::vector<DevTerm> uniq;         // the distinct (term, weight, similarity) among them: one run each
  std::vector<int64_t> uniq_bytes;
  std::vector<int32_t> uniq_of;      // walked clause instance -> uniq
  std::vector<LazyClause> lz;
  std::vector<int32_t> lq_of;        // lazy query -> index in G
  std::vector<int32_t> fixed_info;
  int64_t walked_postings = 0, touched_bytes = 0, all_postings = 0;
  if (c->lazy_uniq.empty()) c->lazy_uniq.assign(65536, rgpu_ctx::UniqSlot{-1, 0u, 0});
  if (++c->lazy_stamp == 0) { for (auto& u : c->lazy_uniq) u.stamp = 0; c->lazy_stamp = 1; }
  const uint32_t stamp = c->lazy_stamp;
  lq.reserve(G.queries.size()); lq_of.reserve(G.queries.size()); fixed_info.reserve(2 * G.queries.size());
  run_of.reserve(G.terms.size()); uniq_of.reserve(G.terms.size()); lz.reserve(G.terms.size() / 2);
  for (int q = 0; q < (int)G.queries.size(); ++q) {
    const DevQuery& wq = G.queries[(size_t)q];
    // the (up to LZ_MAX_LAZY) densest clauses that have a usable bitmap
    struct Cand { int i; int32_t df; const BitmapInfo* bm; };
    Cand cands[RGPU_MAX_QUERY_TERMS];
    int n_cands = 0;
    for (int i = 0; i < wq.n_terms; ++i) {
      const DevTerm& t = G.terms[(size_t)(wq.first_term + i)];
      if (t.df < min_df) continue;
      const BitmapInfo* bm = seg->bitmaps.find((int64_t)t.start_fp);
      if (bm && bm->usable && bm->df == t.df) cands[n_cands++] = Cand{i, t.df, bm};
    }
    // (no dense term: every clause is walked, next to one lazy clause over the segment's empty bitmap — it holds no doc, its
    // bounds are zero. k_or_wide would decode the same postings inside 16384-doc workgroup windows that these short lists
    // leave almost empty: 0.5 ms for the 40 such queries of a 1024-query batch)
    const bool none_dense = n_cands == 0;
    std::stable_sort(cands, cands + n_cands, [](const Cand& a, const Cand& b) { return a.df > b.df; });
    n_cands = std::min(n_cands, LZ_MAX_LAZY);
    uint32_t lazy_mask = 0;
    for (int j = 0; j < n_cands; ++j) lazy_mask |= 1u << cands[j].i;
    // the fixed-point exponent of k_or_wide: over ALL clauses
    double bound = 0.0;
    for (int i = 0; i < wq.n_terms; ++i) {
      const DevTerm& t = G.terms[(size_t)(wq.first_term + i)];
      bound += (double)t.weight * ((double)c->sim_k1[(size_t)t.sim_table] + 1.0) * 1.000001;
    }
    int e = 100;
    if (bound > 0.0) e = std::min(100, std::ilogb((2147483648.0 - 64.0) / bound));  // floor(log2(.))
    LazyQuery Q{};
    Q.first_run = (int32_t)run_of.size();
    Q.first_lazy = (int32_t)lz.size();
    Q.e = e;
    for (int i = 0; i < wq.n_terms; ++i) {
      const DevTerm& t = G.terms[(size_t)(wq.first_term + i)];
      all_postings += t.df;
      if ((lazy_mask >> i) & 1u) continue;
      int u = -1;
      rgpu_ctx::UniqSlot& slot = c->lazy_uniq[(size_t)(((uint64_t)t.start_fp * 0x9E3779B97F4A7C15ull) >> 48)];
      if (slot.stamp == stamp && slot.fp == (int64_t)t.start_fp && t.df > 1) {
        const DevTerm& o = uniq[(size_t)slot.idx];
        if (o.df == t.df && o.sim_table == t.sim_table && std::memcmp(&o.weight, &t.weight, 4) == 0) u = slot.idx;
      }
      if (u < 0) {  // (the same term under another boost or similarity, or a colliding term: its own run)
        u = (int)uniq.size();
        DevTerm ut = t;
        ut.flags &= ~TERM_FLAG_OR_DENSE;
        uniq.push_back(ut);
        uniq_bytes.push_back(G.term_bytes.empty() ? 2 * (int64_t)t.df : G.term_bytes[(size_t)(wq.first_term + i)]);
        if (t.df > 1) slot = rgpu_ctx::UniqSlot{(int64_t)t.start_fp, stamp, u};
      }
      uniq_of.push_back(u);
      run_of.push_back(LazyRun{0, t.df, 0});
    }
    Q.n_runs = (int32_t)run_of.size() - Q.first_run;
    LazyClause mine[LZ_MAX_LAZY];
    if (none_dense) {
      LazyClause L{};
      L.words = reinterpret_cast<const uint2*>(seg->empty_bitmap);
      L.ranks = reinterpret_cast<const uint32_t*>(seg->empty_bitmap);
      L.freqs = seg->empty_bitmap;
      L.ovf = reinterpret_cast<const uint32_t*>(seg->empty_bitmap);
      L.nib = reinterpret_cast<const uint32_t*>(seg->empty_bitmap);
      L.sim_table = G.terms[(size_t)wq.first_term].sim_table;
      mine[0] = L;
      n_cands = 1;
    }
    for (int j = 0; j < (none_dense ? 0 : n_cands); ++j) {
      const DevTerm& t = G.terms[(size_t)(wq.first_term + cands[j].i)];
      LazyClause L{};
      L.words = cands[j].bm->words; L.ranks = cands[j].bm->ranks; L.freqs = cands[j].bm->freqs; L.ovf = cands[j].bm->ovf;
      L.nib = cands[j].bm->nib;
      L.n_ovf = cands[j].bm->n_ovf;
      L.sim_table = t.sim_table;
      const double k1 = (double)c->sim_k1[(size_t)t.sim_table];
      L.wk = t.weight * (c->sim_k1[(size_t)t.sim_table] + 1.0f);
      // no posting of the list scores above wk * fmax / (fmax + the smallest cache entry) (the score grows with the freq and
      // falls with the cache value; only non-negative tables reach this kernel); + slack for the device's rcp and rounding
      const double fmax = (double)std::max(1, cands[j].bm->max_freq);
      const double cmin = std::max(0.0, (double)c->sim_cache_min[(size_t)t.sim_table]);
      const double ub = (double)t.weight * (k1 + 1.0) * (fmax + cmin > 0.0 ? fmax / (fmax + cmin) : 1.0) * 1.00001;
      const double ubf = std::ceil(std::ldexp(ub, e)) + 2.0;
      L.ub = (uint32_t)std::min(2147483647.0, std::max(1.0, ubf));
      // a posting outside the bitmap's `hi` half scores below wk * BITMAP_HI_CUT — under the table the bitmap was built with
      const double ub_lo = (double)t.weight * (k1 + 1.0) * (double)BITMAP_HI_CUT * 1.00002;
      L.ub_lo = cands[j].bm->sim_table == t.sim_table ? std::min(L.ub, (uint32_t)std::min(2147483647.0, std::ceil(std::ldexp(ub_lo, e)) + 2.0)) : L.ub;
      mine[j] = L;
      touched_bytes += (int64_t)(((int64_t)seg->max_doc + 7) / 8);
    }
    std::stable_sort(mine, mine + n_cands, [](const LazyClause& a, const LazyClause& b) { return a.ub > b.ub; });
    // the kernel reads "the h largest (ub - ub_lo)" off a prefix sum in this order: if the order does not hold (a clause whose
    // sketch does not apply next to ones where it does), the query runs without sketches
    for (int j = 1; j < n_cands; ++j)
      if (mine[j].ub - mine[j].ub_lo > mine[j - 1].ub - mine[j - 1].ub_lo) { for (int i = 0; i < n_cands; ++i) mine[i].ub_lo = mine[i].ub; break; }
    uint64_t ub_sum = 0, ub_lo_sum = 0;
    for (int j = 0; j < n_cands; ++j) { ub_sum += mine[j].ub; ub_lo_sum += mine[j].ub_lo; lz.push_back(mine[j]); }
    Q.n_lazy = n_cands;
    Q.ub_sum = (uint32_t)std::min<uint64_t>(ub_sum, 0x7fffffffull);
    Q.ub_lo_sum = (uint32_t)std::min<uint64_t>(ub_lo_sum, 0x7fffffffull);
    lq.push_back(Q);
    lq_of.push_back(q);
    fixed_info.push_back(e);
    fixed_info.push_back((int32_t)(wq.n_terms * (int)ORX_FLOOR_PER_CLAUSE));
  }
  const int nq = (int)lq.size();
  HOST_STAMP(h1);
  if (nq > 0) {
    SCRATCH_TAKE(c);
    // phase 1 plan (k_score_terms): items = (distinct walked term, chunk of blocks); runs end in OR_RUN_PAD sentinels
    const int nu = (int)uniq.size();
    int blocks_per_item = 32;
    std::vector<int64_t> item_prefix((size_t)nu + 1), run_prefix((size_t)nu + 1);
    int64_t items1 = 0, run_slots = 0;
    while (true) {
      items1 = 0;
      for (int j = 0; j < nu; ++j) {
        item_prefix[(size_t)j] = items1;
        items1 += uniq[(size_t)j].nblocks == 0 ? 1 : (uniq[(size_t)j].nblocks + blocks_per_item - 1) / blocks_per_item;
      }
      item_prefix[(size_t)nu] = items1;
      if (items1 <= 1048576 || blocks_per_item >= (1 << 17)) break;
      blocks_per_item *= 2;
    }
    for (int j = 0; j < nu; ++j) {
      run_prefix[(size_t)j] = run_slots;
      run_slots += (int64_t)uniq[(size_t)j].df + OR_RUN_PAD;
      walked_postings += uniq[(size_t)j].df;
      touched_bytes += uniq_bytes[(size_t)j] + uniq[(size_t)j].df;
    }
    run_prefix[(size_t)nu] = run_slots;
    for (size_t i = 0; i < run_of.size(); ++i) run_of[i].base = run_prefix[(size_t)uniq_of[i]];
    // (touched_bytes counts index bytes only: the scored runs are scratch, written once and read once at 16 B per walked posting)
    // phase 2 plan: items = (query, group of windows), one per wavefront
    const int wpq = std::max(1, (int)(((int64_t)seg->max_doc + W - 1) / W));
#ifndef RGPU_LZ_TARGET_WAVES
#define RGPU_LZ_TARGET_WAVES 65536  // (measured on the 1024 x 10-term batch: 16 k 4.15 ms, 32 k 3.95, 64 k 3.77, 128 k 3.88)
#endif
    int ipq = std::min(wpq, std::max(1, (RGPU_LZ_TARGET_WAVES + nq - 1) / nq));
    const int wpi = (wpq + ipq - 1) / ipq;
    ipq = ((wpq + wpi - 1) / wpi + LZ_WAVES - 1) / LZ_WAVES * LZ_WAVES;
    const int64_t lists = (int64_t)nq * ipq;
    std::vector<int64_t> merge_prefix((size_t)nq + 1);
    for (int q = 0; q <= nq; ++q) merge_prefix[(size_t)q] = (int64_t)q * ipq;
    std::vector<int32_t> qmap((size_t)nq);
    for (int q = 0; q < nq; ++q) qmap[(size_t)q] = G.qmap[(size_t)lq_of[(size_t)q]];
    Stager st(c);
    const size_t o_q = st.add((size_t)nq * sizeof(LazyQuery));
    const size_t o_r = st.add(std::max<size_t>(1, run_of.size()) * sizeof(LazyRun));
    const size_t o_t = st.add(std::max<size_t>(1, uniq.size()) * sizeof(DevTerm));
    const size_t o_ip = st.add((size_t)(nu + 1) * 8);
    const size_t o_rp = st.add((size_t)(nu + 1) * 8);
    const size_t o_l = st.add(lz.size() * sizeof(LazyClause));
    const size_t o_mp = st.add((size_t)(nq + 1) * 8);
    const size_t o_m = st.add((size_t)nq * 4);
    const size_t o_fi = st.add((size_t)nq * 8);
    const size_t o_fl = st.add((size_t)nq * 4);   // low flags (k_merge_items)
    const size_t o_bl = st.add((size_t)nq * 4);   // bail flags (k_or_lazy)
    const size_t o_ct = st.add(16 + (size_t)nq * 16);  // (+ per query, variant builds: evaluated docs, lazy-only docs)
    const size_t o_sn = st.add(64 * sizeof(ScoredPosting));
    HIP_TRY(c->S->h_stage.reserve(st.used));
    HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
    std::memset(c->S->h_stage.p + o_fl, 0, st.used - o_fl);
    std::memcpy(c->S->h_stage.p + o_q, lq.data(), (size_t)nq * sizeof(LazyQuery));
    if (!run_of.empty()) std::memcpy(c->S->h_stage.p + o_r, run_of.data(), run_of.size() * sizeof(LazyRun));
    if (!uniq.empty()) std::memcpy(c->S->h_stage.p + o_t, uniq.data(), uniq.size() * sizeof(DevTerm));
    std::memcpy(c->S->h_stage.p + o_ip, item_prefix.data(), (size_t)(nu + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_rp, run_prefix.data(), (size_t)(nu + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_l, lz.data(), lz.size() * sizeof(LazyClause));
    std::memcpy(c->S->h_stage.p + o_mp, merge_prefix.data(), (size_t)(nq + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_m, qmap.data(), (size_t)nq * 4);
    std::memcpy(c->S->h_stage.p + o_fi, fixed_info.data(), (size_t)nq * 8);
    for (int i = 0; i < 64; ++i) reinterpret_cast<ScoredPosting*>(c->S->h_stage.p + o_sn)[i] = ScoredPosting{0x7fffffff, 0.0f};
    HOST_STAMP(h2);
    HIP_TRY(stage_h2d(c, st.used, stream));
    const bool own_runs = (size_t)run_slots + 128 <= RUNS_PER_SLOT_MAX;
    DevVec<ScoredPosting>& runs_buf = own_runs ? c->S->d_runs : c->d_runs;
    HIP_TRY(runs_buf.reserve((size_t)run_slots + 128, 0, stream));
    // 64 sentinel entries behind the runs: what a clause slot without a clause looks at
    HIP_TRY(hipMemcpyAsync(runs_buf.p + run_slots, c->S->d_stage.p + o_sn, 64 * sizeof(ScoredPosting), hipMemcpyDeviceToDevice, stream));
    // per query: the shared threshold slot, then the histogram of finished totals (LZ_HIST u32 counters)
    HIP_TRY(c->S->d_tau.reserve((size_t)nq * (1 + LZ_HIST / 2), 0, stream));
    HIP_TRY(hipMemsetAsync(c->S->d_tau.p, 0, (size_t)nq * (1 + LZ_HIST / 2) * 8, stream));
    uint32_t* d_hist = reinterpret_cast<uint32_t*>(c->S->d_tau.p + nq);
    HIP_TRY(c->S->d_partial_keys.reserve((size_t)lists * (size_t)k, 0, stream));
    HIP_TRY(c->S->d_partial_counts.reserve((size_t)lists, 0, stream));
    const int64_t* dmp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_mp);
    const int32_t* dm = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_m);
    int32_t* dbl = reinterpret_cast<int32_t*>(c->S->d_stage.p + o_bl);
    unsigned long long* dct = reinterpret_cast<unsigned long long*>(c->S->d_stage.p + o_ct);
    const SegView sv = seg_view(seg);
    c->last_counted = nullptr;
    c->last_counted_op = RGPU_OP_OR;
    c->last_counted_queries = nq;
    c->last_counted_postings = all_postings;
    c->last_counted_or_decoded = walked_postings;
    c->last_counted_or_bytes = touched_bytes;
    if (nu > 0) {
      TimedLaunch tl(c, stream, "k_score_terms", walked_postings);
      const unsigned grid = wg_count((items1 + WG_WAVES - 1) / WG_WAVES);
      const DevTerm* dt = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
      const int64_t* dip = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_ip);
      const int64_t* drp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_rp);
      if (legacy)
        RGPU_LAUNCH(k_score_terms<true>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, dt, dip, drp, nu, items1, blocks_per_item, runs_buf.p);
      else
        RGPU_LAUNCH(k_score_terms<false>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, dt, dip, drp, nu, items1, blocks_per_item, runs_buf.p);
    }
    {
      const size_t lds = lz_lds_bytes(W, C);
      // items [g0, g0 + n_g) of every query (n_g a multiple of LZ_WAVES)
      auto launch_part = [&](int g0, int n_g) -> hipError_t {
        TimedLaunch tl(c, stream, "k_or_lazy", all_postings);
        const unsigned grid = (unsigned)((int64_t)nq * n_g / LZ_WAVES);
        auto go = [&](auto kern) -> hipError_t {
          hipError_t e = set_dynamic_lds_once(c, reinterpret_cast<const void*>(kern), lds);
          if (e != hipSuccess) return e;
          RGPU_LAUNCH(kern, dim3(grid), dim3(LZ_THREADS), lds, stream, sv, reinterpret_cast<const LazyQuery*>(c->S->d_stage.p + o_q),
                             reinterpret_cast<const LazyRun*>(c->S->d_stage.p + o_r), runs_buf.p, reinterpret_cast<const LazyClause*>(c->S->d_stage.p + o_l),
                             (int64_t)run_slots, nq, wpq, wpi, ipq, g0, C, (int)k, c->S->d_partial_keys.p, c->S->d_partial_counts.p, c->S->d_tau.p, dbl, dct, d_hist);
          return hipSuccess;
        };
        return wide ? go(k_or_lazy<true, STEPS>) : go(k_or_lazy<false, STEPS>);
      };
      // A small batch puts every window of a query into a wavefront of its own (wpi == 1) and the launch starts them all at
      // once: nobody has a threshold, every doc held by a lazy list is a candidate — a single 10-clause query over 10 M docs
      // evaluated 1.8 M docs that way (2.3 ms; the same query inside a 1024-query batch: 3.4 us of the batch's kernel). Such a
      // batch runs as a PILOT launch over the first sixteenth of each query's windows, whose finished totals fill the query's
      // histogram, and a second launch over the rest that starts from that threshold. The threshold is a lower bound taken from
      // real docs either way: same rows, fewer evaluations.
      int pilot = 0;
      if (wpi == 1 && ipq >= 16 * LZ_WAVES) pilot = ((ipq + 15) / 16 + LZ_WAVES - 1) / LZ_WAVES * LZ_WAVES;
      if (pilot > 0) {
        HIP_TRY(launch_part(0, pilot));
        HIP_TRY(launch_part(pilot, ipq - pilot));
      } else {
        HIP_TRY(launch_part(0, ipq));
      }
    }
    const int2* dfi = reinterpret_cast<const int2*>(c->S->d_stage.p + o_fi);
    int32_t* dfl = reinterpret_cast<int32_t*>(c->S->d_stage.p + o_fl);
    // many lists per query, few queries: fold them sixteen at a time first. (Not for the usual large batch — 64 lists per query:
    // measured there, k_premerge_items 0.078 + k_merge_items 0.055 ms against 0.109 for k_merge_items alone)
    if (ipq >= 256 && (int64_t)nq * ipq <= 262144) {
      constexpr int GROUP = 16;
      const int gpq = (ipq + GROUP - 1) / GROUP;
      TimedLaunch tl(c, stream, "k_premerge_items", 0);
      const unsigned grid = wg_count(((int64_t)nq * gpq + WG_WAVES - 1) / WG_WAVES);
      if (wide) RGPU_LAUNCH(k_premerge_items<true>, dim3(grid), dim3(WG_THREADS), 0, stream, dmp, nq, (int)k, GROUP, gpq, c->S->d_partial_keys.p, c->S->d_partial_counts.p);
      else RGPU_LAUNCH(k_premerge_items<false>, dim3(grid), dim3(WG_THREADS), 0, stream, dmp, nq, (int)k, GROUP, gpq, c->S->d_partial_keys.p, c->S->d_partial_counts.p);
    }
    if (wide) launch_merge<true>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, dfi, dfl, dm);
    else launch_merge<false>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, dfi, dfl, dm);
    HIP_TRY(launch_status());
    // ---- the flags the launch set hands back: [counts: 2 x u64][low: nq x i32][bailed: nq x i32] into the slot's pinned buffer
    HIP_TRY(c->S->h_back.reserve(16 + (size_t)nq * 8));
    unsigned long long* h_counts = reinterpret_cast<unsigned long long*>(c->S->h_back.p);
    int32_t* h_low = reinterpret_cast<int32_t*>(c->S->h_back.p + 16);
    int32_t* h_bailed = h_low + nq;
    HOST_STAMP(h3);
    HIP_TRY(hipMemcpyAsync(h_low, dfl, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_bailed, dbl, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_counts, dct, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(scratch_mark(c, stream));
    Scratch* const slot = c->S;
    // What is left once the launch set has finished: queries whose window did not fit go through k_or_wide (their rows are simply
    // written again), queries whose top-k reaches below the fixed-point floor through the clause-order kernels, as after k_or_wide.
    // Everything it needs is held by value: it may run long after this function returned.
    auto shared_g = std::make_shared<Group>(std::move(G));
    auto shared_lq_of = std::make_shared<std::vector<int32_t>>(std::move(lq_of));
    auto finish = [c, seg, slot, shared_g, shared_lq_of, nq, k, hits_dev, totals_dev, stream, h_counts, h_low, h_bailed, own_runs, pass_at_enqueue = c->pass]() -> int32_t {
      PassScope scope(c, pass_at_enqueue);
      slot->settles_later = false;
      HIP_TRY(hipSetDevice(c->device));
      if (own_runs) HIP_TRY(hipEventSynchronize(slot->done)); else HIP_TRY(hipStreamSynchronize(stream));
      const Group& G0 = *shared_g;
      const std::vector<int32_t>& lq0 = *shared_lq_of;
      c->or_lazy_evals = (int64_t)h_counts[0];
      c->or_lazy_only = (int64_t)h_counts[1];
      c->stats[(size_t)stat_slot(c, "or_lazy_evaluated_docs")].launches += (int64_t)h_counts[0];
      c->stats[(size_t)stat_slot(c, "or_lazy_only_docs")].launches += (int64_t)h_counts[1];
      Group redo, rest;
      redo.op = RGPU_OP_OR;
      rest.op = RGPU_OP_OR;
      rest.or_wide = true;
      rest.no_lazy = true;
      rest.after_lazy = true;
      int64_t n_bailed = 0;
      for (int q = 0; q < nq; ++q) {
        const DevQuery& wq = G0.queries[(size_t)lq0[(size_t)q]];
        if (h_bailed[q]) {
          DevQuery dq = wq;
          dq.first_term = (int32_t)rest.terms.size();
          for (int i = 0; i < wq.n_terms; ++i) {
            rest.terms.push_back(G0.terms[(size_t)(wq.first_term + i)]);
            if (!G0.term_bytes.empty()) rest.term_bytes.push_back(G0.term_bytes[(size_t)(wq.first_term + i)]);
            rest.postings += G0.terms[(size_t)(wq.first_term + i)].df;
          }
          rest.qmap.push_back(G0.qmap[(size_t)lq0[(size_t)q]]);
          rest.queries.push_back(dq);
          ++n_bailed;
          continue;
        }
        if (!h_low[q]) continue;
        DevQuery dq2;
        dq2.op = RGPU_OP_OR;
        dq2.n_terms = wq.n_terms;
        dq2.first_term = (int32_t)redo.terms.size();
        dq2.pad = 0;
        for (int i = 0; i < wq.n_terms; ++i) {
          DevTerm t = G0.terms[(size_t)(wq.first_term + i)];
          t.flags &= ~TERM_FLAG_OR_DENSE;
          redo.terms.push_back(t);
          redo.postings += t.df;
        }
        redo.qmap.push_back(G0.qmap[(size_t)lq0[(size_t)q]]);
        redo.queries.push_back(dq2);
      }
      if (n_bailed) c->stats[(size_t)stat_slot(c, "or_lazy_bail_queries")].launches += n_bailed;
      int32_t rc = RGPU_OK;
      if (!redo.queries.empty()) {
        c->or_wide_redone += (int64_t)redo.queries.size();
        c->stats[(size_t)stat_slot(c, "or_wide_redo_queries")].launches += (int64_t)redo.queries.size();
        rc = search_or_group(seg, redo, k, hits_dev, totals_dev, stream);
      }
      if (rc == RGPU_OK && !rest.queries.empty()) rc = search_or_wide_group(seg, rest, k, hits_dev, totals_dev, stream);
      if (rc == RGPU_OK && (!redo.queries.empty() || !rest.queries.empty())) HIP_TRY(hipStreamSynchronize(stream));  // (the rare path: the rows are final on return)
      return rc;
    };
    if (c->defer_or && own_runs) {
      slot->settles_later = true;
      c->pending_or.push_back(std::move(finish));
      return RGPU_OK;
    }
#ifdef RGPU_LZ_TIME
    {
      HIP_TRY(hipStreamSynchronize(stream));
      HOST_STAMP(h4);
      std::fprintf(stderr, "[lz host] partition %lld us, plan + stage %lld us, enqueue %lld us, copies + sync %lld us\n", HOST_US(h0, h1), HOST_US(h1, h2), HOST_US(h2, h3), HOST_US(h3, h4));
      // developer output: how the evaluated docs spread over the queries
      std::vector<unsigned long long> pq((size_t)nq * 2);
      HIP_TRY(hipMemcpy(pq.data(), dct + 2, (size_t)nq * 16, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> ev((size_t)nq), lo((size_t)nq);
      for (int q = 0; q < nq; ++q) { ev[(size_t)q] = pq[(size_t)2 * q]; lo[(size_t)q] = pq[(size_t)2 * q + 1]; }
      std::sort(ev.begin(), ev.end());
      std::sort(lo.begin(), lo.end());
      auto at = [&](const std::vector<unsigned long long>& v, double f) { return v[(size_t)std::min<double>(v.size() - 1, f * v.size())]; };
      unsigned long long top5 = 0, top5lo = 0;
      for (size_t i = (size_t)(0.95 * nq); i < (size_t)nq; ++i) { top5 += ev[i]; top5lo += lo[i]; }
      std::fprintf(stderr, "[lz] per query evaluated docs: p50 %llu p90 %llu p95 %llu p99 %llu max %llu; the top 5%% hold %llu | lazy-only: p50 %llu p90 %llu p99 %llu max %llu, top 5%% %llu\n",
                   at(ev, .5), at(ev, .9), at(ev, .95), at(ev, .99), ev.back(), top5, at(lo, .5), at(lo, .9), at(lo, .99), lo.back(), top5lo);
    }
#endif
    return finish();
  }
  return RGPU_OK;
}

// OR with >= 10 SHOULD clauses (the reference sums those in heap order: any order is within its own spec): one launch of
// k_or_wide (kernels/search_or_wide.hpp), nothing materialised in HBM
static int32_t search_or_wide_group(rgpu_segment* seg, Group& G, int32_t k, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  const int nq = (int)G.queries.size();
  const int nt = (int)G.terms.size();
  const bool wide = k > 64;
  const bool legacy = seg->version < 1;
  if (nt == 0) return RGPU_OK;
  if (!G.no_lazy && c->cfg.or_bitmaps >= 0 && seg->bitmaps.size() > 0) return search_or_lazy_group(seg, G, k, hits_dev, totals_dev, stream);
  SCRATCH_TAKE(c);
  // a window: a multiple of the workgroup's scan step, small enough for the directory look-ahead and for LDS
  constexpr int WS_MAX = ORX_MAX_WINDOW / ORX_SCAN_STEP * ORX_SCAN_STEP;
#ifndef RGPU_ORX_WS  // the default window (variant builds sweep it together with RGPU_ORX_TABLES / RGPU_ORX_LOOK: LDS is the budget)
#define RGPU_ORX_WS 16384
#endif
#ifndef RGPU_ORX_WS16  // ... and of the one-workgroup-per-CU build (RGPU_ORX_WAVES=16)
#define RGPU_ORX_WS16 24576
#endif
  constexpr int WS_DEFAULT = (ORX_WAVES >= 16 ? RGPU_ORX_WS16 : RGPU_ORX_WS + ORX_SCAN_STEP - 1) / ORX_SCAN_STEP * ORX_SCAN_STEP;
  static_assert(WS_DEFAULT <= WS_MAX, "a window's blocks of one clause fit the directory look-ahead");
  int WS = c->cfg.or_wide_window_docs > 0 ? (c->cfg.or_wide_window_docs + ORX_SCAN_STEP - 1) / ORX_SCAN_STEP * ORX_SCAN_STEP : WS_DEFAULT;
  WS = std::min(WS_MAX, std::max(ORX_SCAN_STEP, WS));
  while (orx_lds_bytes(WS) > 160u * 1024u && WS > ORX_SCAN_STEP) WS -= ORX_SCAN_STEP;
  // score tables for the (up to) ORX_TABLES longest lists of each query
  for (DevQuery& dq : G.queries) {
    uint32_t mask = 0;
    for (int pick = 0; pick < ORX_TABLES; ++pick) {
      int best = -1;
      for (int i = 0; i < dq.n_terms; ++i) {
        const DevTerm& t = G.terms[(size_t)(dq.first_term + i)];
        if (((mask >> i) & 1u) || t.nblocks < 1) continue;
        if (best < 0 || t.df > G.terms[(size_t)(dq.first_term + best)].df) best = i;
      }
      if (best < 0) break;
      mask |= 1u << best;
    }
    dq.op = (dq.op & 0xffff) | (int32_t)(mask << 16);
  }
  // fixed point: a posting's score is at most weight * (k1 + 1) (times f32 rounding): 2^e times the sum of those bounds
  // stays below 2^31 - 64; a returned total below n * ORX_FLOOR_PER_CLAUSE steps sends the query through the f32 kernel
  std::vector<int32_t> fixed_info((size_t)nq * 2);
  for (int q = 0; q < nq; ++q) {
    DevQuery& dq = G.queries[(size_t)q];
    double bound = 0.0;
    for (int i = 0; i < dq.n_terms; ++i) {
      const DevTerm& t = G.terms[(size_t)(dq.first_term + i)];
      bound += (double)t.weight * ((double)c->sim_k1[(size_t)t.sim_table] + 1.0) * 1.000001;
    }
    int e = 100;
    if (bound > 0.0) e = std::min(100, (int)std::floor(std::log2((2147483648.0 - 64.0) / bound)));
    dq.pad = e;
    fixed_info[(size_t)q * 2] = e;
    fixed_info[(size_t)q * 2 + 1] = (int32_t)(dq.n_terms * (int)ORX_FLOOR_PER_CLAUSE);
  }
  // items = (query, group of windows), one per workgroup: enough of them to fill the chip a few times over
  const int wpq = std::max(1, (int)(((int64_t)seg->max_doc + WS - 1) / WS));
#ifndef RGPU_ORX_TARGET_WGS
#define RGPU_ORX_TARGET_WGS 8192
#endif
  int ipq = std::min(wpq, std::max(1, (RGPU_ORX_TARGET_WGS + nq - 1) / nq));
  const int wpi = (wpq + ipq - 1) / ipq;
  ipq = (wpq + wpi - 1) / wpi;
  const int64_t lists = (int64_t)nq * ipq * ORX_WAVES;  // one top-k list per wavefront
  std::vector<int64_t> merge_prefix((size_t)nq + 1);
  for (int q = 0; q <= nq; ++q) merge_prefix[(size_t)q] = (int64_t)q * ipq * ORX_WAVES;

  Stager st(c);
  const size_t o_q = st.add((size_t)nq * sizeof(DevQuery));
  const size_t o_t = st.add((size_t)nt * sizeof(DevTerm));
  const size_t o_mp = st.add((size_t)(nq + 1) * 8);
  const size_t o_m = st.add((size_t)nq * 4);
  const size_t o_fi = st.add((size_t)nq * 8);
  const size_t o_fl = st.add((size_t)nq * 4);  // written by k_merge_items, read back below
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
  std::memcpy(c->S->h_stage.p + o_q, G.queries.data(), (size_t)nq * sizeof(DevQuery));
  std::memcpy(c->S->h_stage.p + o_t, G.terms.data(), (size_t)nt * sizeof(DevTerm));
  std::memcpy(c->S->h_stage.p + o_mp, merge_prefix.data(), (size_t)(nq + 1) * 8);
  std::memcpy(c->S->h_stage.p + o_m, G.qmap.data(), (size_t)nq * 4);
  std::memcpy(c->S->h_stage.p + o_fi, fixed_info.data(), (size_t)nq * 8);
  std::memset(c->S->h_stage.p + o_fl, 0, (size_t)nq * 4);
  HIP_TRY(stage_h2d(c, st.used, stream));
  HIP_TRY(c->S->d_tau.reserve((size_t)nq, 0, stream));
  HIP_TRY(hipMemsetAsync(c->S->d_tau.p, 0, (size_t)nq * 8, stream));
  HIP_TRY(c->S->d_partial_keys.reserve((size_t)lists * (size_t)k, 0, stream));
  HIP_TRY(c->S->d_partial_counts.reserve((size_t)lists, 0, stream));
  const DevQuery* dq = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
  const DevTerm* dt = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
  const int64_t* dmp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_mp);
  const int32_t* dm = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_m);
  const SegView sv = seg_view(seg);
  c->last_counted = nullptr;  // nothing to read back: the wide kernel decodes every posting of every clause
  c->last_counted_op = RGPU_OP_OR;
  if (G.after_lazy) {  // the rest of a batch whose other queries k_or_lazy took: one set of counters for the call
    c->last_counted_queries += nq;
    c->last_counted_postings += G.postings;
    c->last_counted_or_decoded += G.postings;
    int64_t bytes = G.postings;  // 1 norm byte per posting + the encoded bytes
    for (int64_t b : G.term_bytes) bytes += b;
    c->last_counted_or_bytes += bytes;
  } else {
    c->last_counted_queries = nq;
    c->last_counted_postings = G.postings;
    c->last_counted_or_decoded = -1;
  }
  {
    TimedLaunch tl(c, stream, "k_or_wide", G.postings);
    const size_t lds = orx_lds_bytes(WS);
    const unsigned grid = wg_count((int64_t)nq * ipq);
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_dynamic_lds_once(c, reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      RGPU_LAUNCH(kern, dim3(grid), dim3(ORX_THREADS), lds, stream, sv, dq, dt, nq, wpq, wpi, ipq, WS, (int)k,
                         c->S->d_partial_keys.p, c->S->d_partial_counts.p, c->S->d_tau.p);
      return hipSuccess;
    };
    if (legacy) HIP_TRY(wide ? go(k_or_wide<true, true>) : go(k_or_wide<true, false>));
    else HIP_TRY(wide ? go(k_or_wide<false, true>) : go(k_or_wide<false, false>));
  }
  const int2* dfi = reinterpret_cast<const int2*>(c->S->d_stage.p + o_fi);
  int32_t* dfl = reinterpret_cast<int32_t*>(c->S->d_stage.p + o_fl);
  if (wide) launch_merge<true>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, dfi, dfl, dm);
  else launch_merge<false>(c, stream, nq, k, dmp, seg->doc_base, hits_dev, totals_dev, 0, dfi, dfl, dm);
  HIP_TRY(launch_status());
  HIP_TRY(c->S->h_back.reserve((size_t)nq * 4));
  int32_t* h_low = reinterpret_cast<int32_t*>(c->S->h_back.p);
  HIP_TRY(hipMemcpyAsync(h_low, dfl, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(scratch_mark(c, stream));
  Scratch* const slot = c->S;
  // queries whose top-k reaches below the fixed-point floor: once more, summed in f32 in clause order (their rows are
  // simply written again) — once the launch set has finished: now, or when its flags are looked at (rgpu_ctx::pending_or)
  auto shared_g = std::make_shared<Group>(std::move(G));
  auto finish = [c, seg, slot, shared_g, nq, k, hits_dev, totals_dev, stream, h_low, pass_at_enqueue = c->pass]() -> int32_t {
    PassScope scope(c, pass_at_enqueue);
    slot->settles_later = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(slot->done));
    const Group& G0 = *shared_g;
    Group redo;
    redo.op = RGPU_OP_OR;
    for (int q = 0; q < nq; ++q) {
      if (!h_low[q]) continue;
      // a fresh DevQuery (a wide query has no MUST_NOT clauses and min_should_match <= 1: op OR, clause count, nothing else —
      // the wide kernel's table mask and fixed-point exponent must not leak into the clause-order kernel's fields)
      const DevQuery& wq = G0.queries[(size_t)q];
      DevQuery dq2;
      dq2.op = RGPU_OP_OR;
      dq2.n_terms = wq.n_terms;
      dq2.first_term = (int32_t)redo.terms.size();
      dq2.pad = 0;
      const int first = wq.first_term;
      for (int i = 0; i < dq2.n_terms; ++i) {
        DevTerm t = G0.terms[(size_t)(first + i)];
        t.flags &= ~TERM_FLAG_OR_DENSE;
        redo.terms.push_back(t);
        redo.postings += t.df;
      }
      redo.qmap.push_back(G0.qmap[(size_t)q]);
      redo.queries.push_back(dq2);
    }
    if (redo.queries.empty()) return RGPU_OK;
    c->or_wide_redone += (int64_t)redo.queries.size();
    c->stats[(size_t)stat_slot(c, "or_wide_redo_queries")].launches += (int64_t)redo.queries.size();  // read by tests through rgpu_kernel_stats
    const int32_t rc = search_or_group(seg, redo, k, hits_dev, totals_dev, stream);
    if (rc != RGPU_OK) return rc;
    HIP_TRY(hipStreamSynchronize(stream));  // (the rare path: the rows are final on return)
    return RGPU_OK;
  };
  if (c->defer_or) {
    slot->settles_later = true;
    c->pending_or.push_back(std::move(finish));
    return RGPU_OK;
  }
  return finish();
}

// Lead blocks per work item of k_search_and when the caller leaves rgpu_config.and_blocks_per_item at 0: about 32 k items per
// launch. An item's fixed cost (descriptors, directory window, a partial top-k list for the merge) was a sixth of the kernel at 8
// blocks per item; much longer items leave the chip's 4096 wavefront slots with a ragged tail on a small launch.
#ifndef RGPU_AND_TARGET_ITEMS
#define RGPU_AND_TARGET_ITEMS 28672
#endif
#ifndef RGPU_AND_MAX_ITEM_BLOCKS  // (100 M docs, 3.2 M lead blocks: 2.76 ms at 32 blocks per item, 2.54 at 48, 2.35 at 60, 2.28 at 78, 2.59 at 112)
#define RGPU_AND_MAX_ITEM_BLOCKS 72
#endif
// k_search_and deals chunks of `chunk` consecutive workgroups to the XCDs (search_and.hpp) when a list's membership bits fit an
// XCD's L2 with room to spare; the grid then is whole rounds of 8 chunks
static int and_xcd_chunk(const rgpu_segment* seg) {
#ifdef RGPU_AND_XCD_CHUNK
  return RGPU_AND_XCD_CHUNK;
#else
  return (int64_t)seg->max_doc / 8 <= (2ll << 20) ? 64 * 4 / AND_WG_WAVES : 0;  // 2 MB of bits per list (max_doc <= 16.7 M) against 4 MB of L2 per XCD; 256 items per chunk
#endif
}
static unsigned long long and_grid(long long wgs, int chunk) {
  if (chunk <= 0) return (unsigned long long)wgs;
  const long long round = 8ll * chunk;
  return (unsigned long long)((wgs + round - 1) / round * round);
}
// Blocks per work item of k_search_term when the caller leaves rgpu_config.blocks_per_item at 0 — two rules, measured together (round 6,
// scripts/split_sweep.sh, k_search_term + k_merge_items on the headline batch, one box per table):
//   * the launch's size (here): with the chunk frontiers (SegView::dir_sum) in front of the per-block test, an item's cost is its
//     set-up (term descriptor, score table, sketch threshold: ~5 us) plus 1.4 us per block that can still enter, whatever its length;
//     a long list is mostly pruned, so its items are long: the smallest power of two that leaves at most `target_items` (3000) such items,
//     at most 4096 blocks (an item's chunks of 64 blocks are tested one lane each);
//   * a query's own size (term_query_item_blocks below): at least `split` (8) items for every list of 512 blocks or more, fewer down to
//     64 blocks per item.
//   10 M docs:  target 12000 / split 1: 0.034 + 0.008 ms   3000 / 1: 0.037 + 0.007   3000 / 4: 0.031 + 0.007   3000 / 8: 0.031 + 0.007
//               3000 / 16: 0.032 + 0.007   3000 / 32: 0.043 + 0.008   1500 / 8: 0.031 + 0.007   6000 / 8: 0.032 + 0.007
//   100 M docs: 12000 / 1: 0.053 + 0.008   3000 / 1: 0.053 + 0.007   3000 / 4: 0.045 + 0.007   3000 / 8: 0.042 + 0.007   3000 / 16: 0.045 + 0.007
//   (Rounds 4-5, when every block still cost its directory word, one size for all: 512 up to 6 k items, then up to 2048 beyond 30 k —
//   10 M docs 128 0.074 ms, 256 0.076, 512 0.071, 1024 0.082; 100 M docs 256 0.315, 512 0.215, 1024 and 2048 0.178.)
static int term_item_blocks(int64_t total_blocks, int64_t target_items) {
  int blocks_per_item = 8;
  while (blocks_per_item < 4096 && total_blocks / blocks_per_item > target_items) blocks_per_item *= 2;
  return blocks_per_item;
}
// ... and for ONE query of the launch: the ~13-20 blocks of a list that can enter the top-k would otherwise be unpacked one after the
// other by ONE wavefront, 1.4 us each — the launch's critical path (round 6's item timeline, scripts/term_timeline.py: the last items
// to finish were single-item queries of 150-250 blocks, 26-28 us of a 29.5 us launch, while the chip stood two thirds idle). Halving
// keeps the sizes powers of two and every item on a chunk boundary; each item starts from the term's sketch threshold, so the items of a
// query do not wait for one another.
static int term_query_item_blocks(int nblocks, int base, int split, int floor_blocks = 64) {
  int b = base;
  if (split > 1 && b > 0 && (b & (b - 1)) == 0) while (b > floor_blocks && (nblocks + b - 1) / b < split) b >>= 1;
  return b;
}
static int term_item_shift(int b) {  // log2 of a power-of-two item size for the item descriptor; 0: the launch's size
  if (b <= 0 || (b & (b - 1)) != 0) return 0;
  int sh = 0;
  while ((1 << sh) < b) ++sh;
  return sh;
}
// k_search_term's item descriptors: items 0 .. nq-1 are every query's first chunk, the other chunks follow query-major
// (item_prefix[q] = chunks of the queries in front of q beyond their first); {query, chunk, term index or -1, the query's items}
// (the query's items in the low 24 bits of .w, log2 of ITS item size above them — 0: the launch's blocks_per_item)
static void fill_term_item_desc(int4* out, const DevQuery* queries, const DevTerm* terms, const int64_t* item_prefix, int nq, int base, int split, int floor_blocks) {
  for (int q = 0; q < nq; ++q) {
    const int n_mine = 1 + (int)(item_prefix[q + 1] - item_prefix[q]);
    const int ft = queries[q].n_terms >= 1 ? queries[q].first_term : -1;
    const int sh = ft >= 0 ? term_item_shift(term_query_item_blocks(terms[ft].nblocks, base, split, floor_blocks)) : 0;
    const int w = n_mine | (sh << 24);
    out[q] = make_int4(q, 0, ft, w);
    int4* rest = out + nq + item_prefix[q];
    for (int ch = 1; ch < n_mine; ++ch) rest[ch - 1] = make_int4(q, ch, ft, w);
  }
}
static int and_item_blocks(const rgpu_ctx* c, int64_t lead_blocks) {
  if (!c->and_blocks_per_item_auto) return c->cfg.and_blocks_per_item;
  return (int)std::min<int64_t>(RGPU_AND_MAX_ITEM_BLOCKS, std::max<int64_t>(8, (lead_blocks + RGPU_AND_TARGET_ITEMS - 1) / RGPU_AND_TARGET_ITEMS));
}
static int32_t search_pass(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                           int32_t n_terms_total, int32_t k, int32_t k_total, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream);
// TopDocsCollector takes any k (collector/top_docs.rs:28-95). A wavefront's registers hold a 128-key list, so k > 128 runs as
// ceil(k / 128) passes of the whole search: pass p collects the best hits strictly below the worst hit of pass p - 1 (keys —
// score, then doc id — are unique and totally ordered: the passes partition the ranking exactly) and writes columns
// [128 p, 128 p + 128) of the caller's rows. Page 2 of a 100-per-page result list costs two passes, k = 1024 eight.
static int32_t search_impl(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                           int32_t n_terms_total, int32_t k, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  c->pass = rgpu_ctx::Pass{};
  if (k <= RGPU_PASS_K) return search_pass(seg, queries, n_queries, terms, n_terms_total, k, k, hits_dev, totals_dev, stream);
  rgpu_ctx::CeilSlot& cs = c->ceil_slots[c->ceil_next];
  c->ceil_next = (c->ceil_next + 1) % N_SCRATCH;
  if (cs.busy) { HIP_TRY(hipEventSynchronize(cs.done)); cs.busy = false; }
  HIP_TRY(cs.d.reserve((size_t)n_queries * 2, 0, stream));  // two sets, alternating: a pass reads the previous pass's, writes its own
  int32_t rc = RGPU_OK;
  int flip = 0;
  // pass p + 1 reads the ceilings pass p wrote: a group of pass p whose redo ran later would change rows and ceilings behind
  // the next pass's back — inside a multi-pass call every group settles before it returns
  const bool defer_was = c->defer_or;
  c->defer_or = false;
  for (int32_t col0 = 0; col0 < k && rc == RGPU_OK; col0 += RGPU_PASS_K, flip ^= 1) {
    c->pass.stride = k;
    c->pass.col0 = col0;
    c->pass.ceil_in = col0 == 0 ? nullptr : cs.d.p + (size_t)(flip ^ 1) * (size_t)n_queries;
    c->pass.ceil_out = cs.d.p + (size_t)flip * (size_t)n_queries;
    rc = search_pass(seg, queries, n_queries, terms, n_terms_total, std::min<int32_t>(RGPU_PASS_K, k - col0), k, hits_dev, totals_dev, stream);
  }
  c->pass = rgpu_ctx::Pass{};
  c->defer_or = defer_was;
  // no stream sync: the ceiling arrays are this call's until the slot comes round again (four calls later), like the scratch slots
  if (!cs.done) HIP_TRY(hipEventCreateWithFlags(&cs.done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(cs.done, stream));
  cs.busy = true;
  return rc;
}
// k_term_sketch for the listed terms (ctx mutex held), on the stream the search itself is about to use; ends synchronised (a
// one-off per term, ~10 us for a list of 15 k blocks). false: nothing was built (allocation, launch) — the queries do without.
static bool build_sketches_locked(rgpu_segment* seg, const std::vector<SketchJob>& jobs, const std::vector<int64_t>& fps, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  const size_t need = seg->sketch_used + jobs.size();
  auto gave_up = [&]() { (void)hipGetLastError(); return false; };
  if (seg->sketch.reserve(std::max<size_t>(need, 4096) * TERM_SKETCH_K, seg->sketch_used * TERM_SKETCH_K, stream) != hipSuccess) return gave_up();
  if (seg->sketch_jobs.reserve(jobs.size(), 0, stream) != hipSuccess) return gave_up();
  if (hipMemcpyAsync(seg->sketch_jobs.p, jobs.data(), jobs.size() * sizeof(SketchJob), hipMemcpyHostToDevice, stream) != hipSuccess) return gave_up();
  const size_t was = seg->sketch_used;
  seg->sketch_used = need;  // (seg_view hands the array out from here on)
  {
    TimedLaunch tl(c, stream, "k_term_sketch", 0);
    RGPU_LAUNCH(k_term_sketch, dim3(wg_count((jobs.size() + TERM_SKETCH_WAVES - 1) / TERM_SKETCH_WAVES)), dim3(64 * TERM_SKETCH_WAVES), 0, stream,
                seg_view(seg), seg->sketch_jobs.p, (int)jobs.size(), seg->sketch.p);
  }
  if (hipStreamSynchronize(stream) != hipSuccess || launch_status() != hipSuccess) { seg->sketch_used = was; return gave_up(); }
  for (size_t i = 0; i < jobs.size(); ++i) {
    const TermInfo* at = seg->prepared.find(fps[i]);
    if (!at) continue;
    TermInfo info = *at;
    info.sketch = jobs[i].out + 1u;
    seg->prepared.put(fps[i], info);
  }
  return true;
}
static int32_t search_pass(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                           int32_t n_terms_total, int32_t k, int32_t k_total, HitOut* hits_dev, int64_t* totals_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  HOST_STAMP(p0);
  // validate + prepare
  std::vector<const rgpu_term_state*> ptrs;
  std::vector<TermInfo> tinfo((size_t)n_terms_total);
  bool need_prepare = false;
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_query& Q = queries[q];
    const int qop = Q.op & 0xff, qmsm = (Q.op >> 8) & 0xff, qopt = (Q.op >> 16) & 0xff;
    if (qop < RGPU_OP_TERM || qop > RGPU_OP_OR || (Q.op & ~(0xffffff | RGPU_OP_SHOULD_REQUIRED | RGPU_OP_NESTED_MUST | RGPU_OP_NESTED_AT(63))) != 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "unknown query op");
    if (((uint32_t)Q.op >> 26) != 0 && (!(Q.op & (RGPU_OP_SHOULD_REQUIRED | RGPU_OP_NESTED_MUST)) || (int)((uint32_t)Q.op >> 26) > Q.n_terms))
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "RGPU_OP_NESTED_AT: the nested clause's index among the MUST clauses, 0 .. n_terms, with RGPU_OP_SHOULD_REQUIRED / RGPU_OP_NESTED_MUST");
    // "+a +(b c)": the SHOULD clauses as a nested disjunction under MUST (ConjunctionScorer over the MUST clauses and one
    // DisjunctionSumScorer). Ten or more children would sum in heap order (disjunction_scorer.rs:41-45): not served here.
    if ((Q.op & RGPU_OP_SHOULD_REQUIRED) && (qop == RGPU_OP_OR || qopt < 1))
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "RGPU_OP_SHOULD_REQUIRED goes with RGPU_OP_WITH_SHOULD(TERM / AND, n >= 1)");
    // "+a +(+b +c)": the clauses behind the MUST clauses are a nested conjunction (its own f32 sum, formed first)
    if ((Q.op & RGPU_OP_NESTED_MUST) && (qop == RGPU_OP_OR || qopt < 2 || (Q.op & RGPU_OP_SHOULD_REQUIRED)))
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "RGPU_OP_NESTED_MUST goes with RGPU_OP_WITH_SHOULD(TERM / AND, n >= 2) and without RGPU_OP_SHOULD_REQUIRED");
    if ((Q.op & RGPU_OP_SHOULD_REQUIRED) && qopt >= 10) return fail(RGPU_ERR_UNSUPPORTED, "a required disjunction of ten or more clauses sums in heap order");
    // min_should_match beside MUST clauses is legal and has no effect: ReqOptScorer only ever advance()s the optional
    // DisjunctionSumScorer, and advance() does not look at the count (disjunction_scorer.rs approximate_advance)
    if (qmsm > 1 && qop != RGPU_OP_OR && qopt == 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "min_should_match needs SHOULD clauses");
    if (qopt > 0 && qop == RGPU_OP_OR) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "optional SHOULD clauses go with MUST clauses (op TERM / AND); an OR query's clauses are its n_terms");
    // (an OR query of MUST_NOT clauses only: BooleanWeight::create_scorer -> None, boolean_query.rs:274-276 — it matches nothing)
    const int min_terms = (qop == RGPU_OP_OR && Q.n_must_not > 0) ? 0 : 1;
    if (Q.n_terms < min_terms || Q.n_terms > RGPU_MAX_QUERY_TERMS || Q.n_must_not < 0 || Q.n_must_not > RGPU_MAX_QUERY_TERMS ||
        Q.n_terms + qopt + Q.n_must_not > RGPU_MAX_QUERY_TERMS || (qop == RGPU_OP_TERM && Q.n_terms != 1))
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad clause count");
    if (Q.first_term < 0 || (int64_t)Q.first_term + Q.n_terms + qopt + Q.n_must_not > (int64_t)n_terms_total)
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "clause range outside terms[]");
    for (int i = 0; i < Q.n_terms + qopt + Q.n_must_not; ++i) {
      const rgpu_query_term& t = terms[Q.first_term + i];
      if (i < Q.n_terms + qopt && (t.sim_table < 0 || t.sim_table >= c->n_sim_tables)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "unknown sim_table handle");
      if (t.state.doc_freq > 0) ptrs.push_back(&t.state);
      // the term's prepared structures, looked up ONCE per clause and call (by value: a later look-up may grow the table)
      if (t.state.doc_freq >= 2 && !need_prepare) {
        const TermInfo* info = seg->prepared.find(t.state.doc_start_fp);
        if (info && info->df == t.state.doc_freq && (info->norms || !seg->d_norms)) tinfo[(size_t)(Q.first_term + i)] = *info;
        else need_prepare = true;  // (something is not prepared yet, or a state that must be validated: the full path below)
      }
    }
  }
  int32_t rc = RGPU_OK;
  if (need_prepare || c->prepared_budget != 0) {
    rc = prepare_terms_locked(seg, ptrs.data(), ptrs.size());
    if (rc != RGPU_OK) return rc;
    for (int32_t q = 0; q < n_queries; ++q) {
      const rgpu_query& Q = queries[q];
      const int n_all = Q.n_terms + ((Q.op >> 16) & 0xff) + Q.n_must_not;
      for (int i = 0; i < n_all; ++i) {
        const rgpu_query_term& t = terms[Q.first_term + i];
        if (t.state.doc_freq < 2) continue;
        const TermInfo* info = seg->prepared.find(t.state.doc_start_fp);
        if (!info) return fail(RGPU_ERR_ILLEGAL_STATE, "term not prepared");
        tinfo[(size_t)(Q.first_term + i)] = *info;
      }
    }
  }
  // (the fixed-point kernels rank by exact totals and round to f32 afterwards: across passes that would need a ceiling in
  // their own key space — deep result pages of a >= 10-clause disjunction go through the clause-order kernel instead)
  const bool or_wide_ok = c->cfg.or_wide >= 0 && seg->d_norms && seg->n_norm_ranks > 0 && !seg->d_live && k_total <= RGPU_PASS_K;
  if ((c->cfg.or_bitmaps >= 0 || c->cfg.and_bitmaps >= 0) && c->n_sim_tables > 0) {
    // doc bitmaps for the dense terms of the disjunctions k_or_lazy can take, and of conjunctions (k_search_and answers a
    // candidate of such a clause with one bit instead of walking the list's blocks)
    const int64_t min_df_or = bitmap_min_df(seg), min_df_and = bitmap_min_df_and(seg);
    std::vector<const rgpu_term_state*> dense;
    std::vector<int32_t> dense_sim;
    std::vector<const rgpu_term_state*> sparse_first;  // conjunctions: the clause right behind the lead, when it is below the bitmaps' density
    for (int32_t q = 0; q < n_queries; ++q) {
      const rgpu_query& Q = queries[q];
      const int qop = Q.op & 0xff, qopt = (Q.op >> 16) & 0xff;
      int n_look = 0;
      int64_t min_df = INT64_MAX;
      if (qop == RGPU_OP_OR) {
        if (or_wide_ok && ((Q.op >> 8) & 0xff) <= 1 && Q.n_terms >= 10 && Q.n_terms <= ORX_MAX_TERMS && Q.n_must_not == 0) { n_look = Q.n_terms; min_df = min_df_or; }
      } else if (Q.n_terms + qopt + Q.n_must_not >= 2) { n_look = Q.n_terms + qopt + Q.n_must_not; min_df = min_df_and; }
      // (only while a list's bits stay in an XCD's L2 — max_doc <= 16.7 M, the condition of and_xcd_chunk: measured on the 3-term
      // batch at 100 M docs, where they are 12.5 MB per list and every probe of a sparse list is an HBM sector, 1.94 ms with them
      // against 1.71 walking those clauses; at 10 M docs a batch of "rare AND medium" pairs 0.076 against 0.136 ms)
      if (qop == RGPU_OP_AND && Q.n_terms >= 2 && c->cfg.and_bitmaps >= 0 && c->memb_only_on && (int64_t)seg->max_doc / 8 <= (2ll << 20)) {
        // the MUST clause with the second smallest doc_freq (ConjunctionScorer's cost order, stable: conjunction_scorer.rs:30)
        int a = -1, b = -1;
        for (int i = 0; i < Q.n_terms; ++i) {
          const int32_t df = terms[Q.first_term + i].state.doc_freq;
          if (df <= 0) { a = b = -1; break; }  // a missing MUST clause: the conjunction matches nothing
          if (a < 0 || df < terms[Q.first_term + a].state.doc_freq) { b = a; a = i; }
          else if (b < 0 || df < terms[Q.first_term + b].state.doc_freq) b = i;
        }
        if (b >= 0) {
          const rgpu_term_state& s1 = terms[Q.first_term + b].state;
          // (worth it when the lead has blocks of its own to probe with, i.e. >= 128 postings)
          if (s1.doc_freq >= MEMB_ONLY_MIN_DF && s1.doc_freq < min_df_and && terms[Q.first_term + a].state.doc_freq >= 128 && !seg->memb_only.find(s1.doc_start_fp))
            sparse_first.push_back(&s1);
        }
      }
      for (int i = 0; i < n_look; ++i) {
        const rgpu_query_term& t = terms[Q.first_term + i];
        if (t.state.doc_freq >= min_df && !seg->bitmaps.find(t.state.doc_start_fp)) {
          dense.push_back(&t.state);
          dense_sim.push_back(t.sim_table >= 0 && t.sim_table < c->n_sim_tables ? t.sim_table : 0);
        }
      }
    }
    if (!dense.empty()) { rc = ensure_bitmaps_locked(seg, dense.data(), dense_sim.data(), dense.size()); if (rc != RGPU_OK) return rc; }
    if (!sparse_first.empty()) ensure_memb_only_locked(seg, sparse_first.data(), sparse_first.size());
  }
  // block-max sketches (search_term.hpp) for the long lists single-term queries name: built once per term, from the frontier words
  // stage B left in the directory and the table of the query that names the term first. Like the bitmaps an accelerator, never a
  // requirement: a failure to build one only means the query starts without a threshold.
  if (c->term_sketches && seg->d_norms && seg->n_norm_ranks > 0 && !seg->d_live && c->n_sim_tables > 0) {
    std::vector<SketchJob> jobs;
    std::vector<int64_t> job_fp;
    rucene::FlatFpMap<uint32_t> fresh;  // doc_start_fp -> 1 + sketch index, for the clauses of this call
    for (int32_t q = 0; q < n_queries; ++q) {
      const rgpu_query& Q = queries[q];
      if ((Q.op & 0xff) != RGPU_OP_TERM || ((Q.op >> 16) & 0xff) != 0 || Q.n_must_not != 0) continue;
      const rgpu_query_term& t = terms[Q.first_term];
      const TermInfo& ti = tinfo[(size_t)Q.first_term];
      if (t.state.doc_freq < 2 || ti.sketch != 0 || ti.nblocks < TERM_SKETCH_MIN_BLOCKS || fresh.find(t.state.doc_start_fp)) continue;
      const uint32_t idx = (uint32_t)(seg->sketch_used + jobs.size());
      jobs.push_back(SketchJob{ti.dir_base, ti.nblocks, t.sim_table, idx});
      job_fp.push_back(t.state.doc_start_fp);
      fresh.put(t.state.doc_start_fp, idx + 1u);
    }
    if (!jobs.empty() && build_sketches_locked(seg, jobs, job_fp, stream)) {
      for (int32_t q = 0; q < n_queries; ++q) {
        const rgpu_query& Q = queries[q];
        if ((Q.op & 0xff) != RGPU_OP_TERM) continue;
        if (const uint32_t* at = fresh.find(terms[Q.first_term].state.doc_start_fp)) tinfo[(size_t)Q.first_term].sketch = *at;
      }
    }
  }

  HOST_STAMP(p1);
  // one group per op; OR groups are cut so that a group's scored runs stay below ~24 GiB of HBM scratch (288 GB per GPU; with the
  // dense clauses decoded inside the window kernel only about a third of a Zipfian batch's postings go through a run at all)
  const int64_t or_postings_cap = 3000000000LL;
  std::vector<Group> groups(4);
  int cur_group[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i) groups[(size_t)i].op = i;
  groups[3].op = RGPU_OP_OR;  // disjunctions the reference sums in heap order (>= 10 sub-scorers): k_or_wide
  groups[3].or_wide = true;
  groups.emplace_back();
  groups[4].op = RGPU_OP_AND;  // MUST + SHOULD trees scored with ReqOptScorer's sequential rule (k_search_and records + k_req_opt_scan)
  groups[4].req_opt = true;
  int cur_req_opt = 4;
  int64_t req_opt_records = 0;
  std::vector<DevTerm> mine, mine_not, mine_opt, mine_suffix;
  std::vector<int64_t> mine_bytes;
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_query& Q = queries[q];
    // low byte: rgpu_query_op; next byte: min_should_match (OR); third byte: optional SHOULD clauses (TERM / AND)
    const int qop = Q.op & 0xff, qmsm = (Q.op >> 8) & 0xff, qopt = (Q.op >> 16) & 0xff;
    mine.clear();
    mine_not.clear();
    mine_opt.clear();
    mine_suffix.clear();
    mine_bytes.clear();
    bool dead = false;
    for (int i = 0; i < Q.n_terms; ++i) {
      const rgpu_query_term& t = terms[Q.first_term + i];
      if (t.state.doc_freq <= 0) {  // TermWeight::create_scorer -> None for this leaf
        if (qop != RGPU_OP_OR) dead = true;  // a missing MUST clause kills the conjunction (boolean_query.rs:201-207)
        continue;
      }
      DevTerm dt;
      rc = make_dev_term(seg, t.state, t.weight, t.sim_table, &dt, true, &tinfo[(size_t)(Q.first_term + i)]);
      if (rc != RGPU_OK) return rc;
      mine.push_back(dt);
      // the postings' bytes in .doc: FullBlocks end where the skip data starts; a short list is its VInt tail (~2 B a posting)
      mine_bytes.push_back(t.state.doc_freq > 128 && t.state.skip_offset > 0 ? t.state.skip_offset : 2 * (int64_t)t.state.doc_freq);
    }
    if (dead) mine.clear();
    const bool should_required = (Q.op & RGPU_OP_SHOULD_REQUIRED) != 0, nested_must = (Q.op & RGPU_OP_NESTED_MUST) != 0;
    // MUST_NOT clauses (boolean_query.rs:235-252): absent terms drop out; without a positive scorer there is none
    if (!mine.empty()) {
      for (int i = 0; i < qopt; ++i) {  // SHOULD next to MUST (boolean_query.rs:217-233): absent terms drop out
        const rgpu_query_term& t = terms[Q.first_term + Q.n_terms + i];
        if (t.state.doc_freq <= 0) {
          if (nested_must) { mine.clear(); mine_opt.clear(); break; }  // the nested conjunction has no scorer in this leaf (boolean_query.rs:201-207)
          continue;
        }
        DevTerm dt;
        rc = make_dev_term(seg, t.state, t.weight, t.sim_table, &dt, true, &tinfo[(size_t)(Q.first_term + Q.n_terms + i)]);
        if (rc != RGPU_OK) return rc;
        mine_opt.push_back(dt);
      }
      if (nested_must)  // the nested ConjunctionScorer::new: stable sort by cost() (conjunction_scorer.rs:30) — its sum is lead1 + lead2 + others
        std::stable_sort(mine_opt.begin(), mine_opt.end(), [](const DevTerm& a, const DevTerm& b) { return a.df < b.df; });
      // the nested disjunction has no scorer in this leaf = a MUST weight without a scorer: nothing matches (boolean_query.rs:203-207)
      if (should_required && mine_opt.empty()) mine.clear();
      for (int i = 0; i < Q.n_must_not && !mine.empty(); ++i) {
        const rgpu_query_term& t = terms[Q.first_term + Q.n_terms + qopt + i];
        if (t.state.doc_freq <= 0) continue;
        DevTerm dt;
        rc = make_dev_term(seg, t.state, 0.0f, 0, &dt, true, &tinfo[(size_t)(Q.first_term + Q.n_terms + qopt + i)]);  // needs_scores = false: weight and table are never read
        if (rc != RGPU_OK) return rc;
        mine_not.push_back(dt);
      }
    }
    // A nested scorer among the MUST children ("+a +b +(c d)", "+a +(+c +d) +b"): ConjunctionScorer::new sorts ALL its children by
    // cost(), stable (conjunction_scorer.rs:30) — a term's doc_freq, a nested DisjunctionSumScorer's the sum of its clauses', a nested
    // ConjunctionScorer's its cheapest clause's — and score() adds them in that order (:87-95). The nested child's place p in that
    // order splits the MUST terms into the ones summed before it and the ones added after it: device clause order [prefix]
    // [MUST_NOT][nested group][suffix], the kernel forms ((prefix sum) + (group sum)) + suffix clauses one by one. Ties keep the
    // caller's clause order: RGPU_OP_NESTED_AT names the nested child's index among the MUST clauses.
    bool nested_flat = false;  // the nested conjunction is the cheapest child: the tree is the flat conjunction [nested clauses][MUST terms]
    if ((should_required || nested_must) && !mine.empty() && !mine_opt.empty()) {
      const int at = (int)((uint32_t)Q.op >> 26);
      int64_t cost = nested_must ? INT64_MAX : 0;
      for (const DevTerm& m : mine_opt) cost = nested_must ? std::min<int64_t>(cost, m.df) : cost + m.df;
      size_t p = 0;  // MUST terms that precede the nested child in the stable cost order
      for (size_t i = 0; i < mine.size(); ++i) p += (mine[i].df < cost || (mine[i].df == cost && (int)i < at)) ? 1 : 0;
      std::stable_sort(mine.begin(), mine.end(), [](const DevTerm& a, const DevTerm& b) { return a.df < b.df; });
      if (p == 0 && nested_must) {
        // lead1 is the nested conjunction: (N + c1) + c2 ... — its own clauses first, in their order, then the MUST terms in theirs:
        // a flat conjunction in that clause order (and at the flat conjunction's cost: its rarest clause leads)
        mine_opt.insert(mine_opt.end(), mine.begin(), mine.end());
        mine.swap(mine_opt);
        mine_opt.clear();
        nested_flat = true;
      } else {
        if (p == 0) p = 1;  // a disjunction cannot lead here: (N + c1) == (c1 + N), the first add commutes
        mine_suffix.assign(mine.begin() + (ptrdiff_t)p, mine.end());
        mine.resize(p);
      }
    } else if (qop == RGPU_OP_AND) {  // ConjunctionScorer::new: stable sort by cost() = doc_freq (conjunction_scorer.rs:30)
      std::stable_sort(mine.begin(), mine.end(), [](const DevTerm& a, const DevTerm& b) { return a.df < b.df; });
    }
    // a term with prohibited / optional clauses runs as a one-clause conjunction (the lead-driven kernel probes them)
    const int gop = (qop == RGPU_OP_TERM && (!mine_not.empty() || !mine_opt.empty() || nested_flat)) ? (int)RGPU_OP_AND : qop;
    // disjunction_scorer.rs:41-45: >= 10 children and min_should_match <= 1 -> the heap; weights must be >= +0 (the
    // kernel's "untouched" accumulator is -0.0f)
    // (the fixed-point kernels keep per-clause state for up to 16 clauses; a longer disjunction takes the clause-order kernel)
    bool to_wide = or_wide_ok && gop == RGPU_OP_OR && mine.size() >= 10 && mine.size() <= (size_t)ORX_MAX_TERMS && qmsm <= 1 && mine_not.empty();
    for (size_t i = 0; to_wide && i < mine.size(); ++i)  // scores within [0, weight * (k1 + 1)]: what the fixed-point scale relies on
      to_wide = !std::signbit(mine[i].weight) && mine[i].weight <= 3.0e38f && c->sim_nonneg[(size_t)mine[i].sim_table];
    if (to_wide) {
      // k_or_wide's list builder and payload ring read the directory / block store / posting-order norms WITHOUT lane masks
      // (lanes past a list fall back to clause 0's block 0): that needs those arrays to exist, i.e. a prepared term. A
      // disjunction of singletons only (each lives in its term-dictionary entry, nothing is ever prepared for it) has no
      // block to walk anyway: the clause-order kernel takes it.
      bool any_blocks = false;
      for (const DevTerm& m : mine) any_blocks = any_blocks || m.nblocks > 0 || m.tail_n > 0;
      to_wide = any_blocks && seg->dir_used > 0 && seg->bstore.p != nullptr && seg->pnorm.p != nullptr;
    }
    if (gop == RGPU_OP_OR && !to_wide && groups[(size_t)cur_group[2]].postings > or_postings_cap) {
      groups.emplace_back();
      groups.back().op = RGPU_OP_OR;
      cur_group[2] = (int)groups.size() - 1;
    }
    // (a required disjunction is a child of the ConjunctionScorer: plain f32 sums, no ReqOptScorer in that tree)
    const bool to_req_opt = gop == RGPU_OP_AND && !mine_opt.empty() && !mine.empty() && c->cfg.req_opt_rule >= 0 && !should_required && !nested_must;
    if (to_req_opt) {  // one record per lead posting: keep a group's records under 1 GiB
      const int64_t lead_df = mine[0].df;
      if (req_opt_records > 0 && req_opt_records + lead_df > (64ll << 20)) {
        groups.emplace_back();
        groups.back().op = RGPU_OP_AND;
        groups.back().req_opt = true;
        cur_req_opt = (int)groups.size() - 1;
        req_opt_records = 0;
      }
      req_opt_records += lead_df;
    }
    Group& G = to_wide ? groups[3] : (to_req_opt ? groups[(size_t)cur_req_opt] : groups[(size_t)cur_group[gop]]);
    DevQuery dq;
    // the window kernel reads min_should_match from the second byte, the conjunction kernel its optional clause count
    // from the third; device clause order: MUST, MUST_NOT, SHOULD
    dq.op = gop | ((qmsm > 1 && gop == RGPU_OP_OR) ? qmsm << 8 : 0) | ((int32_t)mine_opt.size() << 16) |
            ((should_required && !mine.empty()) ? RGPU_OP_SHOULD_REQUIRED : 0) | ((nested_must && !mine.empty() && !nested_flat) ? RGPU_OP_NESTED_MUST : 0) |
            (int32_t)((uint32_t)mine_suffix.size() << 26);  // (device side: bits 26.. = MUST clauses added after the nested group)
    dq.first_term = (int32_t)G.terms.size();
    dq.n_terms = (int32_t)mine.size();
    dq.pad = (int32_t)mine_not.size();
    for (auto& m : mine) { G.terms.push_back(m); G.postings += m.df; }
    if (to_wide) G.term_bytes.insert(G.term_bytes.end(), mine_bytes.begin(), mine_bytes.end());
    for (auto& m : mine_not) { G.terms.push_back(m); G.postings += m.df; }
    for (auto& m : mine_opt) { G.terms.push_back(m); G.postings += m.df; }
    for (auto& m : mine_suffix) { G.terms.push_back(m); G.postings += m.df; }
    G.qmap.push_back(q);
    G.queries.push_back(dq);
  }

  HOST_STAMP(p2);
#ifdef RGPU_LZ_TIME
  std::fprintf(stderr, "[search_pass host] validate + prepare + bitmaps %lld us, grouping %lld us\n", HOST_US(p0, p1), HOST_US(p1, p2));
#endif
  // defaults for every query (groups overwrite their own rows) — not needed when one group holds the whole batch and
  // its merge writes every row (the usual serving case: two enqueues less per batch)
  auto init_rows = [&]() -> int32_t {
    HIP_TRY(hipMemsetAsync(totals_dev, 0, (size_t)n_queries * 8, stream));
    RGPU_LAUNCH(k_init_hits, dim3(wg_count(((size_t)n_queries * k + 255) / 256)), dim3(256), 0, stream, hits_dev,
                       (int64_t)n_queries, (int)k, c->pass.stride > 0 ? c->pass.stride : (int)k, c->pass.col0);
    if (c->pass.ceil_out) HIP_TRY(hipMemsetAsync(c->pass.ceil_out, 0, (size_t)n_queries * 8, stream));  // rows no group writes have nothing below
    return RGPU_OK;
  };
  int busy_groups = 0;
  for (const Group& G : groups) busy_groups += G.queries.empty() ? 0 : 1;
  const bool single_group = busy_groups == 1;
  if (!single_group) { int32_t rc_i = init_rows(); if (rc_i != RGPU_OK) return rc_i; }

  const bool wide = k > 64;
  const bool legacy = seg->version < 1;
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    Group& G = groups[gi];
    const int op = G.op;
    const int nq = (int)G.queries.size();
    if (nq == 0) continue;
    if (op == RGPU_OP_OR) {
      if (single_group && G.terms.empty()) { int32_t rc_i = init_rows(); if (rc_i != RGPU_OK) return rc_i; }  // nothing will be merged
      int32_t rc_or = G.or_wide ? search_or_wide_group(seg, G, k, hits_dev, totals_dev, stream)
                                : search_or_group(seg, G, k, hits_dev, totals_dev, stream);
      if (rc_or != RGPU_OK) return rc_or;
      continue;
    }
    SCRATCH_TAKE(c);
    if (op == RGPU_OP_AND && !G.req_opt && nq > 2) {
      // Conjunctions that probe the same list run side by side: the group's queries are ordered by the term of the first clause
      // behind the lead — the one every lead posting is looked up in (a doc bitmap: one gather per candidate into a 1 - 5 MB
      // array). Measured on the 1024 x 3-term batch: 0.365 ms against 0.388 in the caller's order; heaviest-lead-first 0.64 and
      // lightest-first 0.57 (a query's items start all at once, each with an empty top-k list and no published threshold; or
      // the heavy ones make the tail). The caller's rows are found through qmap, so the order is free.
      std::vector<int> order((size_t)nq);
      for (int i = 0; i < nq; ++i) order[(size_t)i] = i;
      auto key_of = [&](int i) -> uint64_t {
        const DevQuery& q0 = G.queries[(size_t)i];
        const int n_all = q0.n_terms + q0.pad + ((q0.op >> 16) & 0xff) + (int)((uint32_t)q0.op >> 26);
        return n_all >= 2 && q0.n_terms >= 1 ? G.terms[(size_t)(q0.first_term + 1)].start_fp : 0ull;
      };
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key_of(a) > key_of(b); });
      std::vector<DevQuery> qs((size_t)nq);
      std::vector<int32_t> qm((size_t)nq);
      for (int i = 0; i < nq; ++i) { qs[(size_t)i] = G.queries[(size_t)order[(size_t)i]]; qm[(size_t)i] = G.qmap[(size_t)order[(size_t)i]]; }
      G.queries.swap(qs);
      G.qmap.swap(qm);
    }
    int blocks_per_item = c->cfg.blocks_per_item;
    int term_split = 1;  // TERM: items per query of a few chunks (term_query_item_blocks), when the library sizes the items
    if (op != RGPU_OP_TERM) {
      int64_t lead_blocks = 0;
      for (const DevQuery& q0 : G.queries) if (q0.n_terms >= 1) lead_blocks += G.terms[(size_t)q0.first_term].nblocks;
      blocks_per_item = and_item_blocks(c, lead_blocks);
    }
    int64_t items = 0;
    G.item_prefix.assign((size_t)nq + 1, 0);
    const int head_items = op == RGPU_OP_TERM ? nq : 0;  // TERM: every query's first chunk is scheduled first
    {  // items = chunks of the (lead) term's blocks; the last chunk also takes its tail
      if (op == RGPU_OP_TERM && c->blocks_per_item_auto) {  // fewer, longer items when there are plenty of blocks
        int64_t total_blocks = 0;
        for (auto& t : G.terms) total_blocks += t.nblocks;
        blocks_per_item = term_item_blocks(total_blocks, c->term_target_items);
        term_split = c->term_split;
      }
      while (true) {
        items = 0;
        for (int q = 0; q < nq; ++q) {
          G.item_prefix[(size_t)q] = items;
          if (G.queries[(size_t)q].n_terms >= 1) {
            const DevTerm& t = G.terms[(size_t)G.queries[(size_t)q].first_term];
            const int mine_blocks = op == RGPU_OP_TERM ? term_query_item_blocks(t.nblocks, blocks_per_item, term_split, c->term_min_item_blocks) : blocks_per_item;
            const int64_t mine = t.nblocks == 0 ? 1 : (t.nblocks + mine_blocks - 1) / mine_blocks;
            items += head_items ? mine - 1 : mine;
          }
        }
        G.item_prefix[(size_t)nq] = items;
        if (items <= 262144 || blocks_per_item >= (1 << 17)) break;
        blocks_per_item *= 2;
      }
      items += head_items;
    }
    if (items == 0) {
      if (single_group) { int32_t rc_i = init_rows(); if (rc_i != RGPU_OK) return rc_i; }
      continue;
    }
    // plan + thresholds in one staged copy; the merge writes the caller's rows in place (qmap)
    Stager st(c);
    const size_t o_q = st.add((size_t)nq * sizeof(DevQuery));
    const size_t o_t = st.add(std::max<size_t>(1, G.terms.size()) * sizeof(DevTerm));
    const size_t o_p = st.add((size_t)(nq + 1) * 8);
    const size_t o_m = st.add((size_t)nq * 4);
    const size_t o_tau = st.add((size_t)nq * 8);  // the per-query shared thresholds: zeroed by the same copy that brings the plan
    const bool term_fold = op == RGPU_OP_TERM && c->term_fold && !G.req_opt;  // k_search_term folds the item lists itself (TermMerge)
    const size_t o_done = term_fold ? st.add((size_t)nq * 4) : 0;             // ... counting every query's finished items here
    // ReqOptScorer's rule: one record per lead posting, query after query
    const size_t o_sp = G.req_opt ? st.add((size_t)(nq + 1) * 8) : 0;
    // conjunctions: the doc bitmaps of the clauses behind the lead (parallel to the DevTerm array)
    std::vector<TermBitmap> clause_bitmaps;
    if (op == RGPU_OP_AND && c->cfg.and_bitmaps >= 0 && (seg->bitmaps.size() > 0 || seg->memb_only.size() > 0)) {
      const int64_t min_df = bitmap_min_df_and(seg);
      bool any = false;
      clause_bitmaps.assign(G.terms.size(), TermBitmap{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0});
      for (const DevQuery& q0 : G.queries) {
        const int n_all = q0.n_terms + q0.pad + ((q0.op >> 16) & 0xff) + (int)((uint32_t)q0.op >> 26);
        for (int i = 1; i < n_all; ++i) {
          const DevTerm& t = G.terms[(size_t)(q0.first_term + i)];
          if (t.df < min_df) {
            // the first MUST clause behind the lead may carry its membership bits alone (words == null: the batched probe asks
            // them, the survivors walk the clause's block directory)
            if (i == 1 && q0.n_terms >= 2 && !G.req_opt && seg->memb_only.size() > 0) {
              const rgpu_segment::MembOnly* mo = seg->memb_only.find((int64_t)t.start_fp);
              if (mo && mo->memb != nullptr && mo->df == t.df) {
                clause_bitmaps[(size_t)(q0.first_term + 1)] = TermBitmap{nullptr, nullptr, nullptr, nullptr, nullptr, mo->memb, 0, 0};
                any = true;
              }
            }
            continue;
          }
          const BitmapInfo* bm = seg->bitmaps.find((int64_t)t.start_fp);
          if (!bm || !bm->usable || bm->df != t.df) continue;
          clause_bitmaps[(size_t)(q0.first_term + i)] = TermBitmap{bm->words, bm->ranks, bm->freqs, bm->ovf, bm->nib, bm->memb, bm->n_ovf, 0};
          any = true;
        }
      }
      if (!any) clause_bitmaps.clear();
    }
    const size_t o_bm = clause_bitmaps.empty() ? 0 : st.add(clause_bitmaps.size() * sizeof(TermBitmap));
    const size_t o_id = op == RGPU_OP_TERM ? st.add((size_t)items * sizeof(int4)) : 0;  // k_search_term's item descriptors
    std::vector<int64_t> seq_prefix;
    if (G.req_opt) {
      seq_prefix.assign((size_t)nq + 1, 0);
      for (int q = 0; q < nq; ++q) {
        const DevQuery& dq0 = G.queries[(size_t)q];
        seq_prefix[(size_t)q + 1] = seq_prefix[(size_t)q] + (dq0.n_terms >= 1 ? (int64_t)G.terms[(size_t)dq0.first_term].df : 0);
      }
    }
    HIP_TRY(c->S->h_stage.reserve(st.used));
    HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
    std::memcpy(c->S->h_stage.p + o_q, G.queries.data(), (size_t)nq * sizeof(DevQuery));
    if (!G.terms.empty()) std::memcpy(c->S->h_stage.p + o_t, G.terms.data(), G.terms.size() * sizeof(DevTerm));
    std::memcpy(c->S->h_stage.p + o_p, G.item_prefix.data(), (size_t)(nq + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_m, G.qmap.data(), (size_t)nq * 4);
    std::memset(c->S->h_stage.p + o_tau, 0, (size_t)nq * 8);
    if (term_fold) std::memset(c->S->h_stage.p + o_done, 0, (size_t)nq * 4);
    if (G.req_opt) std::memcpy(c->S->h_stage.p + o_sp, seq_prefix.data(), (size_t)(nq + 1) * 8);
    if (!clause_bitmaps.empty()) std::memcpy(c->S->h_stage.p + o_bm, clause_bitmaps.data(), clause_bitmaps.size() * sizeof(TermBitmap));
    if (op == RGPU_OP_TERM) fill_term_item_desc(reinterpret_cast<int4*>(c->S->h_stage.p + o_id), G.queries.data(), G.terms.data(), G.item_prefix.data(), nq, blocks_per_item, term_split, c->term_min_item_blocks);
    HIP_TRY(stage_h2d(c, st.used, stream));
    HIP_TRY(c->S->d_partial_keys.reserve((size_t)items * (size_t)k, 0, stream));
    HIP_TRY(c->S->d_partial_counts.reserve((size_t)items, 0, stream));
    unsigned long long* d_tau = reinterpret_cast<unsigned long long*>(c->S->d_stage.p + o_tau);
#ifdef RGPU_EXP_KEEP_TAU  // developer experiment (variant builds only): a launch starts from the thresholds the previous launch of
    {                     // the SAME batch ended with — what the kernel costs when every item knows its query's final k-th key
      static unsigned long long* keep = nullptr;
      if (!keep) { HIP_TRY(hipMalloc(&keep, (size_t)1 << 20)); HIP_TRY(hipMemset(keep, 0, (size_t)1 << 20)); }
      if ((size_t)nq * 8 <= ((size_t)1 << 20)) d_tau = keep;
    }
#endif
    const DevQuery* dq = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
    const DevTerm* dt = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
    const int64_t* dp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_p);
    const int32_t* dm = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_m);
    const SegView sv = seg_view(seg);
    if (op == RGPU_OP_AND) {
      HIP_TRY(c->S->d_touched.reserve((size_t)nq * 2, 0, stream));
      HIP_TRY(hipMemsetAsync(c->S->d_touched.p, 0, (size_t)nq * 16, stream));
      c->last_and = c->S;
      c->last_and_queries = nq;
      c->last_counted = c->S;
      c->last_counted_words = c->S->d_touched.p;
      c->last_counted_op = RGPU_OP_AND;
      c->last_counted_queries = nq;
      c->last_counted_postings = G.postings;
      c->last_counted_loose = 0;
      for (const DevQuery& q0 : G.queries) if (q0.n_terms >= 1) { const DevTerm& t0 = G.terms[(size_t)q0.first_term]; c->last_counted_loose += t0.df == 1 ? 1 : t0.tail_n; }
      TimedLaunch tl(c, stream, "k_search_and", G.postings);
      // (whole rounds of 8 x AND_XCD_CHUNK workgroups: the kernel deals chunks of workgroups to the XCDs, search_and.hpp)
      const int xcd_chunk = and_xcd_chunk(seg);
      const unsigned grid = wg_count(and_grid((items + AND_WG_WAVES - 1) / AND_WG_WAVES, xcd_chunk));
      const int64_t* d_sp = nullptr;
      void* d_seq = nullptr;
      if (G.req_opt) {  // records instead of a collector (in the slot's own run buffer: the group only marks its slot)
        static_assert(sizeof(SeqRec) == 2 * sizeof(ScoredPosting), "SeqRec records are laid over the run scratch");
        HIP_TRY(c->S->d_runs.reserve((size_t)seq_prefix[(size_t)nq] * 2 + 64, 0, stream));  // (a group's records stay under 1 GiB: req_opt_records)
        d_sp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_sp);
        d_seq = c->S->d_runs.p;
      }
      auto go = [&](auto kern) {
        RGPU_LAUNCH(kern, dim3(grid), dim3(AND_WG_THREADS), 0, stream, sv, dq, dt, dp, nq, items, blocks_per_item, (int)k,
                           c->S->d_partial_keys.p, c->S->d_partial_counts.p, d_tau, c->S->d_touched.p,
                           d_sp, (unsigned long long*)nullptr, d_seq, c->pass.ceil_in, dm,
                           clause_bitmaps.empty() ? (const TermBitmap*)nullptr : reinterpret_cast<const TermBitmap*>(c->S->d_stage.p + o_bm), xcd_chunk);
      };
      bool has_not = false, has_opt = false;
      for (const DevQuery& q : G.queries) { has_not = has_not || q.pad != 0; has_opt = has_opt || ((q.op >> 16) & 0xff) != 0; }
      if (has_opt) {  // one instantiation serves MUST_NOT too (rare trees: keep the instantiation count down)
        if (legacy) { if (wide) go(k_search_and<true, true, true, true>); else go(k_search_and<true, false, true, true>); }
        else { if (wide) go(k_search_and<false, true, true, true>); else go(k_search_and<false, false, true, true>); }
      } else if (has_not) {
        if (legacy) { if (wide) go(k_search_and<true, true, true, false>); else go(k_search_and<true, false, true, false>); }
        else { if (wide) go(k_search_and<false, true, true, false>); else go(k_search_and<false, false, true, false>); }
      } else {
        if (legacy) { if (wide) go(k_search_and<true, true, false, false>); else go(k_search_and<true, false, false, false>); }
        else { if (wide) go(k_search_and<false, true, false, false>); else go(k_search_and<false, false, false, false>); }
      }
    } else if (op == RGPU_OP_TERM) {
      HIP_TRY(c->S->d_touched.reserve((size_t)nq * 2, 0, stream));
      HIP_TRY(hipMemsetAsync(c->S->d_touched.p, 0, (size_t)nq * 16, stream));
      c->last_counted = c->S;
      c->last_counted_words = c->S->d_touched.p;
      c->last_counted_op = RGPU_OP_TERM;
      c->last_counted_queries = nq;
      c->last_counted_postings = G.postings;
      c->last_counted_dir_blocks = 0;
      c->last_counted_loose = 0;
      for (const DevTerm& t0 : G.terms) { c->last_counted_dir_blocks += t0.nblocks; c->last_counted_loose += t0.df == 1 ? 1 : t0.tail_n; }
      const TermMerge fold = term_fold ? TermMerge{reinterpret_cast<unsigned int*>(c->S->d_stage.p + o_done), dp, hits_dev, totals_dev, c->pass.ceil_out,
                                                   seg->doc_base, c->pass.stride, c->pass.col0}
                                       : TermMerge{};
      TimedLaunch tl(c, stream, "k_search_term", G.postings);
      const unsigned grid = wg_count((items + TERM_WAVES - 1) / TERM_WAVES);
      const size_t lds = term_lds_bytes(wide);
      auto go = [&](auto kern) -> hipError_t {
        hipError_t e = set_dynamic_lds_once(c, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        RGPU_LAUNCH(kern, dim3(grid), dim3(TERM_THREADS), lds, stream, sv, dq, dt, reinterpret_cast<const int4*>(c->S->d_stage.p + o_id), nq, items,
                           blocks_per_item, (int)k, c->S->d_partial_keys.p, c->S->d_partial_counts.p, d_tau, c->S->d_touched.p, c->pass.ceil_in, dm, fold);
        return hipSuccess;
      };
      hipError_t e;
      if (legacy) e = wide ? go(k_search_term<true, true>) : go(k_search_term<true, false>);
      else e = wide ? go(k_search_term<false, true>) : go(k_search_term<false, false>);
      HIP_TRY(e);
    }
    if (G.req_opt) {  // the scan is the collector: rows written in place
      TimedLaunch tl(c, stream, "k_req_opt_scan", G.postings);
      const unsigned grid = wg_count((nq + WG_WAVES - 1) / WG_WAVES);
      const SeqRec* d_seq = reinterpret_cast<const SeqRec*>(c->S->d_runs.p);
      const int64_t* d_sp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_sp);
      if (wide) RGPU_LAUNCH(k_req_opt_scan<true>, dim3(grid), dim3(WG_THREADS), 0, stream, d_seq, d_sp, nq, (int)k, seg->doc_base, dm, hits_dev, totals_dev,
                                   c->pass.stride, c->pass.col0, c->pass.ceil_in, c->pass.ceil_out);
      else RGPU_LAUNCH(k_req_opt_scan<false>, dim3(grid), dim3(WG_THREADS), 0, stream, d_seq, d_sp, nq, (int)k, seg->doc_base, dm, hits_dev, totals_dev,
                              c->pass.stride, c->pass.col0, c->pass.ceil_in, c->pass.ceil_out);
      HIP_TRY(launch_status());
      HIP_TRY(scratch_mark(c, stream));  // no stream sync: the records live in the slot
      continue;
    }
    // the group's rows go straight to the caller's (qmap): no scatter pass
    if (term_fold) {}  // (written by k_search_term's last wavefront per query)
    else if (wide) launch_merge<true>(c, stream, nq, k, dp, seg->doc_base, hits_dev, totals_dev, head_items, nullptr, nullptr, dm);
    else launch_merge<false>(c, stream, nq, k, dp, seg->doc_base, hits_dev, totals_dev, head_items, nullptr, nullptr, dm);
    HIP_TRY(launch_status());
    HIP_TRY(scratch_mark(c, stream));  // no stream sync: the slot is waited for when it is taken again
  }
#ifdef RGPU_LZ_TIME
  { HOST_STAMP(p3); std::fprintf(stderr, "[search_pass host] items + staging + launches %lld us\n", HOST_US(p2, p3)); }
#endif
  return RGPU_OK;
}

extern "C" int32_t rgpu_search_batch_device(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries,
                                            const rgpu_query_term* terms, int32_t n_terms_total, int32_t k, void* hits_dev,
                                            void* totals_dev, void* hip_stream) {
  if (!seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !hits_dev || !totals_dev)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : seg->ctx->stream;
  // enqueue only: TERM / AND batches return without waiting for the GPU (the caller synchronizes the stream, or
  // calls rgpu_synchronize for the context's own, before it reads the outputs); >= 10-clause disjunctions too with
  // rgpu_config.or_deferred (the caller then calls rgpu_synchronize before it reads them)
  rgpu_ctx* c = seg->ctx;
  c->defer_or = c->cfg.or_deferred != 0;
  const int32_t rc = search_impl(seg, queries, n_queries, terms, n_terms_total, k, (HitOut*)hits_dev, (int64_t*)totals_dev, s);
  c->defer_or = false;
  return rc;
}

extern "C" int32_t rgpu_search_batch(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                                     int32_t n_terms_total, int32_t k, rgpu_hit* hits_out, int64_t* total_hits_out) {
  if (!seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !hits_out || !total_hits_out)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  // device-side result rows of the blocking variant live in the context (grow-only): the call ends with a stream
  // sync, so they are free again when it returns
  // IndexSearcher::search is ONE query per call (search/searcher.rs:487-525): a batch of one is a launch-latency exercise — 66 us
  // of wall time around 23 us of kernels (round 6's latency leg), two of its six stream operations being the device-to-host
  // copies of 88 bytes. Rows of up to 256 KiB are therefore written by the kernels straight into pinned host memory
  // (hipHostMalloc: device-visible, coherent) and copied to the caller's arrays by the CPU after the one stream sync.
  const size_t hits_bytes = (size_t)n_queries * (size_t)k * sizeof(HitOut), tot_bytes = (size_t)n_queries * 8;
  if (hits_bytes + tot_bytes <= ((size_t)256 << 10)) {
    HIP_TRY(c->host_api_rows.reserve(hits_bytes + tot_bytes));
    HitOut* p_hits = reinterpret_cast<HitOut*>(c->host_api_rows.p);
    int64_t* p_tot = reinterpret_cast<int64_t*>(c->host_api_rows.p + hits_bytes);
    int32_t rc = search_impl(seg, queries, n_queries, terms, n_terms_total, k, p_hits, p_tot, c->stream);
    if (rc != RGPU_OK) return rc;
    rc = settle_pending(c);  // (nothing is deferred on this path; a closure left by an earlier deferred batch may own the stream's tail)
    if (rc != RGPU_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::memcpy(hits_out, p_hits, hits_bytes);
    std::memcpy(total_hits_out, p_tot, tot_bytes);
    return RGPU_OK;
  }
  HIP_TRY(c->host_api_hits.reserve((size_t)n_queries * (size_t)k, 0, c->stream));
  HIP_TRY(c->host_api_totals.reserve((size_t)n_queries, 0, c->stream));
  HitOut* d_hits = c->host_api_hits.p;
  int64_t* d_tot = c->host_api_totals.p;
  int32_t rc = search_impl(seg, queries, n_queries, terms, n_terms_total, k, d_hits, d_tot, c->stream);
  if (rc == RGPU_OK) {
    hipError_t e1 = hipMemcpyAsync(hits_out, d_hits, (size_t)n_queries * (size_t)k * sizeof(HitOut), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipMemcpyAsync(total_hits_out, d_tot, (size_t)n_queries * 8, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(RGPU_ERR_RUNTIME, "device to host copy failed");
  }
  return rc;
}

extern "C" int32_t rgpu_merge_topk_device(rgpu_ctx* c, const void* hits_dev, const void* totals_dev, int32_t n_lists, int32_t n_queries,
                                          int32_t k, void* hits_out_dev, void* totals_out_dev, void* hip_stream) {
  if (!c || !hits_dev || !totals_dev || !hits_out_dev || !totals_out_dev || n_lists <= 0 || n_queries <= 0)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  {
    TimedLaunch tl(c, s, "k_merge_lists", 0);
    const unsigned grid = wg_count((n_queries + WG_WAVES - 1) / WG_WAVES);
    if (k > 64)
      RGPU_LAUNCH(k_merge_lists<true>, dim3(grid), dim3(WG_THREADS), 0, s, (const HitOut*)hits_dev, (const int64_t*)totals_dev,
                         (int64_t)n_queries * k, (int64_t)n_queries, n_lists, n_queries, k, (HitOut*)hits_out_dev, (int64_t*)totals_out_dev);
    else
      RGPU_LAUNCH(k_merge_lists<false>, dim3(grid), dim3(WG_THREADS), 0, s, (const HitOut*)hits_dev, (const int64_t*)totals_dev,
                         (int64_t)n_queries * k, (int64_t)n_queries, n_lists, n_queries, k, (HitOut*)hits_out_dev, (int64_t*)totals_out_dev);
  }
  HIP_TRY(launch_status());
  return RGPU_OK;  // enqueue only, like rgpu_search_batch_device
}

// ---- exact phrases -----------------------------------------------------------------------------------------------------------
extern "C" int32_t rgpu_segment_attach_positions(rgpu_segment* seg, const uint8_t* pos_file, size_t pos_len) {
  if (!seg || !pos_file) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (!seg->has_positions) return fail(RGPU_ERR_ILLEGAL_STATE, "the segment was not uploaded as a positions field (index_options >= 3)");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  std::vector<uint8_t> head(128);
  const size_t head_n = std::min(seg->doc_len, head.size());
  HIP_TRY(hipMemcpy(head.data(), seg->d_doc, head_n, hipMemcpyDeviceToHost));  // the .doc header: id + suffix to compare with
  int64_t start = 0;
  std::string why;
  const int rc = rucene::parse_pos_file(pos_file, pos_len, head.data(), seg->version, &start, &why);
  if (rc != 0) return fail(rc, why);
  if (seg->d_pos) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(seg->d_pos); seg->d_pos = nullptr; }
  const size_t pad = 8192;  // speculative row / VInt-block loads may run past the last position byte
  HIP_TRY(hipMalloc(&seg->d_pos, pos_len + pad));
  HIP_TRY(hipMemcpy(seg->d_pos, pos_file, pos_len, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(seg->d_pos + pos_len, 0, pad));
  seg->pos_len = pos_len;
  return RGPU_OK;
}

// Lucene50PostingsReader::open's third file (posting_reader.rs:131-156). Nothing on this path reads payload bytes or offsets
// (phrase scorers ask for positions only and get the iterator that walks past them, :189-212), so the file is checked — header,
// version, segment id / suffix, footer — as open() checks it, and not kept.
extern "C" int32_t rgpu_segment_attach_payloads(rgpu_segment* seg, const uint8_t* pay_file, size_t pay_len) {
  if (!seg || !pay_file) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (!seg->has_offsets && !seg->has_payloads)
    return fail(RGPU_ERR_ILLEGAL_STATE, "the segment's field stores neither payloads nor offsets (index_options 4 / RGPU_FIELD_STORES_PAYLOADS)");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  std::vector<uint8_t> head(128);
  const size_t head_n = std::min(seg->doc_len, head.size());
  HIP_TRY(hipMemcpy(head.data(), seg->d_doc, head_n, hipMemcpyDeviceToHost));
  int64_t start = 0;
  std::string why;
  const int rc = rucene::parse_pay_file(pay_file, pay_len, head.data(), seg->version, &start, &why);
  if (rc != 0) return fail(rc, why);
  seg->pay_checked = true;
  return RGPU_OK;
}

// BlockPostingIterator::{next, next_position} to exhaustion for every given term (kernels/decode_positions.hpp): ends synchronised
static int32_t decode_positions_impl(rgpu_segment* seg, const rgpu_term_state* terms, const rgpu_term_positions* positions, int64_t n_terms,
                                     int32_t* positions_dev, hipStream_t stream) {
  rgpu_ctx* c = seg->ctx;
  if (!seg->has_positions || !seg->d_pos) return fail(RGPU_ERR_ILLEGAL_STATE, "a positions decode needs a positions field with its .pos file attached");
  std::vector<const rgpu_term_state*> ptrs;
  int64_t expect = 0;
  for (int64_t i = 0; i < n_terms; ++i) {
    const rgpu_term_state& st = terms[i];
    if (st.doc_freq < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "negative doc_freq");
    if (st.doc_freq == 0) continue;
    if (st.total_term_freq < st.doc_freq) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "total_term_freq below doc_freq");
    if (positions[i].pos_start_fp < 0 || (size_t)positions[i].pos_start_fp >= seg->pos_len) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "pos_start_fp outside the .pos file");
    expect += st.total_term_freq;
    ptrs.push_back(&st);
  }
  if (expect == 0) return RGPU_OK;
  if (expect > 0xfffffff0ll) return fail(RGPU_ERR_UNSUPPORTED, "one call decodes at most 2^32 positions: split the term list");
  int32_t rc = prepare_terms_locked(seg, ptrs.data(), ptrs.size(), false);  // stage A: directory (with dir_pos), block store, tails
  if (rc != RGPU_OK) return rc;
  std::vector<DevTerm> dt;
  std::vector<PosTerm> pt;
  std::vector<int64_t> item_prefix;
  int64_t items = 0;
  for (int64_t i = 0; i < n_terms; ++i) {
    const rgpu_term_state& st = terms[i];
    if (st.doc_freq == 0) continue;
    DevTerm d;
    rc = make_dev_term(seg, st, 0.f, 0, &d, false);
    if (rc != RGPU_OK) return rc;
    PosTerm p{};
    p.pos_start_fp = (uint64_t)positions[i].pos_start_fp;
    p.total_term_freq = st.total_term_freq;
    // posting_reader.rs:1195-1203: fewer than 128 positions -> all VInts; exactly 128 -> one packed block, no trailing one
    p.last_pos_block_fp = st.total_term_freq < 128 ? positions[i].pos_start_fp
                          : (st.total_term_freq == 128 ? -1 : positions[i].pos_start_fp + positions[i].last_pos_block_offset);
    dt.push_back(d);
    pt.push_back(p);
    item_prefix.push_back(items);
    items += (int64_t)d.nblocks + ((d.df == 1 || d.tail_n > 0) ? 1 : 0);
  }
  item_prefix.push_back(items);
  const int nt = (int)dt.size();
  SCRATCH_TAKE(c);
  Stager st(c);
  const size_t o_t = st.add(dt.size() * sizeof(DevTerm));
  const size_t o_pt = st.add(pt.size() * sizeof(PosTerm));
  const size_t o_ip = st.add(item_prefix.size() * 8);
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
  std::memcpy(c->S->h_stage.p + o_t, dt.data(), dt.size() * sizeof(DevTerm));
  std::memcpy(c->S->h_stage.p + o_pt, pt.data(), pt.size() * sizeof(PosTerm));
  std::memcpy(c->S->h_stage.p + o_ip, item_prefix.data(), item_prefix.size() * 8);
  HIP_TRY(stage_h2d(c, st.used, stream));
  const int64_t n_tiles = (items + SCAN_TILE - 1) / SCAN_TILE;
  HIP_TRY(c->pos_counts.reserve((size_t)items + 64, 0, stream));
  HIP_TRY(c->pos_tiles.reserve((size_t)n_tiles + 8, 0, stream));
  HIP_TRY(hipMemsetAsync(c->d_err, 0, 4 * sizeof(int), stream));
  const DevTerm* d_t = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
  const PosTerm* d_pt = reinterpret_cast<const PosTerm*>(c->S->d_stage.p + o_pt);
  const int64_t* d_ip = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_ip);
  unsigned long long* d_tiles = c->pos_tiles.p + 8;
  unsigned long long* d_total = c->pos_tiles.p + 1;
  const SegView sv = seg_view(seg);
  const bool legacy = seg->version < 1;
  const unsigned grid = wg_count((items + WG_WAVES - 1) / WG_WAVES);
  {
    TimedLaunch tl(c, stream, "k_pos_counts", expect);
    if (legacy) RGPU_LAUNCH(k_pos_counts<true>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, d_t, d_ip, nt, items, c->pos_counts.p);
    else RGPU_LAUNCH(k_pos_counts<false>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, d_t, d_ip, nt, items, c->pos_counts.p);
  }
  {
    TimedLaunch tl(c, stream, "k_scan_rows", 0);
    RGPU_LAUNCH(k_scan_reduce, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, stream, c->pos_counts.p, items, d_tiles);
    RGPU_LAUNCH(k_scan_tiles, dim3(1), dim3(PREP_THREADS), 0, stream, d_tiles, n_tiles, (unsigned long long)expect, d_total, c->d_err);
    RGPU_LAUNCH(k_scan_down, dim3((unsigned)n_tiles), dim3(PREP_THREADS), 0, stream, c->pos_counts.p, items, d_tiles);
  }
  {
    TimedLaunch tl(c, stream, "k_decode_positions", expect);
    if (legacy)
      RGPU_LAUNCH(k_decode_positions<true>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, d_t, d_pt, d_ip, nt, items, c->pos_counts.p,
                         (int64_t)seg->pos_len, positions_dev, c->d_err);
    else
      RGPU_LAUNCH(k_decode_positions<false>, dim3(grid), dim3(WG_THREADS), 0, stream, sv, d_t, d_pt, d_ip, nt, items, c->pos_counts.p,
                         (int64_t)seg->pos_len, positions_dev, c->d_err);
  }
  int err4[4] = {0, 0, 0, 0};
  unsigned long long total = 0;
  HIP_TRY(hipMemcpyAsync(err4, c->d_err, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(launch_status());
  if (err4[0] != 0 || total != (unsigned long long)expect)
    return fail(RGPU_ERR_CORRUPT_INDEX, total != (unsigned long long)expect ? "the postings' freqs do not add up to total_term_freq" : "corrupt position data in .pos");
  return RGPU_OK;
}

extern "C" int32_t rgpu_decode_positions_device(rgpu_segment* seg, const rgpu_term_state* terms, const rgpu_term_positions* positions, int64_t n_terms,
                                                void* positions_dev, void* hip_stream) {
  if (!seg || !terms || !positions || n_terms <= 0 || !positions_dev) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  return decode_positions_impl(seg, terms, positions, n_terms, (int32_t*)positions_dev, hip_stream ? (hipStream_t)hip_stream : seg->ctx->stream);
}

extern "C" int32_t rgpu_decode_positions(rgpu_segment* seg, const rgpu_term_state* terms, const rgpu_term_positions* positions, int64_t n_terms,
                                         int32_t* positions_out) {
  if (!seg || !terms || !positions || n_terms <= 0 || !positions_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  int64_t total = 0;
  for (int64_t i = 0; i < n_terms; ++i) if (terms[i].doc_freq > 0) total += std::max<int64_t>(0, terms[i].total_term_freq);
  if (total == 0) return RGPU_OK;
  int32_t* d_pos = nullptr;
  HIP_TRY(hipMalloc(&d_pos, (size_t)total * 4));
  int32_t rc = decode_positions_impl(seg, terms, positions, n_terms, d_pos, c->stream);
  if (rc == RGPU_OK && hipMemcpy(positions_out, d_pos, (size_t)total * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RGPU_ERR_RUNTIME, "device to host copy failed");
  (void)hipFree(d_pos);
  return rc;
}

extern "C" int32_t rgpu_search_phrase_batch(rgpu_segment* seg, const rgpu_phrase_query* queries, int32_t n_queries,
                                            const rgpu_phrase_term* terms, int32_t n_terms_total, int32_t k, rgpu_hit* hits_out,
                                            int64_t* total_hits_out) {
  if (!seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !hits_out || !total_hits_out)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "phrase search: k must be in 1..RGPU_MAX_K");
  if (!seg->has_positions || !seg->d_pos) return fail(RGPU_ERR_ILLEGAL_STATE, "phrase search needs a positions field with its .pos file attached");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = c->stream;
  // ---- validate, resolve, plan
  std::vector<const rgpu_term_state*> ptrs;
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_phrase_query& Q = queries[q];
    if (Q.n_terms < 2 || Q.n_terms > RGPU_MAX_PHRASE_TERMS) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "a phrase has 2..RGPU_MAX_PHRASE_TERMS terms");
    if (Q.slop < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "Slop must be >= 0");  // PhraseQuery::new (phrase_query.rs:77)
    if (Q.next_limit < RGPU_NEXT_LIMIT_ZERO) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rgpu_phrase_query.next_limit: 0 (the searcher's default), n > 0, -1 (none) or RGPU_NEXT_LIMIT_ZERO");
    if (Q.first_term < 0 || (int64_t)Q.first_term + Q.n_terms > (int64_t)n_terms_total) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term range outside terms[]");
    if (Q.sim_table < 0 || Q.sim_table >= c->n_sim_tables) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "unknown sim_table handle");
    for (int i = 0; i < Q.n_terms; ++i) {
      const rgpu_phrase_term& t = terms[Q.first_term + i];
      if (t.state.doc_freq > 0) {
        if (t.positions.pos_start_fp < 0 || (size_t)t.positions.pos_start_fp >= seg->pos_len) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "pos_start_fp outside the .pos file");
        if (t.state.total_term_freq < t.state.doc_freq) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "total_term_freq below doc_freq");
        ptrs.push_back(&t.state);
      }
    }
  }
  int32_t rc = prepare_terms_locked(seg, ptrs.data(), ptrs.size());
  if (rc != RGPU_OK) return rc;
  // the conjunction that finds a phrase's candidates answers a dense clause from its doc bitmap, as a BooleanQuery's does
  const int64_t bitmap_df = (c->cfg.and_bitmaps >= 0 && c->n_sim_tables > 0) ? bitmap_min_df_and(seg) : INT64_MAX;
  if (bitmap_df != INT64_MAX) {
    std::vector<const rgpu_term_state*> dense;
    std::vector<int32_t> dense_sim;
    for (int32_t q = 0; q < n_queries; ++q) {
      const rgpu_phrase_query& Q = queries[q];
      if (Q.n_terms < 2) continue;
      for (int i = 0; i < Q.n_terms; ++i) {
        const rgpu_term_state& st = terms[Q.first_term + i].state;
        if (st.doc_freq >= bitmap_df && !seg->bitmaps.find(st.doc_start_fp)) { dense.push_back(&st); dense_sim.push_back(Q.sim_table); }
      }
    }
    if (!dense.empty()) { rc = ensure_bitmaps_locked(seg, dense.data(), dense_sim.data(), dense.size()); if (rc != RGPU_OK) return rc; }
  }
  std::vector<DevQuery> dq((size_t)n_queries);
  std::vector<DevTerm> dt;
  std::vector<PosTerm> pt;
  std::vector<int64_t> item_prefix((size_t)n_queries + 1), emit_prefix((size_t)n_queries + 1), collect_prefix((size_t)n_queries + 1);
  std::vector<int32_t> slops((size_t)n_queries, 0), limits((size_t)n_queries, -1);
  int64_t collect_items = 0;
  bool any_cutoff = false;
  bool any_sloppy = false, any_exact = false;
  bool sloppy_rpts = false;  // some sloppy phrase names a term twice (SloppyPhraseScorer's repetition groups: k_sloppy_groups)
  int64_t phrase_lead_blocks = 0;  // the conjunctions' lead blocks: every phrase's rarest term
  for (int32_t q = 0; q < n_queries; ++q) {
    int64_t least = INT64_MAX;
    for (int i = 0; i < queries[q].n_terms; ++i) least = std::min<int64_t>(least, std::max<int64_t>(0, terms[queries[q].first_term + i].state.doc_freq));
    if (least != INT64_MAX) phrase_lead_blocks += least / 128;
  }
  const int blocks_per_item = and_item_blocks(c, phrase_lead_blocks);
  int64_t items = 0, slots = 0;
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_phrase_query& Q = queries[q];
    item_prefix[(size_t)q] = items;
    emit_prefix[(size_t)q] = slots;
    collect_prefix[(size_t)q] = collect_items;
    dq[(size_t)q] = DevQuery{RGPU_OP_AND, 0, (int32_t)dt.size(), 0};
    bool dead = false;
    for (int i = 0; i < Q.n_terms; ++i) dead = dead || terms[Q.first_term + i].state.doc_freq <= 0;
    if (dead) continue;  // PhraseWeight::create_scorer -> None (phrase_query.rs:275-283)
    std::vector<int> order((size_t)Q.n_terms);
    for (int i = 0; i < Q.n_terms; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return terms[Q.first_term + a].state.doc_freq < terms[Q.first_term + b].state.doc_freq; });
    for (int i : order) {  // the conjunction is driven by the rarest term (conjunction_scorer.rs:30)
      const rgpu_phrase_term& t = terms[Q.first_term + i];
      DevTerm d;
      rc = make_dev_term(seg, t.state, Q.weight, Q.sim_table, &d);
      if (rc != RGPU_OK) return rc;
      dt.push_back(d);
      PosTerm p{};
      p.pos_start_fp = (uint64_t)t.positions.pos_start_fp;
      p.total_term_freq = t.state.total_term_freq;
      // posting_reader.rs:1195-1203: fewer than 128 positions -> all VInts; exactly 128 -> one packed block, no trailing one
      p.last_pos_block_fp = t.state.total_term_freq < 128 ? t.positions.pos_start_fp
                            : (t.state.total_term_freq == 128 ? -1 : t.positions.pos_start_fp + t.positions.last_pos_block_offset);
      p.phrase_pos = t.position;
      p.query_ord = i;  // PhrasePositions::ord: the sloppy scorer keeps the query's order (phrase_query.rs:324-331 does not sort)
      p.same_as = i;    // ... and tells repeated terms by Term equality (repeating_terms, phrase_scorer.rs:909-931)
      for (int j = 0; j < i; ++j) {
        const rgpu_term_state& o = terms[Q.first_term + j].state;
        if (o.doc_start_fp == t.state.doc_start_fp && o.doc_freq == t.state.doc_freq && o.singleton_doc_id == t.state.singleton_doc_id &&
            o.total_term_freq == t.state.total_term_freq) { p.same_as = j; break; }
      }
      pt.push_back(p);
    }
    slops[(size_t)q] = Q.slop;
    limits[(size_t)q] = Q.next_limit == 0 ? 500000 /* searcher.rs:47 DEFAULT_DISMATCH_NEXT_LIMIT */
                                          : (Q.next_limit == RGPU_NEXT_LIMIT_ZERO ? 0 /* DefaultIndexSearcher::new(reader, Some(0)) */ : Q.next_limit);
    (Q.slop > 0 ? any_sloppy : any_exact) = true;
    if (Q.slop > 0) for (size_t i = pt.size() - (size_t)Q.n_terms; i < pt.size(); ++i) sloppy_rpts = sloppy_rpts || pt[i].same_as != pt[i].query_ord;
    dq[(size_t)q].n_terms = Q.n_terms;
    const DevTerm& lead = dt[(size_t)dq[(size_t)q].first_term];
    items += lead.nblocks == 0 ? 1 : (lead.nblocks + blocks_per_item - 1) / blocks_per_item;
    slots += ((int64_t)lead.df + 63) & ~(int64_t)63;  // (k_phrase_match_lanes: the 64 slots of a wavefront belong to one query)
    collect_items += ((int64_t)lead.df + PHRASE_COLLECT_CHUNK - 1) / PHRASE_COLLECT_CHUNK;  // the lead's doc_freq bounds the candidates
    any_cutoff = any_cutoff || (Q.slop > 0 && limits[(size_t)q] >= 0);
  }
  item_prefix[(size_t)n_queries] = items;
  emit_prefix[(size_t)n_queries] = slots;
  collect_prefix[(size_t)n_queries] = collect_items;
  HIP_TRY(c->host_api_hits.reserve((size_t)n_queries * (size_t)k, 0, stream));
  HIP_TRY(c->host_api_totals.reserve((size_t)n_queries, 0, stream));
  HIP_TRY(hipMemsetAsync(c->host_api_totals.p, 0, (size_t)n_queries * 8, stream));
  RGPU_LAUNCH(k_init_hits, dim3(wg_count(((size_t)n_queries * k + 255) / 256)), dim3(256), 0, stream, c->host_api_hits.p, (int64_t)n_queries, (int)k, (int)k, 0);
  if (items > 0) {
    SCRATCH_TAKE(c);
    Stager st(c);
    const size_t o_q = st.add((size_t)n_queries * sizeof(DevQuery));
    const size_t o_t = st.add(dt.size() * sizeof(DevTerm));
    const size_t o_pt = st.add(pt.size() * sizeof(PosTerm));
    const size_t o_ip = st.add((size_t)(n_queries + 1) * 8);
    const size_t o_ep = st.add((size_t)(n_queries + 1) * 8);
    const size_t o_sl = st.add((size_t)n_queries * 4);
    const size_t o_nl = st.add((size_t)n_queries * 4);
    const size_t o_gr = st.add((size_t)n_queries * sizeof(SloppyGroups));  // written by k_sloppy_groups
    const size_t o_cp = st.add((size_t)(n_queries + 1) * 8);                // the chunked collector's items
    const size_t o_ab = st.add((size_t)n_queries * 4);                      // k_phrase_cutoff's flags (zeroed by the copy)
    const size_t o_cut = st.add((size_t)n_queries * sizeof(PhraseCut));     // ... and what its chunked form reduces into
    std::vector<TermBitmap> clause_bitmaps;  // parallel to dt: the clauses behind a query's lead that have a doc bitmap
    if (bitmap_df != INT64_MAX && seg->bitmaps.size() > 0) {
      bool any = false;
      clause_bitmaps.assign(dt.size(), TermBitmap{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0});
      for (const DevQuery& q0 : dq) {
        for (int i = 1; i < q0.n_terms; ++i) {
          const DevTerm& t = dt[(size_t)(q0.first_term + i)];
          if (t.df < bitmap_df) continue;
          const BitmapInfo* bm = seg->bitmaps.find((int64_t)t.start_fp);
          if (!bm || !bm->usable || bm->df != t.df) continue;
          clause_bitmaps[(size_t)(q0.first_term + i)] = TermBitmap{bm->words, bm->ranks, bm->freqs, bm->ovf, bm->nib, bm->memb, bm->n_ovf, 0};
          any = true;
        }
      }
      if (!any) clause_bitmaps.clear();
    }
    const size_t o_bm = clause_bitmaps.empty() ? 0 : st.add(clause_bitmaps.size() * sizeof(TermBitmap));
    HIP_TRY(c->S->h_stage.reserve(st.used));
    HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
    std::memcpy(c->S->h_stage.p + o_q, dq.data(), (size_t)n_queries * sizeof(DevQuery));
    std::memcpy(c->S->h_stage.p + o_t, dt.data(), dt.size() * sizeof(DevTerm));
    std::memcpy(c->S->h_stage.p + o_pt, pt.data(), pt.size() * sizeof(PosTerm));
    std::memcpy(c->S->h_stage.p + o_sl, slops.data(), (size_t)n_queries * 4);
    std::memcpy(c->S->h_stage.p + o_nl, limits.data(), (size_t)n_queries * 4);
    std::memcpy(c->S->h_stage.p + o_ip, item_prefix.data(), (size_t)(n_queries + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_ep, emit_prefix.data(), (size_t)(n_queries + 1) * 8);
    std::memcpy(c->S->h_stage.p + o_cp, collect_prefix.data(), (size_t)(n_queries + 1) * 8);
    std::memset(c->S->h_stage.p + o_ab, 0, (size_t)n_queries * 4);
    for (int32_t q = 0; q < n_queries; ++q) reinterpret_cast<PhraseCut*>(c->S->h_stage.p + o_cut)[q] = PhraseCut{0x7fffffff, 0x7fffffff, 0ull};
    // "no repetition group" for every query (grp = -1): k_sloppy_groups only runs when some phrase repeats a term, k_sloppy_match
    // reads the region either way (the copy used to carry whatever the pinned buffer held)
    std::memset(c->S->h_stage.p + o_gr, 0xff, (size_t)n_queries * sizeof(SloppyGroups));
    if (!clause_bitmaps.empty()) std::memcpy(c->S->h_stage.p + o_bm, clause_bitmaps.data(), clause_bitmaps.size() * sizeof(TermBitmap));
    HIP_TRY(stage_h2d(c, st.used, stream));
    HIP_TRY(c->phrase_docs.reserve((size_t)slots + 64, 0, stream));
    HIP_TRY(c->phrase_keys.reserve((size_t)slots + 64, 0, stream));
    // (developer knob: RGPU_PHRASE_REDO_CAP=<n> shrinks the list of left-over candidates, so that a test reaches the pass that runs without it)
    int64_t redo_cap = PHRASE_REDO_LIST_CAP;
    if (const char* e = std::getenv("RGPU_PHRASE_REDO_CAP")) redo_cap = std::max<int64_t>(0, std::min<int64_t>(redo_cap, std::atoll(e)));
    redo_cap = std::min<int64_t>(slots, redo_cap);
    HIP_TRY(c->phrase_redo.reserve((size_t)redo_cap + 64, 0, stream));
    HIP_TRY(c->phrase_count.reserve((size_t)n_queries, 0, stream));
    HIP_TRY(hipMemsetAsync(c->phrase_count.p, 0, (size_t)n_queries * 8, stream));
    const int k_emit = std::min<int>(k, 64);  // the conjunction only emits candidates: its (empty) top-k lists are the narrow kind
    const bool chunked = k <= RGPU_PASS_K && collect_items > 0;  // (deeper pages: the one-wavefront collector's passes)
    HIP_TRY(c->S->d_partial_keys.reserve(std::max((size_t)items * (size_t)k_emit, chunked ? (size_t)collect_items * (size_t)k : (size_t)0), 0, stream));
    HIP_TRY(c->S->d_partial_counts.reserve((size_t)std::max(items, chunked ? collect_items : (int64_t)0), 0, stream));
    HIP_TRY(c->S->d_tau.reserve((size_t)n_queries, 0, stream));
    HIP_TRY(c->S->d_touched.reserve((size_t)n_queries * 2, 0, stream));
    HIP_TRY(hipMemsetAsync(c->S->d_tau.p, 0, (size_t)n_queries * 8, stream));
    HIP_TRY(hipMemsetAsync(c->S->d_touched.p, 0, (size_t)n_queries * 16, stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, 4 * sizeof(int), stream));  // [0]: error code, [3]: "a candidate needs the wide lists"
    const DevQuery* d_q = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
    const DevTerm* d_t = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
    const PosTerm* d_pt = reinterpret_cast<const PosTerm*>(c->S->d_stage.p + o_pt);
    const int64_t* d_ip = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_ip);
    const int64_t* d_ep = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_ep);
    const SegView sv = seg_view(seg);
    const bool legacy = seg->version < 1;
    {
      TimedLaunch tl(c, stream, "k_search_and(phrase candidates)", 0);
      const int xcd_chunk = and_xcd_chunk(seg);
      const unsigned grid = wg_count(and_grid((items + AND_WG_WAVES - 1) / AND_WG_WAVES, xcd_chunk));
      auto go = [&](auto kern) {
        RGPU_LAUNCH(kern, dim3(grid), dim3(AND_WG_THREADS), 0, stream, sv, d_q, d_t, d_ip, (int)n_queries, items, blocks_per_item, k_emit,
                           c->S->d_partial_keys.p, c->S->d_partial_counts.p, c->S->d_tau.p, c->S->d_touched.p, d_ep, c->phrase_count.p,
                           (void*)c->phrase_docs.p, (const unsigned long long*)nullptr, (const int32_t*)nullptr,
                           clause_bitmaps.empty() ? (const TermBitmap*)nullptr : reinterpret_cast<const TermBitmap*>(c->S->d_stage.p + o_bm), xcd_chunk);
      };
      if (legacy) go(k_search_and<true, false, false, false>); else go(k_search_and<false, false, false, false>);
    }
    const int32_t* d_sl = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_sl);
    const int32_t* d_nl = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_nl);
    SloppyGroups* d_gr = reinterpret_cast<SloppyGroups*>(c->S->d_stage.p + o_gr);
    if (slots > 0) {
      // One wavefront per candidate slot, first with the small position lists / pools (seven wavefronts per SIMD). A doc that holds
      // a term more often than those hold positions (rare: Rucene clamps freqs to 10) leaves PHRASE_REDO in its slot: one look at
      // the flag, then the wide instantiations over the marked candidates only.
      // (one wavefront per slot: at most PHRASE_LAUNCH_SLOTS slots per launch — a grid of more than 2^32 work-items is cut short)
      auto per_slot = [&](int64_t n, auto launch) {
        for (int64_t s0 = 0; s0 < n; s0 += PHRASE_LAUNCH_SLOTS) {
          const int64_t s1 = std::min(n, s0 + PHRASE_LAUNCH_SLOTS);
          launch(dim3(wg_count((s1 - s0 + WG_WAVES - 1) / WG_WAVES)), s0, s1);
        }
      };
      // (list: null = every slot of the launch; else the listed slots — the ones a 64-candidate kernel handed on)
      auto exact = [&](auto kern, const int64_t* list, int64_t n) {
        per_slot(n, [&](dim3 grid, int64_t s0, int64_t s1) {
          RGPU_LAUNCH(kern, grid, dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl, (int)n_queries, s1,
                      (int64_t)seg->pos_len, c->phrase_keys.p, c->d_err, c->d_err + 3, list, s0);
        });
      };
      auto sloppy = [&](auto kern, const int64_t* list, int64_t n) {
        per_slot(n, [&](dim3 grid, int64_t s0, int64_t s1) {
          RGPU_LAUNCH(kern, grid, dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl, d_gr, (int)n_queries, s1,
                      (int64_t)seg->pos_len, c->phrase_keys.p, c->d_err, c->d_err + 3, list, s0);
        });
      };
      auto sloppy_groups = [&]() {  // the repetition groups of each query's first candidate doc (phrases with a repeated term)
        TimedLaunch tl(c, stream, "k_sloppy_groups", 0);
        PhraseCut* d_cut0 = reinterpret_cast<PhraseCut*>(c->S->d_stage.p + o_cut);
        const int64_t* d_cp0 = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_cp);
        const int32_t* d_nl0 = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_nl);
        if (collect_items > 0)  // every sloppy query's smallest live candidate, chunk by chunk
          RGPU_LAUNCH(k_phrase_cutoff_items<2>, dim3(wg_count((collect_items + WG_WAVES - 1) / WG_WAVES)), dim3(WG_THREADS), 0, stream, d_cp0, d_ep,
                      c->phrase_count.p, c->phrase_keys.p, (const int32_t*)c->phrase_docs.p, d_sl, d_nl0, (int)n_queries, collect_items, d_cut0);
        const unsigned ggrid = wg_count((n_queries + WG_WAVES - 1) / WG_WAVES);
        auto go = [&](auto kern) {
          RGPU_LAUNCH(kern, dim3(ggrid), dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl,
                      (int)n_queries, (int64_t)seg->pos_len, d_gr, c->d_err, (const PhraseCut*)d_cut0);
        };
        if (legacy) go(k_sloppy_groups<true>); else go(k_sloppy_groups<false>);
      };
      const int64_t* const all_slots = nullptr;
      const int64_t groups = slots / 64;
      const dim3 lanes_grid(wg_count((groups + WG_WAVES - 1) / WG_WAVES));
      // ---- first pass. Packed (.doc version 1) segments: 64 candidates per wavefront; what those kernels hand on is listed.
      // Legacy segments (the packed streams of a block are laid out differently): one candidate per wavefront throughout.
      if (any_exact) {
        if (legacy) {
          TimedLaunch tl(c, stream, "k_phrase_match", 0);
          exact(k_phrase_match<true, PHRASE_SMALL_CAP, false>, all_slots, slots);
        } else {
          TimedLaunch tl(c, stream, "k_phrase_match_lanes", 0);
          RGPU_LAUNCH(k_phrase_match_lanes, lanes_grid, dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl,
                      (int)n_queries, groups, (int64_t)seg->pos_len, c->phrase_keys.p, c->d_err + 3, c->phrase_redo.p, (int)redo_cap, c->d_err + 2);
        }
      }
      if (any_sloppy) {
        if (legacy) {
          sloppy_groups();
          TimedLaunch tl(c, stream, "k_sloppy_match", 0);
          sloppy(k_sloppy_match<true, SLOPPY_SMALL_POOL, false>, all_slots, slots);
        } else {
          if (sloppy_rpts) sloppy_groups();  // the repetition groups: k_sloppy_rpt_lanes (and what it hands on) needs them
          {
            TimedLaunch tl(c, stream, "k_sloppy_match_lanes", 0);
            RGPU_LAUNCH(k_sloppy_match_lanes, lanes_grid, dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl,
                        (int)n_queries, groups, (int64_t)seg->pos_len, c->phrase_keys.p, c->d_err + 3, c->phrase_redo.p, (int)redo_cap, c->d_err + 2,
                        sloppy_rpts ? 1 : 0);
          }
          if (sloppy_rpts) {  // phrases that repeat a term: the scorer's repeats machinery per lane
            TimedLaunch tl(c, stream, "k_sloppy_rpt_lanes", 0);
            RGPU_LAUNCH(k_sloppy_rpt_lanes, lanes_grid, dim3(WG_THREADS), 0, stream, sv, d_q, d_t, d_pt, d_ep, c->phrase_count.p, c->phrase_docs.p, d_sl,
                        d_gr, (int)n_queries, groups, (int64_t)seg->pos_len, c->phrase_keys.p, c->d_err + 3, c->phrase_redo.p, (int)redo_cap, c->d_err + 2);
          }
        }
      }
      // ---- which candidates wait for another pass (bits PHRASE_REDO_*): one look per stage that can raise one
      int listed = 0, redo = 0;  // d_err[2]: slots on the list, d_err[3]: the bits
      auto redo_bits = [&]() -> int32_t {
        int two[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(two, c->d_err + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        listed = two[0];
        redo = two[1];
        return RGPU_OK;
      };
      rc = redo_bits();
      if (rc != RGPU_OK) return rc;
      // the one-candidate kernels over the listed slots (or, should the list have overflowed, over every slot)
      const bool by_list = !legacy && (int64_t)listed <= redo_cap;
      const int64_t* const left = by_list ? (const int64_t*)c->phrase_redo.p : all_slots;
      const int64_t n_left = by_list ? (int64_t)listed : slots;
      if (HostClock::on() && (redo & (PHRASE_REDO_LANES | PHRASE_REDO_SLOPPY_LANES)))
        std::fprintf(stderr, "[phrase] %d of %lld candidate slots left for the one-candidate kernels\n", listed, (long long)slots);
      if (redo & PHRASE_REDO_LANES) {
        TimedLaunch tl(c, stream, "k_phrase_match(left by the 64-candidate kernel)", 0);
        exact(k_phrase_match<false, PHRASE_SMALL_CAP, true>, left, n_left);
      }
      if (redo & PHRASE_REDO_SLOPPY_LANES) {
        TimedLaunch tl(c, stream, "k_sloppy_match(left by the 64-candidate kernel)", 0);  // (the groups are there: computed in front of the 64-candidate kernels)
        sloppy(k_sloppy_match<false, SLOPPY_SMALL_POOL, true>, left, n_left);
      }
      if (redo & (PHRASE_REDO_LANES | PHRASE_REDO_SLOPPY_LANES)) {
        rc = redo_bits();
        if (rc != RGPU_OK) return rc;
      }
      if (redo & PHRASE_REDO_WIDE) {
        TimedLaunch tl(c, stream, "k_phrase_match(wide lists)", 0);
        if (legacy) exact(k_phrase_match<true, PHRASE_LIST_CAP, true>, left, n_left); else exact(k_phrase_match<false, PHRASE_LIST_CAP, true>, left, n_left);
      }
      if (redo & PHRASE_REDO_SLOPPY) {
        TimedLaunch tl(c, stream, "k_sloppy_match(wide pool)", 0);
        if (legacy) sloppy(k_sloppy_match<true, SLOPPY_POOL, true>, left, n_left); else sloppy(k_sloppy_match<false, SLOPPY_POOL, true>, left, n_left);
      }
    }
    if (chunked) {
      const int64_t* d_cp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_cp);
      int32_t* d_ab = reinterpret_cast<int32_t*>(c->S->d_stage.p + o_ab);
      {
      TimedLaunch tl(c, stream, "k_phrase_collect", 0);
      const unsigned grid = wg_count((collect_items + WG_WAVES - 1) / WG_WAVES);
      if (any_cutoff) {  // the two-phase rule's cut-off, chunk by chunk: first collected doc, candidates in front of it, the decision
        PhraseCut* d_cut = reinterpret_cast<PhraseCut*>(c->S->d_stage.p + o_cut);
        RGPU_LAUNCH(k_phrase_cutoff_items<0>, dim3(grid), dim3(WG_THREADS), 0, stream, d_cp, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)c->phrase_docs.p, d_sl, d_nl, (int)n_queries, collect_items, d_cut);
        RGPU_LAUNCH(k_phrase_cutoff_items<1>, dim3(grid), dim3(WG_THREADS), 0, stream, d_cp, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)c->phrase_docs.p, d_sl, d_nl, (int)n_queries, collect_items, d_cut);
        RGPU_LAUNCH(k_phrase_cutoff_decide, dim3(wg_count((n_queries + 255) / 256)), dim3(256), 0, stream, c->phrase_count.p, d_sl, d_nl, (int)n_queries,
                           d_cut, d_ab);
      }
      if (k > 64)
        RGPU_LAUNCH(k_phrase_collect_items<true>, dim3(grid), dim3(WG_THREADS), 0, stream, d_cp, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)d_ab, (int)n_queries, collect_items, (int)k, c->S->d_partial_keys.p, c->S->d_partial_counts.p);
      else
        RGPU_LAUNCH(k_phrase_collect_items<false>, dim3(grid), dim3(WG_THREADS), 0, stream, d_cp, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)d_ab, (int)n_queries, collect_items, (int)k, c->S->d_partial_keys.p, c->S->d_partial_counts.p);
      }
      if (k > 64) launch_merge<true>(c, stream, n_queries, k, d_cp, seg->doc_base, c->host_api_hits.p, c->host_api_totals.p);
      else launch_merge<false>(c, stream, n_queries, k, d_cp, seg->doc_base, c->host_api_hits.p, c->host_api_totals.p);
    } else {
      TimedLaunch tl(c, stream, "k_phrase_collect", 0);
      const unsigned grid = wg_count((n_queries + WG_WAVES - 1) / WG_WAVES);
      if (k > 64)
        RGPU_LAUNCH(k_phrase_collect<true>, dim3(grid), dim3(WG_THREADS), 0, stream, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)c->phrase_docs.p, d_sl, d_nl, (int)n_queries, (int)k, seg->doc_base, c->host_api_hits.p, c->host_api_totals.p);
      else
        RGPU_LAUNCH(k_phrase_collect<false>, dim3(grid), dim3(WG_THREADS), 0, stream, d_ep, c->phrase_count.p, c->phrase_keys.p,
                           (const int32_t*)c->phrase_docs.p, d_sl, d_nl, (int)n_queries, (int)k, seg->doc_base, c->host_api_hits.p, c->host_api_totals.p);
    }
    HIP_TRY(launch_status());
  }
  int err = 0;
  hipError_t e0 = items > 0 ? hipMemcpyAsync(&err, c->d_err, sizeof(int), hipMemcpyDeviceToHost, stream) : hipSuccess;
  hipError_t e1 = hipMemcpyAsync(hits_out, c->host_api_hits.p, (size_t)n_queries * (size_t)k * sizeof(HitOut), hipMemcpyDeviceToHost, stream);
  hipError_t e2 = hipMemcpyAsync(total_hits_out, c->host_api_totals.p, (size_t)n_queries * 8, hipMemcpyDeviceToHost, stream);
  hipError_t e3 = hipStreamSynchronize(stream);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(RGPU_ERR_RUNTIME, "device to host copy failed");
  if (err != 0)
    return fail(err, err == RGPU_ERR_UNSUPPORTED ? "a doc holds one of an exact phrase's terms more than 1024 times (a sloppy phrase's terms: more than 2048 times in all)"
                                                 : (err == RGPU_ERR_ILLEGAL_STATE ? "internal: a conjunction match was not found again" : "corrupt position data in .pos"));
  return RGPU_OK;
}

// ---- QueryRescorer ------------------------------------------------------------------------------------------------------------
static_assert(sizeof(RescoreParams) == sizeof(rgpu_rescore_request), "RescoreParams mirrors rgpu_rescore_request");
extern "C" int32_t rgpu_rescore_batch(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                                      int32_t n_terms_total, const rgpu_rescore_request* requests, int32_t k, rgpu_hit* hits_inout,
                                      int32_t finish) {
  if (!seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !requests || !hits_inout)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_PASS_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "rescoring: k must be in 1..128");
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = c->stream;
  std::vector<const rgpu_term_state*> ptrs;
  std::vector<RescoreParams> rp((size_t)n_queries);
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_query& Q = queries[q];
    const int qop = Q.op & 0xff;
    if (qop < RGPU_OP_TERM || qop > RGPU_OP_OR || (Q.op >> 8) > 1 || Q.n_must_not != 0)
      return fail(RGPU_ERR_UNSUPPORTED, "rescore queries are TERM, all-MUST or all-SHOULD term queries");
    if (Q.n_terms < 1 || Q.n_terms > RGPU_MAX_QUERY_TERMS || (qop == RGPU_OP_TERM && Q.n_terms != 1)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad clause count");
    if (Q.first_term < 0 || (int64_t)Q.first_term + Q.n_terms > (int64_t)n_terms_total) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "clause range outside terms[]");
    const rgpu_rescore_request& r = requests[q];
    if (r.mode < RGPU_RESCORE_AVG || r.mode > RGPU_RESCORE_MULTIPLY || r.window_size < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad rescore request");
    rp[(size_t)q] = RescoreParams{r.query_weight, r.rescore_weight, r.mode, std::min(std::min(r.window_size, k), RGPU_PASS_K)};
    for (int i = 0; i < Q.n_terms; ++i) {
      const rgpu_query_term& t = terms[Q.first_term + i];
      if (t.sim_table < 0 || t.sim_table >= c->n_sim_tables) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "unknown sim_table handle");
      if (t.state.doc_freq > 0) ptrs.push_back(&t.state);
    }
  }
  int32_t rc = prepare_terms_locked(seg, ptrs.data(), ptrs.size());
  if (rc != RGPU_OK) return rc;
  std::vector<DevQuery> dq((size_t)n_queries);
  std::vector<DevTerm> dt;
  std::vector<DevTerm> mine;
  for (int32_t q = 0; q < n_queries; ++q) {
    const rgpu_query& Q = queries[q];
    const int qop = Q.op & 0xff;
    mine.clear();
    bool dead = false;
    for (int i = 0; i < Q.n_terms; ++i) {
      const rgpu_query_term& t = terms[Q.first_term + i];
      if (t.state.doc_freq <= 0) { if (qop != RGPU_OP_OR) dead = true; continue; }  // create_scorer -> None: the query matches nothing here
      DevTerm d;
      rc = make_dev_term(seg, t.state, t.weight, t.sim_table, &d);
      if (rc != RGPU_OK) return rc;
      mine.push_back(d);
    }
    if (dead) mine.clear();
    if (qop == RGPU_OP_AND) std::stable_sort(mine.begin(), mine.end(), [](const DevTerm& a, const DevTerm& b) { return a.df < b.df; });
    dq[(size_t)q] = DevQuery{qop, (int32_t)mine.size(), (int32_t)dt.size(), 0};
    for (auto& m : mine) dt.push_back(m);
  }
  SCRATCH_TAKE(c);
  Stager st(c);
  const size_t o_q = st.add((size_t)n_queries * sizeof(DevQuery));
  const size_t o_t = st.add(std::max<size_t>(1, dt.size()) * sizeof(DevTerm));
  const size_t o_r = st.add((size_t)n_queries * sizeof(RescoreParams));
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
  std::memcpy(c->S->h_stage.p + o_q, dq.data(), (size_t)n_queries * sizeof(DevQuery));
  if (!dt.empty()) std::memcpy(c->S->h_stage.p + o_t, dt.data(), dt.size() * sizeof(DevTerm));
  std::memcpy(c->S->h_stage.p + o_r, rp.data(), (size_t)n_queries * sizeof(RescoreParams));
  HIP_TRY(stage_h2d(c, st.used, stream));
  const size_t n_hits = (size_t)n_queries * (size_t)k;
  HIP_TRY(c->host_api_hits.reserve(n_hits, 0, stream));
  HIP_TRY(hipMemcpyAsync(c->host_api_hits.p, hits_inout, n_hits * sizeof(HitOut), hipMemcpyHostToDevice, stream));
  const DevQuery* d_q = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
  const DevTerm* d_t = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
  RescoreParams* d_r = reinterpret_cast<RescoreParams*>(c->S->d_stage.p + o_r);
  {
    TimedLaunch tl(c, stream, "k_rescore", 0);
    const unsigned grid = wg_count((n_hits + WG_WAVES - 1) / WG_WAVES);
    if (seg->version < 1) RGPU_LAUNCH(k_rescore<true>, dim3(grid), dim3(WG_THREADS), 0, stream, seg_view(seg), d_q, d_t, d_r, (int)n_queries, (int)k, c->host_api_hits.p, finish ? 1 : 0);
    else RGPU_LAUNCH(k_rescore<false>, dim3(grid), dim3(WG_THREADS), 0, stream, seg_view(seg), d_q, d_t, d_r, (int)n_queries, (int)k, c->host_api_hits.p, finish ? 1 : 0);
  }
  if (finish) {
    TimedLaunch tl(c, stream, "k_rescore_sort", 0);
    RGPU_LAUNCH(k_rescore_sort, dim3(wg_count((n_queries + WG_WAVES - 1) / WG_WAVES)), dim3(WG_THREADS), 0, stream, d_r, (int)n_queries, (int)k, c->host_api_hits.p);
  }
  HIP_TRY(launch_status());
  HIP_TRY(hipMemcpyAsync(hits_inout, c->host_api_hits.p, n_hits * sizeof(HitOut), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return RGPU_OK;
}

// ---- segment-sharded search: RCCL all-gather of per-shard top-k + device merge -------------------------------------------
constexpr int N_COMM_SLOTS = 4;
struct CommSlot {
  // One record (record_bytes: hits, counts, status word) per rank. The all-gather is IN PLACE: this rank's search writes its
  // record straight into its own region, recv + rank * record (ncclAllGather with sendbuff == recvbuff + rank * count moves
  // nothing locally: no send buffer, no device copy of the rank's own record).
  DevVec<uint8_t> recv;
  size_t record = 0;           // the record size the status words below were prepared for
  bool status_zero = false;    // this rank's status word already reads RGPU_OK (zeroed with the buffer, never dirtied since): a
                               // successful search then needs no extra launch to say so
  hipEvent_t done = nullptr;
  bool busy = false;
};
struct rgpu_comm {
  rgpu_ctx* ctx = nullptr;
  ncclComm_t nccl = nullptr;
  int n_ranks = 1, rank = 0;
  CommSlot slots[N_COMM_SLOTS];
  int next = 0;
  // collectives of one communicator must not run side by side: when consecutive calls come on different streams, the
  // later all-gather waits for the earlier one (an event), while the searches in front of them still overlap
  hipStream_t last_stream = nullptr;
  hipEvent_t last_collective = nullptr;
  bool have_last = false;
  CommSlot* last_slot = nullptr;  // the most recent batch's records (rgpu_comm_status)
  int32_t last_queries = 0, last_k = 0;
  int64_t gathers_issued = 0;     // ncclAllGather calls enqueued on this communicator (rgpu_comm_gathers_issued)
};
#define NCCL_TRY(expr)                                                                                          \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) return fail(RGPU_ERR_RUNTIME, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
  } while (0)
static_assert(RGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rgpu_comm ids are ncclUniqueIds");

extern "C" int32_t rgpu_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "id_out is null");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  std::memcpy(id_out, id.internal, RGPU_COMM_ID_BYTES);
  return RGPU_OK;
}

extern "C" int32_t rgpu_comm_init(rgpu_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t* id_bytes, rgpu_comm** out) {
  if (!out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out_comm is null");
  *out = nullptr;
  if (!c || !id_bytes) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "rank outside [0, n_ranks)");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, RGPU_COMM_ID_BYTES);
  auto comm = std::make_unique<rgpu_comm>();
  comm->ctx = c;
  comm->n_ranks = n_ranks;
  comm->rank = rank;
  NCCL_TRY(ncclCommInitRank(&comm->nccl, n_ranks, id, rank));
  *out = comm.release();
  return RGPU_OK;
}

// One process, one thread, n contexts (Rucene itself is one process: search_parallel hands leaves to threads,
// searcher.rs:527-630): ncclCommInitRank blocks until every rank of the id has joined, so n of them issued one after the
// other from one thread would never return — they have to sit inside one ncclGroupStart / ncclGroupEnd.
extern "C" int32_t rgpu_comm_init_all(rgpu_ctx* const* ctxs, int32_t n, rgpu_comm** out_comms) {
  if (!ctxs || !out_comms || n < 1) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  for (int32_t r = 0; r < n; ++r) {
    out_comms[r] = nullptr;
    if (!ctxs[r]) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null context");
    for (int32_t j = 0; j < r; ++j)
      if (ctxs[j] == ctxs[r] || ctxs[j]->device == ctxs[r]->device) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "every rank needs its own context on its own device");
  }
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  std::vector<std::unique_ptr<rgpu_comm>> comms;
  for (int32_t r = 0; r < n; ++r) {
    comms.emplace_back(new rgpu_comm());
    comms.back()->ctx = ctxs[r];
    comms.back()->n_ranks = n;
    comms.back()->rank = r;
  }
  ncclResult_t res = ncclGroupStart();
  for (int32_t r = 0; r < n && res == ncclSuccess; ++r) {
    if (hipSetDevice(ctxs[r]->device) != hipSuccess) { res = ncclUnhandledCudaError; break; }
    res = ncclCommInitRank(&comms[(size_t)r]->nccl, n, id, r);
  }
  const ncclResult_t end = ncclGroupEnd();
  if (res == ncclSuccess) res = end;
  if (res != ncclSuccess) {
    for (auto& cm : comms) if (cm->nccl) (void)ncclCommAbort(cm->nccl);
    return fail(RGPU_ERR_RUNTIME, std::string("ncclCommInitRank (grouped): ") + ncclGetErrorString(res));
  }
  for (int32_t r = 0; r < n; ++r) out_comms[r] = comms[(size_t)r].release();
  return RGPU_OK;
}

extern "C" void rgpu_comm_destroy(rgpu_comm* comm) {
  if (!comm) return;
  (void)hipSetDevice(comm->ctx->device);
  for (auto& sl : comm->slots) {
    if (sl.busy) (void)hipEventSynchronize(sl.done);
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl.recv.release();
  }
  if (comm->last_collective) (void)hipEventDestroy(comm->last_collective);
  if (comm->nccl) (void)ncclCommDestroy(comm->nccl);
  delete comm;
}

// ---- a shard's record: what one rank contributes to the all-gather ---------------------------------------------------------
// [n_queries x k rgpu_hit][n_queries x int64 hit count][int64 status] — the status word is the rank's rgpu_status for the
// batch: a rank whose local search failed still takes part in the collective (with empty rows), so that the other ranks
// neither hang in the all-gather nor silently merge a stale record; every rank can read everybody's status afterwards.
static size_t record_hits_bytes(int32_t n_queries, int32_t k) { return (size_t)n_queries * (size_t)k * sizeof(HitOut); }
static size_t record_bytes(int32_t n_queries, int32_t k) { return record_hits_bytes(n_queries, k) + (size_t)n_queries * 8 + 8; }
extern "C" int64_t rgpu_record_bytes(int32_t n_queries, int32_t k) {
  if (n_queries <= 0 || k <= 0) return 0;
  return (int64_t)record_bytes(n_queries, k);
}

__global__ void k_set_i64(int64_t* p, int64_t v) { *p = v; }

// local search of one shard straight into a record (device memory, record_bytes long); enqueue-only. The call's own status
// is returned AND left in the record's status word; on failure the record holds empty rows.
// `status_zero` (in / out, may be null): the record's status word is known to hold RGPU_OK already — a successful search then
// leaves it alone (one launch less per batch on the serving path); a failure writes it and clears the flag.
// `defer`: >= 10-clause disjunctions leave their hand-back flags for settle_pending (the caller runs it before the record is
// read by anybody: before the collective) — the one-process form enqueues every shard's search before it waits for any.
// A uniform batch that is still to be planned (the fused plan + search entry points): n_queries x n_clauses flat-table ids
struct UniformBatch {
  rgpu_planner* planner;
  int32_t op, n_clauses;
  const int64_t* ids;
};
static int32_t search_uniform_locked(rgpu_segment* seg, const UniformBatch& ub, int32_t n_queries, int32_t k, HitOut* hits_dev, int64_t* totals_dev,
                                     hipStream_t stream);
static int32_t search_into_record(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                                  int32_t n_terms_total, int32_t k, uint8_t* record, hipStream_t s, bool* status_zero = nullptr, bool defer = false,
                                  const UniformBatch* uniform = nullptr) {
  const size_t hits_bytes = record_hits_bytes(n_queries, k);
  seg->ctx->defer_or = defer;
  int32_t rc = uniform ? search_uniform_locked(seg, *uniform, n_queries, k, (HitOut*)record, (int64_t*)(record + hits_bytes), s)
                       : search_impl(seg, queries, n_queries, terms, n_terms_total, k, (HitOut*)record, (int64_t*)(record + hits_bytes), s);
  seg->ctx->defer_or = false;
  std::string why = rc == RGPU_OK ? std::string() : g_last_error;
  // No early return below: the status word is written whatever else fails (a peer that merged a record without one would
  // take stale rows for this shard's answer), and the first error is the one reported.
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == RGPU_OK) { rc = RGPU_ERR_RUNTIME; why = std::string(what) + ": " + hipGetErrorString(e); }
  };
  if (rc != RGPU_OK) {  // whatever was enqueued before the failure is overwritten behind it on the same stream
    note(hipMemsetAsync(record + hits_bytes, 0, (size_t)n_queries * 8, s), "hipMemsetAsync(record counts)");
    RGPU_LAUNCH(k_init_hits, dim3(wg_count(((size_t)n_queries * k + 255) / 256)), dim3(256), 0, s, (HitOut*)record, (int64_t)n_queries, (int)k, (int)k, 0);
    note(launch_status(), "k_init_hits");
  }
  if (rc == RGPU_OK && status_zero != nullptr && *status_zero) return RGPU_OK;  // the word says RGPU_OK already
  if (status_zero != nullptr) *status_zero = false;  // (set again by the caller once it has zeroed the word)
  RGPU_LAUNCH(k_set_i64, dim3(1), dim3(1), 0, s, (int64_t*)(record + hits_bytes + (size_t)n_queries * 8), (int64_t)rc);
  const hipError_t e_status = launch_status();
  if (e_status != hipSuccess) {  // the launch itself was refused: put the word there with a copy from the host instead
    const int64_t word = rc != RGPU_OK ? (int64_t)rc : (int64_t)RGPU_ERR_RUNTIME;
    (void)hipMemcpyAsync(record + hits_bytes + (size_t)n_queries * 8, &word, 8, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);  // (`word` lives on this frame)
    note(e_status, "k_set_i64");
  }
  if (rc != RGPU_OK) return fail(rc, why);
  return RGPU_OK;
}

extern "C" int32_t rgpu_search_batch_record_device(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries,
                                                   const rgpu_query_term* terms, int32_t n_terms_total, int32_t k, void* record_dev,
                                                   void* hip_stream) {
  if (!seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !record_dev) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  std::lock_guard<std::mutex> g(seg->ctx->mu);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : seg->ctx->stream;
  return search_into_record(seg, queries, n_queries, terms, n_terms_total, k, (uint8_t*)record_dev, s);
}

// the canonical k-way merge over `n_ranks` gathered records (k_merge_lists reading them in place, with the record stride)
static int32_t merge_records(rgpu_ctx* c, const uint8_t* records, int32_t n_ranks, int32_t n_queries, int32_t k, HitOut* hits_dev,
                             int64_t* totals_dev, hipStream_t s) {
  const size_t hits_bytes = record_hits_bytes(n_queries, k), record = record_bytes(n_queries, k);
  {
    TimedLaunch tl(c, s, "k_merge_lists", 0);
    const unsigned grid = wg_count((n_queries + WG_WAVES - 1) / WG_WAVES);
    auto go = [&](auto kern) {
      RGPU_LAUNCH(kern, dim3(grid), dim3(WG_THREADS), 0, s, (const HitOut*)records, (const int64_t*)(records + hits_bytes),
                         (int64_t)(record / sizeof(HitOut)), (int64_t)(record / 8), n_ranks, n_queries, (int)k, hits_dev, totals_dev);
    };
    if (k > 64) go(k_merge_lists<true>); else go(k_merge_lists<false>);
  }
  HIP_TRY(launch_status());
  return RGPU_OK;
}

extern "C" int32_t rgpu_merge_records_device(rgpu_ctx* c, const void* records_dev, int32_t n_ranks, int32_t n_queries, int32_t k,
                                             void* hits_out_dev, void* totals_out_dev, void* hip_stream) {
  if (!c || !records_dev || !hits_out_dev || !totals_out_dev || n_ranks <= 0 || n_queries <= 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  return merge_records(c, (const uint8_t*)records_dev, n_ranks, n_queries, k, (HitOut*)hits_out_dev, (int64_t*)totals_out_dev,
                       hip_stream ? (hipStream_t)hip_stream : c->stream);
}

// ---- the three phases of a sharded batch (context mutex held by the caller) -------------------------------------------------
struct ShardedCall {
  CommSlot* sl = nullptr;
  size_t record = 0;
  int32_t local_rc = RGPU_OK;
  std::string local_why;
  int32_t pre_rc = RGPU_OK;  // a failure in front of the collective that did not keep this rank from joining it
  std::string pre_why;
};
// phase 0: the slot and its buffers. The ONE thing that can fail here and leave the communicator out of step with its peers
// is the allocation of the record buffer itself (nowhere to gather into): rgpu_comm_reserve does it at start-up, so a serving
// host never allocates on the data path. Everything else that goes wrong (a wait on the slot's previous batch, the status
// word's memset) is remembered in call->pre_rc and reported AFTER the collective has been joined.
static int32_t sharded_reserve(rgpu_comm* comm, int32_t n_queries, int32_t k, hipStream_t s, ShardedCall* call) {
  CommSlot& sl = comm->slots[comm->next];
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && call->pre_rc == RGPU_OK) { call->pre_rc = RGPU_ERR_RUNTIME; call->pre_why = std::string(what) + ": " + hipGetErrorString(e); }
  };
  if (sl.busy) { note(hipEventSynchronize(sl.done), "hipEventSynchronize(slot)"); sl.busy = false; }
  const size_t record = record_bytes(n_queries, k);
  const uint8_t* before = sl.recv.p;
  HIP_TRY(sl.recv.reserve(record * (size_t)comm->n_ranks, 0, s));
  if (sl.recv.p != before || sl.record != record || !sl.status_zero) {
    // this rank's status word, zeroed once per (buffer, record size): a batch that succeeds never touches it again
    const hipError_t e = hipMemsetAsync(sl.recv.p + (size_t)comm->rank * record + record - 8, 0, 8, s);
    sl.record = record;
    sl.status_zero = e == hipSuccess;  // not zeroed: search_into_record writes the word with a launch instead
  }
  call->sl = &sl;
  call->record = record;
  return RGPU_OK;
}
// phase 1: from here on this rank WILL enqueue the collective, whatever its local search does (search_into_record leaves the
// status in the record)
static void sharded_local(rgpu_comm* comm, rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                          int32_t n_terms_total, int32_t k, hipStream_t s, ShardedCall* call, bool defer = false, const UniformBatch* uniform = nullptr) {
  comm->next = (comm->next + 1) % N_COMM_SLOTS;
  call->local_rc = search_into_record(seg, queries, n_queries, terms, n_terms_total, k, call->sl->recv.p + (size_t)comm->rank * call->record, s,
                                      &call->sl->status_zero, defer, uniform);
  if (call->local_rc != RGPU_OK) call->local_why = g_last_error;
}
// phase 1b (after a deferred sharded_local): whatever the shard's disjunctions still have to run again runs now, before the
// record is gathered; a failure becomes the shard's status like a failure of the search itself
static void sharded_settle(rgpu_comm* comm, int32_t n_queries, int32_t k, hipStream_t s, ShardedCall* call) {
  const int32_t rc = settle_pending(comm->ctx);
  if (rc == RGPU_OK || call->local_rc != RGPU_OK) return;
  call->local_rc = rc;
  call->local_why = g_last_error;
  uint8_t* record = call->sl->recv.p + (size_t)comm->rank * call->record;
  const size_t hits_bytes = record_hits_bytes(n_queries, k);
  call->sl->status_zero = false;
  (void)hipMemsetAsync(record + hits_bytes, 0, (size_t)n_queries * 8, s);
  RGPU_LAUNCH(k_init_hits, dim3(wg_count(((size_t)n_queries * k + 255) / 256)), dim3(256), 0, s, (HitOut*)record, (int64_t)n_queries, (int)k, (int)k, 0);
  RGPU_LAUNCH(k_set_i64, dim3(1), dim3(1), 0, s, (int64_t*)(record + hits_bytes + (size_t)n_queries * 8), (int64_t)rc);
  (void)launch_status();
}
static int32_t sharded_gather(rgpu_comm* comm, hipStream_t s, ShardedCall& call) {
  // A communicator of one rank has nothing to gather: its record is already where the merge reads it. (Before round 5 this
  // path cost a device copy of the record, a one-thread launch for the status word and an event wait between consecutive
  // batches on different streams: 0.157 ms per 1024-query TERM step against 0.093 for the local search.)
  // rgpu_config.comm_force_gather issues the in-place all-gather all the same: the N > 1 path on a one-GPU box.
  if (comm->n_ranks == 1 && comm->ctx->cfg.comm_force_gather == 0) return RGPU_OK;
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && call.pre_rc == RGPU_OK) { call.pre_rc = RGPU_ERR_RUNTIME; call.pre_why = std::string(what) + ": " + hipGetErrorString(e); }
    return e;
  };
  // collectives of one communicator run one after the other; when the stream changed, the later one waits for the earlier
  // one's event. If that wait cannot be enqueued the host waits instead — this rank still joins the collective (its peers
  // are in it already): nothing between sharded_reserve and ncclAllGather returns
  if (comm->have_last && comm->last_stream != s)
    if (note(hipStreamWaitEvent(s, comm->last_collective, 0), "hipStreamWaitEvent(last collective)") != hipSuccess)
      (void)hipEventSynchronize(comm->last_collective);
  uint8_t* const mine = call.sl->recv.p + (size_t)comm->rank * call.record;  // in place: sendbuff == recvbuff + rank * count
  NCCL_TRY(ncclAllGather(mine, call.sl->recv.p, call.record, ncclInt8, comm->nccl, s));
  comm->gathers_issued++;
  if (!comm->last_collective) (void)note(hipEventCreateWithFlags(&comm->last_collective, hipEventDisableTiming), "hipEventCreate(last collective)");
  if (comm->last_collective && note(hipEventRecord(comm->last_collective, s), "hipEventRecord(last collective)") == hipSuccess) {
    comm->last_stream = s;
    comm->have_last = true;
  } else {  // no event to order the next collective behind this one: the host waits for this one now
    (void)hipStreamSynchronize(s);
    comm->have_last = false;
  }
  return RGPU_OK;
}
static int32_t sharded_merge(rgpu_comm* comm, int32_t n_queries, int32_t k, void* hits_dev, void* totals_dev, hipStream_t s, const ShardedCall& call) {
  int32_t rc = merge_records(comm->ctx, call.sl->recv.p, comm->n_ranks, n_queries, k, (HitOut*)hits_dev, (int64_t*)totals_dev, s);
  if (rc != RGPU_OK) return rc;
  if (!call.sl->done) HIP_TRY(hipEventCreateWithFlags(&call.sl->done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(call.sl->done, s));
  call.sl->busy = true;
  comm->last_slot = call.sl;
  comm->last_queries = n_queries;
  comm->last_k = k;
  return RGPU_OK;
}

extern "C" int32_t rgpu_search_batch_sharded(rgpu_comm* comm, rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries,
                                             const rgpu_query_term* terms, int32_t n_terms_total, int32_t k, void* hits_dev,
                                             void* totals_dev, void* hip_stream) {
  if (!comm || !seg || !queries || n_queries <= 0 || !terms || n_terms_total <= 0 || !hits_dev || !totals_dev)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (seg->ctx != comm->ctx) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "segment and communicator belong to different contexts");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  rgpu_ctx* c = comm->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  ShardedCall call;
  int32_t rc = sharded_reserve(comm, n_queries, k, s, &call);
  if (rc != RGPU_OK) return rc;
  sharded_local(comm, seg, queries, n_queries, terms, n_terms_total, k, s, &call);
  rc = sharded_gather(comm, s, call);
  if (rc != RGPU_OK) return rc;
  rc = sharded_merge(comm, n_queries, k, hits_dev, totals_dev, s, call);
  if (rc != RGPU_OK) return rc;
  // the collective has been honoured; now this rank's own failure, if any, is reported (its peers see it in the status words)
  if (call.local_rc != RGPU_OK) return fail(call.local_rc, call.local_why);
  if (call.pre_rc != RGPU_OK) return fail(call.pre_rc, call.pre_why);
  return RGPU_OK;
}

// Start-up sizing: every slot's gather buffer for batches of up to n_queries x k, allocated now — the data path then never
// allocates, and the one failure that would leave a rank out of a collective (no buffer to gather into) cannot happen there.
// Not a collective; every rank calls it with the shapes it will serve.
extern "C" int32_t rgpu_comm_reserve(rgpu_comm* comm, int32_t n_queries, int32_t k) {
  if (!comm || n_queries <= 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  rgpu_ctx* c = comm->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = record_bytes(n_queries, k) * (size_t)comm->n_ranks;
  for (auto& sl : comm->slots) {
    if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.done)); sl.busy = false; }
    const uint8_t* before = sl.recv.p;
    HIP_TRY(sl.recv.reserve(bytes, 0, c->stream));
    if (sl.recv.p != before) sl.status_zero = false;
    if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  }
  if (!comm->last_collective) HIP_TRY(hipEventCreateWithFlags(&comm->last_collective, hipEventDisableTiming));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RGPU_OK;
}

// ncclAllGather calls this communicator has enqueued so far (a communicator of one rank issues none unless
// rgpu_config.comm_force_gather is set): tests and bench.py assert that the collective really ran.
extern "C" int64_t rgpu_comm_gathers_issued(rgpu_comm* comm) {
  if (!comm) return -1;
  std::lock_guard<std::mutex> g(comm->ctx->mu);
  return comm->gathers_issued;
}

// The one-process form: rank r = comms[r] / segs[r] (from rgpu_comm_init_all), one thread issues the whole batch. The n
// all-gathers sit in one NCCL group (issued one by one from one thread the first would wait for peers that this very
// thread has not reached yet).
extern "C" int32_t rgpu_search_batch_sharded_all(rgpu_comm* const* comms, rgpu_segment* const* segs, int32_t n, const rgpu_query* queries,
                                                 int32_t n_queries, const rgpu_query_term* const* terms_per_rank, int32_t n_terms_total,
                                                 int32_t k, void* const* hits_dev, void* const* totals_dev, void* const* hip_streams) {
  if (!comms || !segs || n < 1 || !queries || n_queries <= 0 || !terms_per_rank || n_terms_total <= 0 || !hits_dev || !totals_dev)
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  for (int32_t r = 0; r < n; ++r) {
    if (!comms[r] || !segs[r] || !terms_per_rank[r] || !hits_dev[r] || !totals_dev[r]) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null per-rank argument");
    if (segs[r]->ctx != comms[r]->ctx || comms[r]->rank != r || comms[r]->n_ranks != n)
      return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "comms[r] / segs[r] must be rank r of an n-rank rgpu_comm_init_all group");
  }
  std::vector<ShardedCall> calls((size_t)n);
  std::vector<hipStream_t> ss((size_t)n);
  for (int32_t r = 0; r < n; ++r) {  // phase 0 for every rank: a failure here returns before ANY rank has moved on
    rgpu_ctx* c = comms[r]->ctx;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    ss[(size_t)r] = (hip_streams && hip_streams[r]) ? (hipStream_t)hip_streams[r] : c->stream;
    int32_t rc = sharded_reserve(comms[r], n_queries, k, ss[(size_t)r], &calls[(size_t)r]);
    if (rc != RGPU_OK) return rc;
  }
  // From here to ncclGroupEnd nothing returns: every rank advances its slot, searches (a failing shard leaves its status in
  // its record) and issues its all-gather — a rank left out would leave the others hanging in theirs.
  int32_t first_rc = RGPU_OK;
  std::string first_why;
  auto note = [&](int32_t rc, const std::string& why) { if (rc != RGPU_OK && first_rc == RGPU_OK) { first_rc = rc; first_why = why; } };
  for (int32_t r = 0; r < n; ++r) {  // every shard's search is enqueued before any collective: the GPUs work side by side
    rgpu_ctx* c = comms[r]->ctx;
    std::lock_guard<std::mutex> g(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) {  // the search cannot run: an empty record with a status would need the device too
      comms[r]->next = (comms[r]->next + 1) % N_COMM_SLOTS;
      calls[(size_t)r].local_rc = RGPU_ERR_RUNTIME;
      calls[(size_t)r].local_why = "hipSetDevice";
      continue;
    }
    sharded_local(comms[r], segs[r], queries, n_queries, terms_per_rank[r], n_terms_total, k, ss[(size_t)r], &calls[(size_t)r], true);
  }
  // (round 5: every shard's kernels are enqueued by now — disjunctions included, whose groups used to end in a stream sync and so
  // ran shard after shard; only here does the thread wait, shard by shard, for flags that are back by then as a rule)
  for (int32_t r = 0; r < n; ++r) {
    rgpu_ctx* c = comms[r]->ctx;
    std::lock_guard<std::mutex> g(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) continue;
    sharded_settle(comms[r], n_queries, k, ss[(size_t)r], &calls[(size_t)r]);
  }
  {
    const ncclResult_t gs = ncclGroupStart();
    if (gs != ncclSuccess) note(RGPU_ERR_RUNTIME, std::string("ncclGroupStart: ") + ncclGetErrorString(gs));
    for (int32_t r = 0; r < n; ++r) {
      rgpu_ctx* c = comms[r]->ctx;
      std::lock_guard<std::mutex> g(c->mu);
      if (hipSetDevice(c->device) != hipSuccess) { note(RGPU_ERR_RUNTIME, "hipSetDevice"); continue; }
      const int32_t rc = sharded_gather(comms[r], ss[(size_t)r], calls[(size_t)r]);
      if (rc != RGPU_OK) note(rc, "rank " + std::to_string(r) + ": " + g_last_error);
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (ge != ncclSuccess) note(RGPU_ERR_RUNTIME, std::string("ncclGroupEnd: ") + ncclGetErrorString(ge));
  }
  if (first_rc != RGPU_OK) return fail(first_rc, first_why);  // a collective that could not be issued: rgpu_comm_destroy / a new communicator
  for (int32_t r = 0; r < n; ++r) {
    rgpu_ctx* c = comms[r]->ctx;
    std::lock_guard<std::mutex> g(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) { note(RGPU_ERR_RUNTIME, "hipSetDevice"); continue; }
    const int32_t rc = sharded_merge(comms[r], n_queries, k, hits_dev[r], totals_dev[r], ss[(size_t)r], calls[(size_t)r]);
    if (rc != RGPU_OK) note(rc, "rank " + std::to_string(r) + ": " + g_last_error);
    if (calls[(size_t)r].local_rc != RGPU_OK) note(calls[(size_t)r].local_rc, "rank " + std::to_string(r) + ": " + calls[(size_t)r].local_why);
    if (calls[(size_t)r].pre_rc != RGPU_OK) note(calls[(size_t)r].pre_rc, "rank " + std::to_string(r) + ": " + calls[(size_t)r].pre_why);
  }
  if (first_rc != RGPU_OK) return fail(first_rc, first_why);
  return RGPU_OK;
}

// Every rank's status word of the most recent sharded batch on this communicator (waits for that batch): 0, or the
// rgpu_status its local search failed with — the merged rows then lack that shard's hits.
extern "C" int32_t rgpu_comm_status(rgpu_comm* comm, int32_t* status_out) {
  if (!comm || !status_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  rgpu_ctx* c = comm->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  for (int r = 0; r < comm->n_ranks; ++r) status_out[r] = RGPU_OK;
  if (!comm->last_slot) return RGPU_OK;
  if (comm->last_slot->busy) { HIP_TRY(hipEventSynchronize(comm->last_slot->done)); comm->last_slot->busy = false; }
  const size_t record = record_bytes(comm->last_queries, comm->last_k);
  for (int r = 0; r < comm->n_ranks; ++r) {
    int64_t v = 0;
    HIP_TRY(hipMemcpy(&v, comm->last_slot->recv.p + (size_t)r * record + record - 8, 8, hipMemcpyDeviceToHost));
    status_out[r] = (int32_t)v;
  }
  return RGPU_OK;
}

// ---- host helpers: BM25Similarity (no GPU involved) ------------------------------------------------------------
#include "host/bm25_similarity.hpp"

extern "C" int32_t rgpu_bm25_compute_weight(float k1, float b, int64_t max_doc, int64_t doc_count, int64_t sum_total_term_freq,
                                            const int64_t* doc_freqs, int32_t n_terms, float boost, float* weight_out,
                                            float* idf_out, float* cache_out) {
  if (!doc_freqs || n_terms <= 0 || n_terms > 64) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  rucene::CollectionStatistics cs;
  cs.max_doc = max_doc;
  cs.doc_count = doc_count;
  cs.sum_total_term_freq = sum_total_term_freq;
  rucene::TermStatistics ts[64];
  for (int i = 0; i < n_terms; ++i) ts[i].doc_freq = doc_freqs[i];
  rucene::BM25SimWeight w = rucene::BM25Similarity(k1, b).compute_weight(cs, ts, (size_t)n_terms, boost);
  if (weight_out) *weight_out = w.weight;
  if (idf_out) *idf_out = w.idf;
  if (cache_out) std::memcpy(cache_out, w.cache.data(), 256 * sizeof(float));
  return RGPU_OK;
}

extern "C" int32_t rgpu_bm25_term_weights(int64_t max_doc, int64_t doc_count, const int64_t* doc_freqs, int64_t n, float boost,
                                          float* weights_out) {
  if (!doc_freqs || !weights_out || n < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  rucene::CollectionStatistics cs;
  cs.max_doc = max_doc;
  cs.doc_count = doc_count;
  for (int64_t i = 0; i < n; ++i) {
    rucene::TermStatistics ts;
    ts.doc_freq = doc_freqs[i];
    weights_out[i] = rucene::BM25Similarity::idf(&ts, 1, cs) * boost;  // BM25SimWeight::weight = idf * boost
  }
  return RGPU_OK;
}

extern "C" int32_t rgpu_norms_from_lucene53(const uint8_t* nvm, size_t nvm_len, const uint8_t* nvd, size_t nvd_len,
                                            int32_t field_number, int32_t max_doc, uint8_t* norms_out) {
  std::string why;
  const int rc = rucene::read_lucene53_norms(nvm, nvm_len, nvd, nvd_len, field_number, max_doc, norms_out, &why);
  return rc == 0 ? RGPU_OK : fail(rc, why);
}

extern "C" int32_t rgpu_live_docs_from_lucene50(const uint8_t* liv, size_t liv_len, int32_t max_doc, int32_t del_count,
                                                uint64_t* words_out) {
  std::string why;
  const int rc = rucene::read_lucene50_live_docs(liv, liv_len, max_doc, del_count, words_out, &why);
  return rc == 0 ? RGPU_OK : fail(rc, why);
}

// ---- term dictionary (host) --------------------------------------------------------------------------------------------
struct rgpu_terms {
  std::unique_ptr<rucene::TermDictionary> dict;
};
static_assert(sizeof(rucene::TermState) == sizeof(rgpu_term_state) && offsetof(rucene::TermState, doc_freq) == offsetof(rgpu_term_state, doc_freq) &&
                  offsetof(rucene::TermState, skip_offset) == offsetof(rgpu_term_state, skip_offset),
              "rucene::TermState must mirror rgpu_term_state");
static_assert(sizeof(rucene::TermFieldStats) == sizeof(rgpu_field_stats), "rucene::TermFieldStats must mirror rgpu_field_stats");

extern "C" int32_t rgpu_terms_open(const uint8_t* tim, size_t tim_len, const uint8_t* tip, size_t tip_len, const rgpu_field_info* infos,
                                   int32_t n_infos, int32_t max_doc, rgpu_terms** out_terms) {
  if (!out_terms) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out_terms is null");
  *out_terms = nullptr;
  if (n_infos < 0 || (n_infos > 0 && !infos)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad field infos");
  std::vector<rucene::TermFieldInfo> fi((size_t)n_infos);
  for (int32_t i = 0; i < n_infos; ++i) fi[i] = rucene::TermFieldInfo{infos[i].number, infos[i].index_options, infos[i].has_payloads};  // .flags is informational
  std::string why;
  auto h = std::make_unique<rgpu_terms>();
  int rc;
  try {
    rc = rucene::TermDictionary::open(tim, tim_len, tip, tip_len, fi.data(), n_infos, max_doc, &h->dict, &why);
  } catch (const std::bad_alloc&) {
    return fail(RGPU_ERR_RUNTIME, "out of host memory while building the term dictionary");
  }
  if (rc != 0) return fail(rc, why);
  *out_terms = h.release();
  return RGPU_OK;
}

extern "C" void rgpu_terms_close(rgpu_terms* terms) { delete terms; }

extern "C" int32_t rgpu_segment_info_from_lucene62(const uint8_t* si, size_t si_len, const uint8_t* expected_id16_or_null, rgpu_segment_info* out) {
  if (!out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out is null");
  rucene::SegmentInfoEntry e;
  std::string why;
  const int rc = rucene::read_lucene62_segment_info(si, si_len, expected_id16_or_null, &e, &why);
  if (rc != 0) return fail(rc, why);
  *out = rgpu_segment_info{};
  out->max_doc = e.max_doc;
  out->is_compound_file = e.is_compound_file ? 1 : 0;
  for (int i = 0; i < 3; ++i) out->version[i] = e.version[i];
  out->n_files = e.n_files;
  out->n_sort_fields = e.n_sort_fields;
  std::memcpy(out->id, e.id, 16);
  return RGPU_OK;
}

extern "C" int32_t rgpu_commit_from_segments_file(const uint8_t* data, size_t len, int64_t generation, rgpu_commit_segment* out, int32_t cap) {
  std::vector<rucene::CommitSegmentEntry> segs;
  std::string why;
  const int rc = rucene::read_segments_file(data, len, generation, &segs, &why);
  if (rc != 0) return fail(rc, why);
  for (size_t i = 0; i < segs.size() && out && (int64_t)i < cap; ++i) {
    const rucene::CommitSegmentEntry& s = segs[i];
    if (s.name.size() >= sizeof(out[i].name) || s.codec.size() >= sizeof(out[i].codec)) return fail(RGPU_ERR_UNSUPPORTED, "segment or codec name too long");
    out[i] = rgpu_commit_segment{};
    std::memcpy(out[i].name, s.name.c_str(), s.name.size());
    std::memcpy(out[i].codec, s.codec.c_str(), s.codec.size());
    std::memcpy(out[i].id, s.id, 16);
    out[i].del_gen = s.del_gen;
    out[i].field_infos_gen = s.field_infos_gen;
    out[i].dv_gen = s.dv_gen;
    out[i].del_count = s.del_count;
  }
  return (int32_t)segs.size();
}

extern "C" int32_t rgpu_compound_entries_from_lucene50(const uint8_t* cfe, size_t cfe_len, const uint8_t* cfs_or_null, size_t cfs_len,
                                                       const uint8_t* expected_id16_or_null, rgpu_compound_entry* out, int32_t cap) {
  std::vector<rucene::CompoundEntry> entries;
  std::string why;
  const int rc = rucene::read_lucene50_compound_entries(cfe, cfe_len, cfs_or_null, cfs_len, expected_id16_or_null, &entries, &why);
  if (rc != 0) return fail(rc, why);
  for (size_t i = 0; i < entries.size() && out && (int64_t)i < cap; ++i) {
    if (entries[i].id.size() >= sizeof(out[i].id)) return fail(RGPU_ERR_UNSUPPORTED, "compound entry name too long");
    out[i] = rgpu_compound_entry{};
    std::memcpy(out[i].id, entries[i].id.c_str(), entries[i].id.size());
    out[i].offset = entries[i].offset;
    out[i].length = entries[i].length;
  }
  return (int32_t)entries.size();
}

extern "C" int32_t rgpu_field_infos_from_lucene60(const uint8_t* fnm, size_t fnm_len, rgpu_field_info* infos_out, int32_t cap, char* names_out,
                                                  size_t names_cap, size_t* names_len_out) {
  std::vector<rucene::FieldInfoEntry> infos;
  std::string why;
  const int rc = rucene::read_lucene60_field_infos(fnm, fnm_len, &infos, &why);
  if (rc != 0) return fail(rc, why);
  size_t names_len = 0;
  for (size_t i = 0; i < infos.size(); ++i) {
    const rucene::FieldInfoEntry& fi = infos[i];
    if (infos_out && (int64_t)i < cap)
      infos_out[i] = rgpu_field_info{fi.number, fi.index_options, fi.store_payloads ? 1 : 0,
                                     (fi.omit_norms ? 1 : 0) | (fi.store_term_vector ? 2 : 0) | (fi.doc_values_type << 8)};
    if (names_out && names_len + fi.name.size() + 1 <= names_cap) std::memcpy(names_out + names_len, fi.name.c_str(), fi.name.size() + 1);
    names_len += fi.name.size() + 1;
  }
  if (names_len_out) *names_len_out = names_len;
  return (int32_t)infos.size();
}

extern "C" int32_t rgpu_terms_field_stats(const rgpu_terms* terms, int32_t field_number, rgpu_field_stats* out) {
  if (!terms || !out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  const rucene::TermFieldStats* st = terms->dict->field_stats(field_number);
  if (!st) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "no such indexed field in this segment");
  std::memcpy(out, st, sizeof(*out));
  return RGPU_OK;
}

static int32_t terms_lookup_impl(const rgpu_terms* terms, int32_t field_number, const uint8_t* term_bytes, const int64_t* term_offsets,
                                 int32_t n_terms, rgpu_term_state* states_out, rgpu_term_positions* positions_out, uint8_t* found_out) {
  if (!terms || n_terms < 0 || (n_terms > 0 && (!term_offsets || !states_out))) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  for (int32_t i = 0; i < n_terms; ++i)
    if (term_offsets[i] < 0 || term_offsets[i + 1] < term_offsets[i]) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_offsets must be non-decreasing");
  static const uint8_t kEmpty = 0;
  if (n_terms > 0 && term_offsets[n_terms] > term_offsets[0] && !term_bytes) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_bytes is null");
  static_assert(sizeof(rucene::TermState) == sizeof(rgpu_term_state), "layout");
  static_assert(sizeof(rucene::TermPositions) == sizeof(rgpu_term_positions), "layout");
  terms->dict->lookup_batch(field_number, term_bytes ? term_bytes : &kEmpty, term_offsets, n_terms,
                            reinterpret_cast<rucene::TermState*>(states_out), found_out,
                            reinterpret_cast<rucene::TermPositions*>(positions_out));
  return RGPU_OK;
}

extern "C" int32_t rgpu_terms_lookup(const rgpu_terms* terms, int32_t field_number, const uint8_t* term_bytes, const int64_t* term_offsets,
                                     int32_t n_terms, rgpu_term_state* states_out, uint8_t* found_out) {
  return terms_lookup_impl(terms, field_number, term_bytes, term_offsets, n_terms, states_out, nullptr, found_out);
}

extern "C" int32_t rgpu_terms_lookup_positions(const rgpu_terms* terms, int32_t field_number, const uint8_t* term_bytes,
                                               const int64_t* term_offsets, int32_t n_terms, rgpu_term_state* states_out,
                                               rgpu_term_positions* positions_out, uint8_t* found_out) {
  if (n_terms > 0 && !positions_out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "positions_out is null");
  return terms_lookup_impl(terms, field_number, term_bytes, term_offsets, n_terms, states_out, positions_out, found_out);
}

// ---- batch planner (host) -----------------------------------------------------------------------------------------------
#include "host/batch_planner.hpp"
struct rgpu_planner {
  std::unique_ptr<rucene::BatchPlanner> p;
};
static int32_t planner_sim_table(rgpu_ctx* c, const rgpu_plan_stats* ps, int32_t* table_out) {
  if (!ps) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  if (!(ps->k1 >= 0.0f) || !(ps->b >= 0.0f && ps->b <= 1.0f)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "illegal k1 / b value");  // BM25Similarity::new
  if (!c) { *table_out = -1; return RGPU_OK; }  // the host uploads the field's norm cache itself: rgpu_planner_set_sim_table
  rucene::CollectionStatistics cs;
  cs.max_doc = ps->max_doc;
  cs.doc_count = ps->doc_count;
  cs.sum_total_term_freq = ps->sum_total_term_freq;
  rucene::TermStatistics ts;
  const rucene::BM25SimWeight w = rucene::BM25Similarity(ps->k1, ps->b).compute_weight(cs, &ts, 1, 1.0f);  // the cache depends on the collection alone
  const int32_t t = rgpu_sim_table_upload(c, w.cache.data(), ps->k1);
  if (t < 0) return t;
  *table_out = t;
  return RGPU_OK;
}

extern "C" int32_t rgpu_planner_create_flat(rgpu_ctx* c, const rgpu_plan_stats* ps, const rgpu_term_state* leaf_states, int64_t n_leaf,
                                            const rgpu_term_state* stats_states_or_null, int64_t n_stats, rgpu_planner** out) {
  if (!out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out is null");
  *out = nullptr;
  if (n_leaf < 0 || (n_leaf > 0 && !leaf_states) || n_stats < 0 || (n_stats > 0 && !stats_states_or_null)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad term tables");
  int32_t table = -1;
  const int32_t rc = planner_sim_table(c, ps, &table);
  if (rc != RGPU_OK) return rc;
  try {
    auto h = std::make_unique<rgpu_planner>();
    h->p.reset(new rucene::BatchPlanner(*ps, table, leaf_states, n_leaf, stats_states_or_null, n_stats));
    *out = h.release();
  } catch (const std::bad_alloc&) { return fail(RGPU_ERR_RUNTIME, "out of host memory"); }
  return RGPU_OK;
}

extern "C" int32_t rgpu_planner_create(rgpu_ctx* c, const rgpu_plan_stats* ps, const rgpu_terms* leaf_terms, const rgpu_terms* stats_terms_or_null,
                                       int32_t field_number, rgpu_planner** out) {
  if (!out) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "out is null");
  *out = nullptr;
  if (!leaf_terms) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "leaf_terms is null");
  if (!leaf_terms->dict->field_stats(field_number)) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "no such indexed field in this segment");
  int32_t table = -1;
  const int32_t rc = planner_sim_table(c, ps, &table);
  if (rc != RGPU_OK) return rc;
  auto h = std::make_unique<rgpu_planner>();
  h->p.reset(new rucene::BatchPlanner(*ps, table, leaf_terms->dict.get(), stats_terms_or_null ? stats_terms_or_null->dict.get() : nullptr, field_number));
  *out = h.release();
  return RGPU_OK;
}

extern "C" void rgpu_planner_destroy(rgpu_planner* p) { delete p; }
extern "C" int32_t rgpu_planner_sim_table(const rgpu_planner* p) { return p ? p->p->sim_table() : RGPU_ERR_ILLEGAL_ARGUMENT; }
extern "C" int32_t rgpu_planner_set_sim_table(rgpu_planner* p, int32_t sim_table) {
  if (!p || sim_table < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  p->p->set_sim_table(sim_table);
  return RGPU_OK;
}

static int32_t plan_checked(rgpu_planner* p, int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not,
                            const int64_t* ids, const uint8_t* bytes, const int64_t* offsets, const float* boosts, rgpu_query* queries_out,
                            rgpu_query_term* terms_out, int64_t terms_cap) {
  if (!p || n_queries <= 0 || !ops || !n_terms || !queries_out || !terms_out || terms_cap < 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad arguments");
  std::string why;
  const int rc = p->p->plan(n_queries, ops, n_terms, n_must_not, ids, bytes, offsets, boosts, queries_out, terms_out, terms_cap, &why);
  return rc == RGPU_OK ? RGPU_OK : fail(rc, why);
}

extern "C" int32_t rgpu_plan_batch_ids(rgpu_planner* p, int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not_or_null,
                                       const int64_t* term_ids, const float* boosts_or_null, rgpu_query* queries_out, rgpu_query_term* terms_out,
                                       int64_t terms_cap) {
  if (!term_ids) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_ids is null");
  return plan_checked(p, n_queries, ops, n_terms, n_must_not_or_null, term_ids, nullptr, nullptr, boosts_or_null, queries_out, terms_out, terms_cap);
}

extern "C" int32_t rgpu_plan_batch_bytes(rgpu_planner* p, int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not_or_null,
                                         const uint8_t* term_bytes, const int64_t* term_offsets, const float* boosts_or_null, rgpu_query* queries_out,
                                         rgpu_query_term* terms_out, int64_t terms_cap) {
  if (!term_offsets) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_offsets is null");
  static const uint8_t kEmpty = 0;
  return plan_checked(p, n_queries, ops, n_terms, n_must_not_or_null, nullptr, term_bytes ? term_bytes : &kEmpty, term_offsets, boosts_or_null, queries_out,
                      terms_out, terms_cap);
}

// every query the same shape: `op` (RGPU_OP_TERM / AND / OR, OR optionally RGPU_OP_OR_MSM) over n_clauses terms, boost 1
static int32_t plan_uniform(rgpu_planner* p, int32_t op, int32_t n_queries, int32_t n_clauses, const int64_t* ids, const uint8_t* bytes,
                            const int64_t* offsets, rgpu_query* queries_out, rgpu_query_term* terms_out) {
  if (n_queries <= 0 || n_clauses <= 0 || n_clauses > RGPU_MAX_QUERY_TERMS) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad batch shape");
  if ((op >> 16) != 0) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "a uniform batch has no optional clauses");
  thread_local std::vector<int32_t> ops, counts;
  ops.assign((size_t)n_queries, op);
  counts.assign((size_t)n_queries, n_clauses);
  return plan_checked(p, n_queries, ops.data(), counts.data(), nullptr, ids, bytes, offsets, nullptr, queries_out, terms_out,
                      (int64_t)n_queries * n_clauses);
}
extern "C" int32_t rgpu_plan_uniform_ids(rgpu_planner* p, int32_t op, int32_t n_queries, int32_t n_clauses, const int64_t* term_ids,
                                         rgpu_query* queries_out, rgpu_query_term* terms_out) {
  if (!term_ids) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_ids is null");
  return plan_uniform(p, op, n_queries, n_clauses, term_ids, nullptr, nullptr, queries_out, terms_out);
}
extern "C" int32_t rgpu_plan_uniform_bytes(rgpu_planner* p, int32_t op, int32_t n_queries, int32_t n_clauses, const uint8_t* term_bytes,
                                           const int64_t* term_offsets, rgpu_query* queries_out, rgpu_query_term* terms_out) {
  if (!term_offsets) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "term_offsets is null");
  static const uint8_t kEmpty = 0;
  return plan_uniform(p, op, n_queries, n_clauses, nullptr, term_bytes ? term_bytes : &kEmpty, term_offsets, queries_out, terms_out);
}

// ---- plan + search in ONE call ------------------------------------------------------------------------------------------------
// A serving host's steady state is "a batch of term ids arrives -> rows": rgpu_plan_uniform_ids writes rgpu_query[] /
// rgpu_query_term[] (40 B a clause), rgpu_search_batch_device reads them back, validates them as foreign input, looks every
// clause's prepared structures up, sorts the queries into groups and only then builds the device descriptors — 63 us of one
// host thread per 1024-query TERM batch against 47 us of GPU time (round 5). The fused form keeps the ids inside the library:
// for single-term batches the planner's per-clause result (term state, idf) and the prepared-term table entry go straight
// into the staged DevQuery / DevTerm arrays — one pass, no intermediate arrays, nothing to validate (the planner's own
// tables are trusted; an id outside them is an absent term) — and the zeroed per-query counters travel with the same staged
// copy (three enqueues per batch: copy, k_search_term, k_merge_items). Anything the fast pass does not handle — another op,
// a term that is not prepared yet or lacks its block-max sketch, k > 128, a prepared-term budget, a dictionary planner — goes
// through plan() + search_impl() inside the same call: same rows either way (tests/test_gpu_parity.py).
// RGPU_HOST_TIME=1 in the environment: where the calling thread's time goes inside term_batch_fast, averaged over 256 calls (stderr)
struct HostLaps {
  static constexpr int N = 7;
  long long ns[N] = {0, 0, 0, 0, 0, 0, 0};
  int calls = 0;
  std::chrono::steady_clock::time_point last;
  void start() { last = std::chrono::steady_clock::now(); }
  void lap(int i) { const auto t = std::chrono::steady_clock::now(); ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(t - last).count(); last = t; }
  void done() {
    if (++calls < 256) return;
    std::fprintf(stderr, "[term_batch_fast host, us per call] slot + stage room %.2f | term states -> descriptors %.2f | items %.2f | item descriptors + zeroes %.2f | "
                         "stage copy enqueue %.2f | reserve + kernel enqueue %.2f | status + mark %.2f\n",
                 ns[0] / 256e3, ns[1] / 256e3, ns[2] / 256e3, ns[3] / 256e3, ns[4] / 256e3, ns[5] / 256e3, ns[6] / 256e3);
    *this = HostLaps{};
  }
};
static int32_t term_batch_fast(rgpu_segment* seg, rucene::BatchPlanner* P, int32_t nq, const int64_t* ids, int32_t k, HitOut* hits_dev,
                               int64_t* totals_dev, hipStream_t stream, bool* taken) {
  rgpu_ctx* c = seg->ctx;
  *taken = false;
  static thread_local HostLaps laps;
  const bool timed = c->host_time;
  if (timed) laps.start();
  const int32_t sim_table = P->sim_table();
  if (!P->flat() || k > RGPU_PASS_K || c->prepared_budget != 0 || sim_table < 0 || sim_table >= c->n_sim_tables || seg->dir_used == 0) return RGPU_OK;
  const bool want_sketch = c->term_sketches && seg->d_norms && seg->n_norm_ranks > 0 && !seg->d_live;
  const bool need_norms = seg->d_norms != nullptr;
  const uint32_t flags = c->sim_monotone[(size_t)sim_table] ? TERM_FLAG_MONOTONE : 0u;
  c->pass = rgpu_ctx::Pass{};
  SCRATCH_TAKE(c);
  Stager st(c);
  const size_t o_q = st.add((size_t)nq * sizeof(DevQuery));
  const size_t o_t = st.add((size_t)nq * sizeof(DevTerm));
  const size_t o_p = st.add((size_t)(nq + 1) * 8);
  const size_t o_m = st.add((size_t)nq * 4);
  const size_t o_sh = st.add((size_t)nq);            // log2 of every query's own item size (0: the launch's)
  const size_t o_tau = st.add((size_t)nq * 8);       // per-query shared thresholds ...
  const size_t o_w = st.add((size_t)nq * 16);        // ... and the launch's counters: zeroed by the copy that brings the plan
  const size_t o_done = st.add((size_t)nq * 4);      // ... and the per-query counts of finished items (TermMerge::done)
  const size_t o_zero_end = st.add(0);               // (the three above are one range: [o_tau, o_zero_end))
  // ... and the item descriptors, LAST: their number is known only once the terms have been read (at most 262144 + nq, the item
  // loop's cap), the stage has room for the worst case and the copy takes what is used
  const size_t o_id = st.add(((size_t)262144 + (size_t)nq) * sizeof(int4));
  HIP_TRY(c->S->h_stage.reserve(st.used));
  HIP_TRY(c->S->d_stage.reserve(st.used, 0, stream));
  if (timed) laps.lap(0);
  DevQuery* hq = reinterpret_cast<DevQuery*>(c->S->h_stage.p + o_q);
  DevTerm* ht = reinterpret_cast<DevTerm*>(c->S->h_stage.p + o_t);
  int64_t* hp = reinterpret_cast<int64_t*>(c->S->h_stage.p + o_p);
  int32_t* hm = reinterpret_cast<int32_t*>(c->S->h_stage.p + o_m);
  int32_t nt = 0;
  int64_t postings = 0, total_blocks = 0, loose = 0;
  bool bail = false;
  const int32_t max_doc = seg->max_doc;
  const bool has_freqs = seg->has_freqs;
  // term id -> finished descriptor through the planner's memo (host/batch_planner.hpp for_each_flat_memo): a descriptor is a function
  // of the planner's tables (immutable), the segment and what its prepared store holds (uid, epoch), and the flags below
  const rucene::BatchPlanner::MemoKey memo_key{{seg->uid, seg->prepared.epoch, (uint64_t)(uint32_t)sim_table | ((uint64_t)flags << 32),
                                                (uint64_t)want_sketch | ((uint64_t)need_norms << 1) | ((uint64_t)has_freqs << 2) | ((uint64_t)(uint32_t)max_doc << 8)}};
  bail = !P->for_each_flat_memo<DevTerm>(ids, nq, memo_key, [&](const rgpu_term_state& s0, float idf, DevTerm* out) -> int32_t {
    DevTerm t;
    t.start_fp = (uint64_t)std::max<int64_t>(0, s0.doc_start_fp);
    t.pn_base = 0;
    t.bs_base = 0;
    t.dir_base = 0;
    t.nblocks = 0;
    t.df = s0.doc_freq;
    t.tail_n = s0.doc_freq > 1 ? s0.doc_freq % 128 : 0;
    t.singleton_doc = s0.singleton_doc_id;
    t.singleton_freq = has_freqs ? (int32_t)s0.total_term_freq : 1;
    t.weight = idf;
    t.sim_table = sim_table;
    t.flags = flags;
    t.sketch = 0;
    if (s0.doc_freq == 1) {
      if (s0.singleton_doc_id < 0 || s0.singleton_doc_id >= max_doc) return -1;  // (the full path names the error)
    } else {
      const TermInfo* info = seg->prepared.find(s0.doc_start_fp);
      if (!info || info->df != s0.doc_freq || (need_norms && !info->norms) ||
          (want_sketch && info->sketch == 0 && info->nblocks >= TERM_SKETCH_MIN_BLOCKS)) return -1;  // (the full path prepares it: another epoch)
      t.dir_base = info->dir_base;
      t.nblocks = info->nblocks;
      t.pn_base = info->pn_base;
      t.bs_base = info->bs_base;
      t.sketch = info->sketch;
    }
    *out = t;
    return 1;
  }, [&](int64_t q, const DevTerm* t) {
    hm[q] = (int32_t)q;
    if (!t) { hq[q] = DevQuery{RGPU_OP_TERM, 0, nt, 0}; return; }  // TermWeight::create_scorer -> None for this leaf
    hq[q] = DevQuery{RGPU_OP_TERM, 1, nt, 0};
    ht[nt++] = *t;
    postings += t->df;
    total_blocks += t->nblocks;
    loose += t->df == 1 ? 1 : t->tail_n;
  });
  if (bail) return RGPU_OK;  // (the slot was taken and not marked: it is simply free again)
  if (timed) laps.lap(1);
  // items: chunks of a term's blocks; every query's first chunk is scheduled first (search_pass's rule, to the letter)
  int blocks_per_item = c->cfg.blocks_per_item;
  int term_split = 1;
  if (c->blocks_per_item_auto) {
    blocks_per_item = term_item_blocks(total_blocks, c->term_target_items);
    term_split = c->term_split;
  }
  int64_t items = 0;
  uint8_t* hsh = c->S->h_stage.p + o_sh;
  while (true) {
    items = 0;
    const int base_sh = term_item_shift(blocks_per_item);  // 0: not a power of two (a caller's own size) — no per-query sizes then
    const int floor_sh = std::max(1, term_item_shift(c->term_min_item_blocks));
    for (int q = 0; q < nq; ++q) {
      hp[q] = items;
      hsh[q] = 0;
      if (hq[q].n_terms >= 1) {
        const int32_t nb = ht[hq[q].first_term].nblocks;
        if (base_sh > 0) {  // term_query_item_blocks in shifts: halve while the query has fewer than `split` items
          int sh = base_sh;
          if (term_split > 1) while (sh > floor_sh && ((nb + (1 << sh) - 1) >> sh) < term_split) --sh;
          hsh[q] = (uint8_t)sh;
          items += (nb == 0 ? 1 : (nb + (1 << sh) - 1) >> sh) - 1;
        } else {
          items += (nb == 0 ? 1 : (nb + blocks_per_item - 1) / blocks_per_item) - 1;
        }
      }
    }
    hp[nq] = items;
    if (items <= 262144 || blocks_per_item >= (1 << 17)) break;
    blocks_per_item *= 2;
  }
  items += nq;
  if (timed) laps.lap(2);
  if (items > 262144 + (int64_t)nq) return RGPU_OK;  // (blocks_per_item hit its ceiling on an absurd batch: the full path takes it)
  const bool plan_on_device = c->stage_by_kernel && !c->upload_aside && o_id + (size_t)items * sizeof(int4) <= 0xffffffffull;
  if (plan_on_device) {
    // zeroes and item descriptors are the copy kernel's work (k_stage_term_plan)
    if (timed) laps.lap(3);
    const size_t n16 = (o_id + 15) / 16;  // (o_id is 256-aligned: everything in front of the descriptors)
    const TermPlanLayout L{(uint32_t)o_q, (uint32_t)o_p, (uint32_t)o_sh, (uint32_t)o_id, (uint32_t)o_tau, (uint32_t)o_zero_end, nq};
    TimedLaunch tl(c, stream, "k_stage_term_plan", 0);
    const unsigned grid = (unsigned)std::min<size_t>(1024, std::max<size_t>((n16 + 255) / 256, ((size_t)nq + 255) / 256));
    RGPU_LAUNCH(k_stage_term_plan, dim3(grid), dim3(256), 0, stream, c->S->h_stage.p, c->S->d_stage.p, n16, L);
  } else {
    std::memset(c->S->h_stage.p + o_tau, 0, o_zero_end - o_tau);
    int4* hd = reinterpret_cast<int4*>(c->S->h_stage.p + o_id);
    for (int q = 0; q < nq; ++q) {  // fill_term_item_desc with the sizes the item loop chose
      const int n_mine = 1 + (int)(hp[q + 1] - hp[q]);
      const int ft = hq[q].n_terms >= 1 ? hq[q].first_term : -1;
      const int w = n_mine | ((int)hsh[q] << 24);
      hd[q] = make_int4(q, 0, ft, w);
      int4* rest = hd + nq + hp[q];
      for (int ch = 1; ch < n_mine; ++ch) rest[ch - 1] = make_int4(q, ch, ft, w);
    }
    const size_t staged = o_id + (size_t)items * sizeof(int4);
    if (timed) laps.lap(3);
    HIP_TRY(stage_h2d(c, staged, stream));
  }
  if (timed) laps.lap(4);
  HIP_TRY(c->S->d_partial_keys.reserve((size_t)items * (size_t)k, 0, stream));
  HIP_TRY(c->S->d_partial_counts.reserve((size_t)items, 0, stream));
  unsigned long long* d_tau = reinterpret_cast<unsigned long long*>(c->S->d_stage.p + o_tau);
  unsigned long long* d_work = reinterpret_cast<unsigned long long*>(c->S->d_stage.p + o_w);
  const DevQuery* dq = reinterpret_cast<const DevQuery*>(c->S->d_stage.p + o_q);
  const DevTerm* dt = reinterpret_cast<const DevTerm*>(c->S->d_stage.p + o_t);
  const int64_t* dp = reinterpret_cast<const int64_t*>(c->S->d_stage.p + o_p);
  const int32_t* dm = reinterpret_cast<const int32_t*>(c->S->d_stage.p + o_m);
  c->last_counted = c->S;
  c->last_counted_words = d_work;
  c->last_counted_op = RGPU_OP_TERM;
  c->last_counted_queries = nq;
  c->last_counted_postings = postings;
  c->last_counted_dir_blocks = total_blocks;
  c->last_counted_loose = loose;
  const bool wide = k > 64;
  const bool legacy = seg->version < 1;
  // the fold of every query's item lists happens inside k_search_term (TermMerge), unless RGPU_TERM_FOLD=0 asks for k_merge_items
  const TermMerge fold = c->term_fold ? TermMerge{reinterpret_cast<unsigned int*>(c->S->d_stage.p + o_done), dp, hits_dev, totals_dev, nullptr, seg->doc_base, 0, 0}
                                      : TermMerge{};
  {
    TimedLaunch tl(c, stream, "k_search_term", postings);
    const unsigned grid = wg_count((items + TERM_WAVES - 1) / TERM_WAVES);
    const size_t lds = term_lds_bytes(wide);
    const SegView sv = seg_view(seg);
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = set_dynamic_lds_once(c, reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      RGPU_LAUNCH(kern, dim3(grid), dim3(TERM_THREADS), lds, stream, sv, dq, dt, reinterpret_cast<const int4*>(c->S->d_stage.p + o_id), nq, items,
                         blocks_per_item, (int)k, c->S->d_partial_keys.p, c->S->d_partial_counts.p, d_tau, d_work, (const unsigned long long*)nullptr, dm, fold);
      return hipSuccess;
    };
    hipError_t e;
    if (legacy) e = wide ? go(k_search_term<true, true>) : go(k_search_term<true, false>);
    else e = wide ? go(k_search_term<false, true>) : go(k_search_term<false, false>);
    HIP_TRY(e);
  }
  if (!c->term_fold) {
    if (wide) launch_merge<true>(c, stream, nq, k, dp, seg->doc_base, hits_dev, totals_dev, nq, nullptr, nullptr, dm);
    else launch_merge<false>(c, stream, nq, k, dp, seg->doc_base, hits_dev, totals_dev, nq, nullptr, nullptr, dm);
  }
  if (timed) laps.lap(5);
  HIP_TRY(launch_status());
  HIP_TRY(scratch_mark(c, stream));
  c->stats[(size_t)stat_slot(c, "fused_term_batches")].launches += 1;  // (tests ask whether this path ran: rgpu_kernel_stats)
  if (timed) { laps.lap(6); laps.done(); }
  *taken = true;
  return RGPU_OK;
}

// context mutex held by the caller; c->defer_or as the caller set it
static int32_t search_uniform_locked(rgpu_segment* seg, const UniformBatch& ub, int32_t n_queries, int32_t k, HitOut* hits_dev, int64_t* totals_dev,
                                     hipStream_t stream) {
  const int32_t op = ub.op & 0xff;
  // BooleanQuery::build: a lone clause IS that clause (boolean_query.rs:56-68) — as plan() rewrites it
  if (ub.n_clauses == 1 && (ub.op >> 8) == 0 && op >= RGPU_OP_TERM && op <= RGPU_OP_OR) {
    bool taken = false;
    const int32_t rc = term_batch_fast(seg, ub.planner->p.get(), n_queries, ub.ids, k, hits_dev, totals_dev, stream, &taken);
    if (rc != RGPU_OK || taken) return rc;
  }
  thread_local std::vector<rgpu_query> queries;
  thread_local std::vector<rgpu_query_term> terms;
  queries.resize((size_t)n_queries);
  terms.resize((size_t)n_queries * (size_t)ub.n_clauses);
  const int32_t rc = plan_uniform(ub.planner, ub.op, n_queries, ub.n_clauses, ub.ids, nullptr, nullptr, queries.data(), terms.data());
  if (rc != RGPU_OK) return rc;
  return search_impl(seg, queries.data(), n_queries, terms.data(), (int32_t)terms.size(), k, hits_dev, totals_dev, stream);
}

static int32_t uniform_args_ok(rgpu_planner* p, rgpu_segment* seg, int32_t op, int32_t n_queries, int32_t n_clauses, const int64_t* ids, int32_t k,
                               const void* hits_dev, const void* totals_dev) {
  if (!p || !seg || !ids || !hits_dev || !totals_dev) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  if (n_queries <= 0 || n_clauses <= 0 || n_clauses > RGPU_MAX_QUERY_TERMS || (int64_t)n_queries * n_clauses > 0x7fffffff) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "bad batch shape");
  if ((op >> 16) != 0 || (op & 0xff) < RGPU_OP_TERM || (op & 0xff) > RGPU_OP_OR || ((op & 0xff) == RGPU_OP_TERM && n_clauses != 1))
    return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "a uniform batch is TERM (one clause), AND or OR (optionally RGPU_OP_OR_MSM)");
  if (k <= 0 || k > RGPU_MAX_K) return fail(k <= 0 ? RGPU_ERR_ILLEGAL_ARGUMENT : RGPU_ERR_UNSUPPORTED, "k must be in 1..RGPU_MAX_K");
  if (!p->p->flat()) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "this planner names terms by their bytes (rgpu_planner_create): plan with rgpu_plan_uniform_bytes, then rgpu_search_batch_device");
  return RGPU_OK;
}

extern "C" int32_t rgpu_planner_search_uniform_ids_device(rgpu_planner* p, rgpu_segment* seg, int32_t op, int32_t n_queries, int32_t n_clauses,
                                                          const int64_t* term_ids, int32_t k, void* hits_dev, void* totals_dev, void* hip_stream) {
  const int32_t ok = uniform_args_ok(p, seg, op, n_queries, n_clauses, term_ids, k, hits_dev, totals_dev);
  if (ok != RGPU_OK) return ok;
  rgpu_ctx* c = seg->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  const UniformBatch ub{p, op, n_clauses, term_ids};
  c->defer_or = c->cfg.or_deferred != 0;
  const int32_t rc = search_uniform_locked(seg, ub, n_queries, k, (HitOut*)hits_dev, (int64_t*)totals_dev, s);
  c->defer_or = false;
  return rc;
}

// ... and its sharded form: rgpu_search_batch_sharded with the batch named by ids (local plan + search -> all-gather -> merge)
extern "C" int32_t rgpu_planner_search_uniform_ids_sharded(rgpu_comm* comm, rgpu_planner* p, rgpu_segment* seg, int32_t op, int32_t n_queries,
                                                           int32_t n_clauses, const int64_t* term_ids, int32_t k, void* hits_dev, void* totals_dev,
                                                           void* hip_stream) {
  if (!comm) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "null argument");
  const int32_t ok = uniform_args_ok(p, seg, op, n_queries, n_clauses, term_ids, k, hits_dev, totals_dev);
  if (ok != RGPU_OK) return ok;
  if (seg->ctx != comm->ctx) return fail(RGPU_ERR_ILLEGAL_ARGUMENT, "segment and communicator belong to different contexts");
  rgpu_ctx* c = comm->ctx;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  const UniformBatch ub{p, op, n_clauses, term_ids};
  ShardedCall call;
  int32_t rc = sharded_reserve(comm, n_queries, k, s, &call);
  if (rc != RGPU_OK) return rc;
  sharded_local(comm, seg, nullptr, n_queries, nullptr, 0, k, s, &call, false, &ub);
  rc = sharded_gather(comm, s, call);
  if (rc != RGPU_OK) return rc;
  rc = sharded_merge(comm, n_queries, k, hits_dev, totals_dev, s, call);
  if (rc != RGPU_OK) return rc;
  if (call.local_rc != RGPU_OK) return fail(call.local_rc, call.local_why);
  if (call.pre_rc != RGPU_OK) return fail(call.pre_rc, call.pre_why);
  return RGPU_OK;
}

extern "C" uint8_t rgpu_bm25_encode_norm(float boost, int32_t field_length) {
  return rucene::BM25Similarity::encode_norm_value(boost, field_length);
}

#ifdef RGPU_ORX_TIME
extern "C" int32_t rgpu_debug_counters(unsigned long long* out8, int32_t reset) {  // k_or_wide wave-cycles per phase (search_or_wide.hpp)
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_orx_dbg), 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_orx_dbg), z, 64) != hipSuccess) return -1;
  }
  return 0;
}
#endif
#ifdef RGPU_LZ_TIME
extern "C" int32_t rgpu_debug_counters(unsigned long long* out8, int32_t reset) {  // k_or_lazy wave-cycles per phase (search_or_lazy.hpp)
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lz_dbg), 64) != hipSuccess) return -1;
  unsigned long long more[8];
  if (hipMemcpyFromSymbol(more, HIP_SYMBOL(g_lz_dbg), 64, 64) != hipSuccess) return -1;
  std::fprintf(stderr, "[lz steps] lazy-only test on in %llu of %llu steps; a doc in enough lists in %llu; bound iterations %llu; mean need %.2f, need_hi %.2f\n", more[0], more[3],
               more[1], more[2], more[0] ? (double)more[4] / (double)more[0] : 0.0, more[0] ? (double)more[5] / (double)more[0] : 0.0);
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lz_dbg), z, 128) != hipSuccess) return -1;
  }
  return 0;
}
#endif
#ifdef RGPU_AND_TIME
extern "C" int32_t rgpu_debug_counters(unsigned long long* out8, int32_t reset) {  // k_search_and wave-cycles per phase (search_and.hpp)
  unsigned long long all[16];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_and_time), 128) != hipSuccess) return -1;
  std::memcpy(out8, all, 64);
  std::fprintf(stderr, "[and time] cycles: set-up %llu, first probe %llu, loop(queued) %llu, loop(block by block) %llu, epilogue %llu | items %llu, pops %llu, block vectors %llu, "
               "queued survivors %llu, walked decodes: after a pop %llu, block by block %llu\n", all[0], all[1], all[2], all[3], all[4], all[5], all[6], all[7], all[8], all[9], all[10]);
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_and_time), z, 128) != hipSuccess) return -1;
  }
  return 0;
}
#endif
#ifdef RGPU_TERM_TRACE
extern "C" int32_t rgpu_debug_trace(void* out, int32_t n) {  // the most recent k_search_term launch's item timeline
  if (n > TERM_TRACE_CAP) n = TERM_TRACE_CAP;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_term_trace), (size_t)n * sizeof(TermTraceRec)) == hipSuccess ? n : -1;
}
#endif
#ifdef RGPU_AND_TRACE
// the most recent k_search_and launch's item timeline: n records of {t0, t1 (100 MHz wall clock), q, chunk, lead blocks, survivors popped}
extern "C" int32_t rgpu_debug_trace(void* out, int32_t n) {
  if (n > AND_TRACE_CAP) n = AND_TRACE_CAP;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_and_trace), (size_t)n * sizeof(AndTraceRec)) == hipSuccess ? n : -1;
}
#endif
#ifdef RGPU_EXP_COUNT
extern "C" int32_t rgpu_debug_counters(unsigned long long* out8, int32_t reset) {  // [0..3] TERM, [4..7] AND
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_term_dbg), 32) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out8 + 4, HIP_SYMBOL(g_and_dbg), 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_term_dbg), z, 32) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_and_dbg), z, 32) != hipSuccess) return -1;
  }
  return 0;
}
#endif
