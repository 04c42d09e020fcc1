// gfx950 wavefront primitives (wave64): DPP prefix scans, readlane helpers, monotone score keys.
// Everything here assumes a full 64-lane wavefront executing convergently.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rgpu {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// Wave index inside the workgroup, as a scalar: everything derived from it (work item, term descriptor,
// block counts, header words) then lives in SGPRs and control flow on it is branch-, not exec-mask-based.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Wave-synchronous LDS hand-off between lanes of one wavefront: LDS operations of a wave complete in issue
// order, so only the compiler has to be kept from moving accesses across the hand-off point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP controls (CDNA ISA, DPP_CTRL): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143,
// wave_shr:1 = 0x138.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or0(int v) {
  // lanes whose source is out of range (or masked off by ROW_MASK) receive `old` = 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive +scan over the 64 lanes: 4 row_shr steps inside each 16-lane row, then two row broadcasts.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_or0<0x111, 0xf>(v);
  v += dpp_or0<0x112, 0xf>(v);
  v += dpp_or0<0x114, 0xf>(v);
  v += dpp_or0<0x118, 0xf>(v);
  v += dpp_or0<0x142, 0xa>(v);  // lane 15 of rows 0,2 -> rows 1,3
  v += dpp_or0<0x143, 0xc>(v);  // lane 31 -> rows 2,3
  return v;
}

__device__ __forceinline__ int readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }

// whole-wave shift towards higher lanes by one (lane 0 receives `fill`)
__device__ __forceinline__ uint64_t wave_shr1_64(uint64_t v, uint64_t fill) {
  int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

__device__ __forceinline__ int wave_reduce_add(int v) {
  return readlane(wave_incl_scan(v), 63);
}

// max over the 64 lanes (unsigned), returned as a scalar; out-of-range DPP sources read 0
__device__ __forceinline__ uint32_t wave_reduce_max_u32(uint32_t v) {
  auto mx = [](uint32_t a, int b) { return a > (uint32_t)b ? a : (uint32_t)b; };
  v = mx(v, dpp_or0<0x111, 0xf>((int)v));
  v = mx(v, dpp_or0<0x112, 0xf>((int)v));
  v = mx(v, dpp_or0<0x114, 0xf>((int)v));
  v = mx(v, dpp_or0<0x118, 0xf>((int)v));
  v = mx(v, dpp_or0<0x142, 0xa>((int)v));
  v = mx(v, dpp_or0<0x143, 0xc>((int)v));
  return (uint32_t)readlane((int)v, 63);
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- (score, doc) -> one sortable u64; larger == better under "score desc, then doc asc" ----------------------
__device__ __forceinline__ uint32_t float_order_bits(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_bits_float(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, int32_t doc) {
  return ((uint64_t)float_order_bits(score) << 32) | (uint32_t)(~(uint32_t)doc);
}
__device__ __forceinline__ int32_t key_doc(uint64_t k) { return (int32_t)(~(uint32_t)k); }
// k > 128 is served in passes of up to 128 hits (the widest list a wavefront's registers hold): pass p collects the best
// keys strictly BELOW the worst key of pass p - 1 (keys are unique per doc and totally ordered, so the passes partition the
// ranking exactly). `ceil` = that key (~0 in the first pass, 0 when the previous pass ran out of hits).
__device__ __forceinline__ uint64_t below(uint64_t key, uint64_t ceil) { return key < ceil ? key : 0ull; }
__device__ __forceinline__ float key_score(uint64_t k) { return order_bits_float((uint32_t)(k >> 32)); }

// ---- wave-resident top-k (k <= 128): rank r lives in lane r of `a` (r < 64) or lane r-64 of `b` ----------------
struct WaveTopK {
  uint64_t a = 0, b = 0;  // 0 == empty slot (every real key has a nonzero high word)
};

template <bool WIDE>  // WIDE: k > 64
__device__ __forceinline__ void topk_insert(WaveTopK& t, uint64_t key, int lane) {
  int pa = __popcll(__ballot(t.a > key));
  if (pa < 64) {
    uint64_t carry = readlane64(t.a, 63);
    uint64_t up = wave_shr1_64(t.a, 0);
    t.a = lane < pa ? t.a : (lane == pa ? key : up);
    if (WIDE) t.b = wave_shr1_64(t.b, carry);
  } else if (WIDE) {
    int pb = __popcll(__ballot(t.b > key));
    uint64_t up = wave_shr1_64(t.b, 0);
    t.b = lane < pb ? t.b : (lane == pb ? key : up);
  }
}
template <bool WIDE>
__device__ __forceinline__ uint64_t topk_threshold(const WaveTopK& t, int k) {
  if (WIDE && k > 64) return readlane64(t.b, k - 65);
  return readlane64(t.a, k - 1);
}

// Offer one key per lane (invalid lanes pass 0); keeps `tau` (the current entry threshold) up to date.
// `floor` is a lower bound on the query's global k-th best learnt from other wavefronts (0 if none): a key
// at or below it can never be in the final top-k, so it is not worth an insertion here either.
template <bool WIDE>
__device__ __forceinline__ void topk_offer(WaveTopK& t, uint64_t key, uint64_t& tau, int k, int lane, uint64_t floor = 0) {
  uint64_t m = __ballot(key > tau);
  while (m) {
    int src = __builtin_ctzll(m);
    uint64_t x = readlane64(key, src);
    topk_insert<WIDE>(t, x, lane);
    const uint64_t kth = topk_threshold<WIDE>(t, k);
    tau = kth > floor ? kth : floor;
    m &= m - 1;
    m &= __ballot(key > tau);
  }
}

// ---- merging a SORTED list into the wave-resident top-k: a bitonic merge instead of one insertion per key ------------------------
// Every partial list a search kernel leaves is a WaveTopK at rest: best first, 0 = empty. Offering such a list key by key
// costs an insertion (ballot, two shifts, a threshold read: ~150 cycles) per ENTERING key — a single 10-clause query's 612 lists
// of up to 100 keys took one wavefront 167 us. Two sorted n-lists merge in log2(n) + 1 compare-exchange rounds instead:
// x[i] = max(T[i], L[n - 1 - i]) is a bitonic sequence that holds the n best keys of the union; rounds of "pair i with
// i ^ d, the lower lane keeps the larger key" for d = n/2 .. 1 sort it (descending). Keys are unique (score, doc) pairs or 0.
__device__ __forceinline__ uint64_t wave_shfl64(uint64_t v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t bitonic_desc64(uint64_t x, int lane) {  // x: a bitonic 64-sequence, one key per lane
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint64_t y = wave_shfl64(x, lane ^ d);
    x = (lane & d) ? (x < y ? x : y) : (x > y ? x : y);
  }
  return x;
}
// t := the 64 (WIDE: 128) best keys of t and the sorted list {la, lb} (lb: ranks 64..127, unused unless WIDE), sorted.
// Ranks at and beyond k may then hold keys below the k-th best: nobody reads them (topk_threshold, the list's writers).
template <bool WIDE>
__device__ __forceinline__ void topk_merge_sorted(WaveTopK& t, uint64_t la, uint64_t lb, int lane) {
  if (!WIDE) {
    const uint64_t r = wave_shfl64(la, 63 - lane);
    t.a = bitonic_desc64(t.a > r ? t.a : r, lane);
  } else {
    const uint64_t ra = wave_shfl64(la, 63 - lane), rb = wave_shfl64(lb, 63 - lane);
    const uint64_t xa = t.a > rb ? t.a : rb;  // ranks 0..63 against the list's ranks 127..64
    const uint64_t xb = t.b > ra ? t.b : ra;  // ranks 64..127 against its ranks 63..0
    t.a = bitonic_desc64(xa > xb ? xa : xb, lane);
    t.b = bitonic_desc64(xa > xb ? xb : xa, lane);
  }
}
// One sorted list offered to the top-k: the bitonic merge when several of its keys enter, single insertions otherwise
// (a merge costs ~14 (WIDE: ~28) lane exchanges whatever enters; an insertion ~150 cycles per entering key)
template <bool WIDE>
__device__ __forceinline__ void topk_offer_sorted(WaveTopK& t, uint64_t la, uint64_t lb, uint64_t& tau, int k, int lane) {
  const int entering = __popcll(__ballot(la > tau)) + (WIDE ? __popcll(__ballot(lb > tau)) : 0);
  if (entering == 0) return;
  if (entering >= (WIDE ? 10 : 6)) {
    topk_merge_sorted<WIDE>(t, la, lb, lane);
    const uint64_t kth = topk_threshold<WIDE>(t, k);
    if (kth > tau) tau = kth;
    return;
  }
  if (__ballot(la > tau)) topk_offer<WIDE>(t, la, tau, k, lane);
  if (WIDE && __ballot(lb > tau)) topk_offer<WIDE>(t, lb, tau, k, lane);
}

// Per-query threshold shared between the wavefronts working on one query (one u64 per query in HBM, zeroed
// per launch). A wave whose list is full publishes its k-th best with an atomic max; every wave folds the
// published value into its own entry threshold. Only keys that provably cannot reach the final top-k are
// dropped, so results stay exact and independent of timing.
struct SharedTau {
  unsigned long long* slot;
  uint64_t published = 0;
  // issue the (L1-bypassing) read early, fold() it later: the latency hides behind the work in between
  __device__ __forceinline__ uint64_t peek() const { return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ void fold(uint64_t g, uint64_t& tau, uint64_t& floor) const {
    const uint64_t gu = ((uint64_t)(uint32_t)readfirstlane((int)(uint32_t)(g >> 32)) << 32) | (uint32_t)readfirstlane((int)(uint32_t)g);
    if (gu > floor) floor = gu;
    if (floor > tau) tau = floor;
  }
  __device__ __forceinline__ void publish_key(uint64_t kth, int lane) {  // kth: wave-uniform, 0 = nothing to say yet
    if (kth > published) {
      if (lane == 0) atomicMax(slot, (unsigned long long)kth);
      published = kth;
    }
  }
  template <bool WIDE>
  __device__ __forceinline__ void publish(const WaveTopK& t, int k, int lane) {
    publish_key(topk_threshold<WIDE>(t, k), lane);  // 0 until the list holds k keys
  }
};

}  // namespace rgpu
