// Conjunction (AND) on the GPU, lead-driven: the data-parallel form of ConjunctionScorer's leapfrog
// (search/scorer/conjunction_scorer.rs:44-83). One wavefront owns a chunk of the cheapest clause's blocks; each
// decoded lead block yields 128 sorted candidates, two per lane. For every other clause, in cost order, the wave
//   1. brackets the candidates in that clause's block directory (two wave-uniform binary searches, then a short
//      per-lane one) — what Lucene50SkipReader::skip_to does per probe (skip_reader.rs:554-584);
//   2. visits only the distinct blocks that hold at least one live candidate, decoding each once into LDS;
//   3. lets every lane whose candidate maps to that block binary-search the 128 decoded docs — the in-block
//      scan of BlockDocIterator::advance (posting_reader.rs:714-731).
// A candidate dies at the first clause that misses it, so later clauses touch fewer blocks. Scores are summed
// lead1, lead2, others... in f32 exactly as conjunction_scorer.rs:87-95 (the host sorts clauses by doc_freq).
#pragma once
#include "search.hpp"

namespace rgpu {

#ifdef RGPU_EXP_COUNT  // developer instrumentation (variant builds only): [0] lead blocks, [1] other-clause block decodes,
__device__ unsigned long long g_and_dbg[4];  // [2] probed candidates, [3] (lead block, clause) visits
#define AND_DBG(i, n) do { if (lane == 0) atomicAdd(&g_and_dbg[i], (unsigned long long)(n)); } while (0)
#else
#define AND_DBG(i, n) do {} while (0)
#endif

// The kernel is latency bound (dependent directory probes and block decodes): occupancy buys more than registers.
#ifndef RGPU_AND_WAVES
#define RGPU_AND_WAVES 8
#endif
constexpr int AND_WAVES_PER_SIMD = RGPU_AND_WAVES;
#ifndef RGPU_AND_SERIAL
#define RGPU_AND_SERIAL 8
#endif
constexpr int AND_SERIAL_PROBES = RGPU_AND_SERIAL;  // up to this many candidates per block are probed by broadcast

// first slot in [lo, hi] whose last doc >= target; slot `hi` is returned without being read
__device__ __forceinline__ int find_block_in(const int32_t* __restrict__ dir_last, uint32_t dir_base, int lo, int hi, int32_t target) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (dir_last[dir_base + mid] >= target) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Wave-cooperative form of the same search (target is wave-uniform): first a coalesced look at the 64 directory
// entries right after `from` — consecutive lead blocks probe monotonically, so the answer is usually there (one
// load instead of ~14 dependent ones) — then a 64-ary search: each round the lanes probe 64 evenly spaced
// entries and a ballot picks the sub-range. Returns the first slot in [from, nblocks] whose last doc >= target.
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

__device__ __forceinline__ int lds_lower_bound(const int32_t* a, int n, int32_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// HAS_NOT / HAS_OPT: some query of the launch carries MUST_NOT / optional SHOULD clauses (separate instantiations keep the
// common kernel lean). Clause order on the device: [MUST x n_terms][MUST_NOT x pad][SHOULD x (op >> 16)].
template <bool LEGACY, bool WIDE, bool HAS_NOT, bool HAS_OPT>
__global__ __launch_bounds__(WG_THREADS, AND_WAVES_PER_SIMD) void k_search_and(SegView seg, const DevQuery* __restrict__ queries,
                                                           const DevTerm* __restrict__ terms,
                                                           const int64_t* __restrict__ item_prefix, int n_queries,
                                                           int64_t n_items, int blocks_per_item, int k,
                                                           uint64_t* __restrict__ partial_keys,
                                                           int32_t* __restrict__ partial_counts,
                                                           unsigned long long* __restrict__ tau_slots) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ float caches[WG_WAVES][256];
  __shared__ int32_t bdocs[WG_WAVES][128];
  __shared__ uint32_t bfreqs[WG_WAVES][128];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (item >= n_items) return;
  const int q = upper_slot(item_prefix, n_queries, item);
  const int chunk = (int)(item - item_prefix[q]);
  const DevQuery Q = queries[q];
  const DevTerm L = terms[Q.first_term];
  uint8_t* slab = slabs[wave];
  float* cache = caches[wave];
  int32_t* bd = bdocs[wave];
  uint32_t* bf = bfreqs[wave];
  const bool has_norms = seg.norms != nullptr;
  int cur_table = -1;
  float k1 = 0.f;
  auto use_table = [&](int id) {
    if (id != cur_table) { load_sim_table(seg, id, cache, lane, k1); cur_table = id; }
  };

  WaveTopK top;
  uint64_t tau = 0, floor = 0;
  int count = 0;
  int cursor = 0;  // lane ti holds clause ti's directory cursor (a register array indexed by ti would spill)
  SharedTau shared{tau_slots + q};
  shared.fold(shared.peek(), tau, floor);

  // nb0 / nb1: the candidates' norm bytes — from the lead's posting-order norms for FullBlocks, gathered for
  // its tail; every other clause scores the same docs, so no clause ever gathers norms again
  auto intersect = [&](int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t nb0, uint32_t nb1, bool a0, bool a1) {
    a0 = a0 && doc_is_live(seg.live, d0);
    a1 = a1 && doc_is_live(seg.live, d1);
    use_table(L.sim_table);
    float wk = L.weight * (k1 + 1.0f);
    float s0 = bm25_score(wk, (float)(int32_t)f0, has_norms ? cache[nb0] : k1);
    float s1 = bm25_score(wk, (float)(int32_t)f1, has_norms ? cache[nb1] : k1);
    // clauses 1 .. n_terms-1 are required (MUST), the n_not after them prohibited (MUST_NOT: ReqNotScorer,
    // req_not_scorer.rs:47-63 — a candidate found there dies, nothing is scored), the n_opt after those optional
    // (SHOULD next to MUST: ReqOptScorer, req_opt_scorer.rs:41-66 — a candidate found there adds that clause's score to
    // a separate sum, as DisjunctionSumScorer does, which is added to the required sum at the end; a miss costs nothing.
    // The reference's sequential "skip the optional clause for low scorers after 100 docs" rule is NOT applied:
    // scores are the exact sums, >= the reference's)
    const int n_req_not = HAS_NOT ? Q.n_terms + Q.pad : Q.n_terms;
    const int n_clauses = HAS_OPT ? n_req_not + ((Q.op >> 16) & 0xff) : n_req_not;
    float r0 = 0.f, r1 = 0.f;  // required sums, parked while s0 / s1 collect the optional sum
    bool in_opt = false;
    for (int ti = 1; ti < n_clauses; ++ti) {
      const uint64_t m0 = __ballot(a0), m1 = __ballot(a1);
      if (!(m0 | m1)) break;
      const bool excl = HAS_NOT && ti >= Q.n_terms && ti < n_req_not;  // wave-uniform
      const bool opt = HAS_OPT && ti >= n_req_not;                      // wave-uniform
      if (HAS_OPT && opt && !in_opt) { r0 = s0; r1 = s1; s0 = 0.f; s1 = 0.f; in_opt = true; }
      const DevTerm T = terms[Q.first_term + ti];
      if (!excl) {
        use_table(T.sim_table);
        wk = T.weight * (k1 + 1.0f);
      }
      const float n0 = has_norms ? cache[nb0] : k1;
      const float n1 = has_norms ? cache[nb1] : k1;
      // what finding / missing a candidate in this clause means
      auto found = [&](bool& alive, float& s, uint32_t fq, float nrm) {
        if (excl) alive = false; else s += bm25_score(wk, (float)(int32_t)fq, nrm);
      };
      auto missed = [&](bool& alive) { if (!excl && !opt) alive = false; };
      if (T.df == 1) {
        if (a0) { if (d0 == T.singleton_doc) found(a0, s0, (uint32_t)T.singleton_freq, n0); else missed(a0); }
        if (a1) { if (d1 == T.singleton_doc) found(a1, s1, (uint32_t)T.singleton_freq, n1); else missed(a1); }
        continue;
      }
      // candidates are sorted across (lane, slot): first / last live candidate bracket the directory range
      const int fl0 = m0 ? __builtin_ctzll(m0) : 64, fl1 = m1 ? __builtin_ctzll(m1) : 64;
      const int32_t dmin = fl0 <= fl1 ? readlane(d0, fl0 & 63) : readlane(d1, fl1 & 63);
      const int ll0 = m0 ? 63 - __builtin_clzll(m0) : -1, ll1 = m1 ? 63 - __builtin_clzll(m1) : -1;
      const int32_t dmax = ll1 >= ll0 ? readlane(d1, ll1 & 63) : readlane(d0, ll0 & 63);
      // This clause's cursor only moves forward: lead blocks of an item arrive in doc order. Usual case: the 64
      // directory entries after the cursor (ONE coalesced load) bracket all candidates — they give lo, hi and,
      // parked in LDS, the per-candidate block search without any further trip to memory. Otherwise the general
      // wave-cooperative searches and a per-lane search in the directory itself.
      int lo, hi, blk0, blk1;
      AND_DBG(3, 1);
      {
        const int from = readlane(cursor, ti);
        const int p = from + lane;
        const int32_t v = p < T.nblocks ? seg.dir_last[T.dir_base + p] : 0x7fffffff;  // the virtual end slot bounds every doc
        const uint64_t mlo = __ballot(v >= dmin), mhi = __ballot(v >= dmax);
        if (from < T.nblocks && mhi) {  // mhi != 0 implies mlo != 0 (dmin <= dmax)
          lo = from + (int)__builtin_ctzll(mlo);
          hi = from + (int)__builtin_ctzll(mhi);
          const int span = hi - lo;  // slots lo .. hi-1 are real and < dmax's slot; slot hi is the answer for the rest
          if (p >= lo && p < hi) bd[p - lo] = v;
          wave_sync();
          blk0 = a0 ? lo + lds_lower_bound(bd, span, d0) : 0x7fffffff;
          blk1 = a1 ? lo + lds_lower_bound(bd, span, d1) : 0x7fffffff;
          wave_sync();
        } else {
          lo = find_block_wave(seg.dir_last, T.dir_base, from, T.nblocks, dmin, lane);
          hi = find_block_wave(seg.dir_last, T.dir_base, lo, T.nblocks, dmax, lane);
          blk0 = a0 ? find_block_in(seg.dir_last, T.dir_base, lo, hi, d0) : 0x7fffffff;
          blk1 = a1 ? find_block_in(seg.dir_last, T.dir_base, lo, hi, d1) : 0x7fffffff;
        }
      }
      cursor = lane == ti ? lo : cursor;
      bool p0 = a0, p1 = a1;
      // first pending candidate's block (candidates are sorted across (lane, slot)); INT_MAX when none is pending
      auto first_pending = [&]() -> int {
        const uint64_t q0 = __ballot(p0), q1 = __ballot(p1);
        if (!(q0 | q1)) return 0x7fffffff;
        const int l0 = q0 ? __builtin_ctzll(q0) : 64, l1 = q1 ? __builtin_ctzll(q1) : 64;
        return l0 <= l1 ? readlane(blk0, l0 & 63) : readlane(blk1, l1 & 63);
      };
      auto probe = [&](bool c0, bool c1, int n_in) {  // lanes whose candidate maps to the block now in bd / bf
        if (c0) {
          const int pos = lds_lower_bound(bd, n_in, d0);
          if (pos < n_in && bd[pos] == d0) found(a0, s0, bf[pos], n0); else missed(a0);
        }
        if (c1) {
          const int pos = lds_lower_bound(bd, n_in, d1);
          if (pos < n_in && bd[pos] == d1) found(a1, s1, bf[pos], n1); else missed(a1);
        }
      };
      int cur = first_pending();
      // FullBlocks, software-pipelined: the rows (and directory words) of the NEXT distinct block are requested
      // before the current one is decoded and probed — the kernel is bound by this dependent-load chain
      if (cur < T.nblocks) {
        const uint8_t* term_rows = seg.bstore + T.bs_base;
        struct Fetched { uint4 rows; uint32_t hdr; int32_t base; };
        auto fetch = [&](int b) -> Fetched {
          Fetched f;
          f.hdr = seg.dir_hdr[T.dir_base + b];
          f.base = b == 0 ? 0 : seg.dir_last[T.dir_base + b - 1];
          f.rows = block_rows_load(block_rows_at(term_rows, seg.dir_row[T.dir_base + b]), f.hdr, lane);
          return f;
        };
        Fetched A = fetch(cur);
        while (true) {
          const bool c0 = p0 && blk0 == cur, c1 = p1 && blk1 == cur;
          p0 = p0 && !c0;
          p1 = p1 && !c1;
          const int nxt = first_pending();
          const bool more = nxt < T.nblocks;
          const Fetched B = fetch(more ? nxt : cur);  // unconditional: a load behind a branch would serialise the two
          const BlockPair bp = block_rows_decode<LEGACY>(A.rows, A.hdr, slab, lane);
          AND_DBG(1, 1);
          int32_t e0, e1;
          deltas_to_docs(bp.d0, bp.d1, A.base, e0, e1);
          const uint64_t k0 = __ballot(c0), k1m = __ballot(c1);
          AND_DBG(2, __popcll(k0) + __popcll(k1m));
          if (__popcll(k0) + __popcll(k1m) <= AND_SERIAL_PROBES) {
            // a few candidates (the usual case from the second clause on): broadcast each one and let the 128
            // decoded docs, still in registers, answer with two compares — no LDS round trips, no per-lane search
            auto serial = [&](uint64_t km, const int32_t dcand, float& s, bool& alive, float nrm) {
              while (km) {
                const int j = __builtin_ctzll(km);
                km &= km - 1;
                const int32_t d = readlane(dcand, j);
                const uint64_t h0 = __ballot(e0 == d), h1 = __ballot(e1 == d);
                if (h0 | h1) {
                  const uint32_t fq = h0 ? (uint32_t)readlane((int)bp.f0, __builtin_ctzll(h0)) : (uint32_t)readlane((int)bp.f1, __builtin_ctzll(h1));
                  if (lane == j) found(alive, s, fq, nrm);
                } else if (lane == j) {
                  missed(alive);
                }
              }
            };
            serial(k0, d0, s0, a0, n0);
            serial(k1m, d1, s1, a1, n1);
          } else {
            bd[2 * lane] = e0; bd[2 * lane + 1] = e1;
            bf[2 * lane] = bp.f0; bf[2 * lane + 1] = bp.f1;
            wave_sync();
            probe(c0, c1, 128);
            wave_sync();
          }
          cur = nxt;
          if (!more) break;
          A = B;
        }
      }
      if (cur != 0x7fffffff) {  // candidates past the last FullBlock: the VInt tail, or nothing
        int n_in = 0;
        if (T.tail_n > 0) {
          const uint32_t toff = T.nblocks ? seg.dir_off[T.dir_base + T.nblocks] : 0u;
          const int32_t base = T.nblocks ? seg.dir_last[T.dir_base + T.nblocks - 1] : 0;
          int32_t e0, e1;
          uint32_t g0, g1;
          decode_tail(seg.doc + T.start_fp + toff, T.tail_n, base, slab, lane, e0, e1, g0, g1);
          bd[2 * lane] = e0; bd[2 * lane + 1] = e1;
          bf[2 * lane] = g0; bf[2 * lane + 1] = g1;
          n_in = T.tail_n;
        }
        wave_sync();
        probe(p0, p1, n_in);
        wave_sync();
      }
    }
    if (HAS_OPT && in_opt) { s0 = r0 + s0; s1 = r1 + s1; }
    count += __popcll(__ballot(a0)) + __popcll(__ballot(a1));
    topk_offer<WIDE>(top, a0 ? make_key(s0, d0) : 0ull, tau, k, lane, floor);
    topk_offer<WIDE>(top, a1 ? make_key(s1, d1) : 0ull, tau, k, lane, floor);
  };

  const int b0 = chunk * blocks_per_item;
  const int b1 = min(L.nblocks, b0 + blocks_per_item);
  int32_t base = b0 == 0 ? 0 : seg.dir_last[L.dir_base + b0 - 1];
  for (int blk = b0; blk < b1; ++blk) {
    uint32_t nn = 0;
    if (has_norms) nn = *reinterpret_cast<const uint16_t*>(seg.pnorm + L.pn_base + 128 * (size_t)blk + 2 * lane);
    const BlockPair bp = decode_block<LEGACY>(seg.bstore + L.bs_base, seg.dir_row[L.dir_base + blk], seg.dir_hdr[L.dir_base + blk], slab, lane);
    int32_t d0, d1;
    deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
    base = readlane(d1, 63);
    AND_DBG(0, 1);
    intersect(d0, d1, bp.f0, bp.f1, nn & 0xffu, nn >> 8, true, true);
  }
  if (b1 == L.nblocks) {
    if (L.df == 1) {
      const bool a0 = lane == 0;
      intersect(L.singleton_doc, L.singleton_doc, (uint32_t)L.singleton_freq, 0u, (has_norms && a0) ? seg.norms[L.singleton_doc] : 0u, 0u,
                a0, false);
    } else if (L.tail_n > 0) {
      const uint32_t toff = L.nblocks ? seg.dir_off[L.dir_base + L.nblocks] : 0u;
      int32_t d0, d1;
      uint32_t f0, f1;
      decode_tail(seg.doc + L.start_fp + toff, L.tail_n, base, slab, lane, d0, d1, f0, f1);
      const bool a0 = 2 * lane < L.tail_n, a1 = 2 * lane + 1 < L.tail_n;
      intersect(d0, d1, f0, f1, (has_norms && a0) ? seg.norms[d0] : 0u, (has_norms && a1) ? seg.norms[d1] : 0u, a0, a1);
    }
  }
  shared.publish<WIDE>(top, k, lane);
  uint64_t* pk = partial_keys + (size_t)item * (size_t)k;
  if (lane < k) pk[lane] = top.a;
  if (WIDE && lane + 64 < k) pk[lane + 64] = top.b;
  if (lane == 0) partial_counts[item] = count;
}

}  // namespace rgpu
