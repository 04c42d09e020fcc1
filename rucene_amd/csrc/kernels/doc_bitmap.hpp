// Doc bitmaps of a segment's densest terms — the prepared structure behind k_or_lazy (search_or_lazy.hpp).
//
// A term that holds at least one doc in `rgpu_config.or_bitmap_density` (default 64) gets, once, next to its block
// directory:
//   words[2 * (ceil(max_doc / 32) + pad)]   pairs {any, hi} per 32 docs: bit d & 31 of any = "the term's list holds doc d", of
//                                     hi = "... and its posting's freq / (freq + cache[norm]) is >= BITMAP_HI_CUT" under the
//                                     similarity table the bitmap was built with — a one-bit sketch of the posting's score:
//                                     a posting outside `hi` scores below weight * (k1 + 1) * BITMAP_HI_CUT whatever the weight
//                                     (zero padded: a window kernel reads whole windows past max_doc without a bounds test)
//   ranks[same length]                postings of the list before word w (exclusive prefix popcount): the posting index of
//                                     doc d is ranks[d >> 5] + popcount(words[d >> 5] below bit d & 31)
//   freqs[doc_freq + pad]             min(freq, 255) by posting index
//   ovf[2 * n_ovf]                    {posting index, freq} of the postings whose freq is >= 255 (Rucene clamps term freqs
//                                     to 10 when it writes an index; a Lucene-written one may hold such a posting)
//   memb[ceil(max_doc / 32) + pad]    the `any` bits alone, one per doc: what a conjunction's FIRST probe needs of the list — "is the
//                                     candidate in it" for every candidate of a lead block. A gather moves a 64-byte sector
//                                     whatever it needs of it: the sector then covers 512 docs instead of the 128 of `nib`, so a
//                                     dense lead's neighbouring candidates share it and the probed footprint is max_doc / 8 bytes
//                                     per list (k_search_and: 0.284 ms probing nib first, 0.274 probing `words`, at 10 M docs;
//                                     2.27 vs 2.04 ms at 100 M); the survivors (a sixth of the candidates) then ask nib for the freq
//   nib[ceil(max_doc / 8) + pad]      only for a term that holds at least one doc in BITMAP_NIBBLE_DENSITY: four bits per doc —
//                                     0 = the list does not hold the doc, 1..14 = its freq, 15 = "look the freq up through
//                                     ranks / freqs". Membership AND freq in one 4-byte gather instead of a word + rank
//                                     gather followed by a freq gather: the conjunction kernel's two memory round trips per
//                                     clause become one (max_doc / 2 bytes per term)
// (the sketch only sharpens a bound: a clause that names the term under another similarity table treats every posting as hi)
// i.e. a random-access view of (doc -> freq) for the lists that hold ~90 % of a Zipfian disjunction's postings: "is doc d
// in the list" is one bit, and a whole window's membership is one coalesced read of 4 bytes per 32 docs.
// The reference has no such structure (its DisjunctionSumScorer walks every sub-scorer posting by posting,
// search/scorer/disjunction_scorer.rs:24-104); results are unchanged by it — see search_or_lazy.hpp for the argument.
#pragma once
#include "wave.hpp"

namespace rgpu {

constexpr int BITMAP_OVF_CAP = 4096;   // more postings with freq >= 255 than this: the term gets no bitmap
constexpr float BITMAP_HI_CUT = 0.72f;  // ~ the top fifth of a BM25 list (k1 1.2, b 0.75): freq >= 3 in a doc of average length
#ifndef RGPU_NIBBLE_DENSITY
#define RGPU_NIBBLE_DENSITY 128  // (3-term batch, k_search_and: 32 -> 0.484 ms, 128 -> 0.474, 512 -> 0.484; without the array 0.505)
#endif
constexpr int BITMAP_NIBBLE_DENSITY = RGPU_NIBBLE_DENSITY;  // a term holding >= 1 doc in this many also gets the four-bits-per-doc array
constexpr int BITMAP_PAD_WORDS = 2048;  // zero words behind the last real one (>= the widest window of k_or_lazy in words)

struct BitmapStats {
  unsigned int n_ovf;     // postings with freq >= 255
  unsigned int max_freq;  // the list's largest freq
  unsigned int bad_docs;  // doc ids outside [0, max_doc): a corrupt list (FullBlocks are validated at prepare time: 0)
  unsigned int pad;
};

// one thread per posting of the decoded list. norms: the segment's norm bytes as the kernels see them (ranks or raw bytes);
// cache: the similarity table's 256 norm-cache floats; rank_to_norm: null when the segment holds raw norm bytes
__global__ __launch_bounds__(256) void k_bitmap_fill(const int32_t* __restrict__ docs, const int32_t* __restrict__ freqs, int64_t df,
                                                     int32_t max_doc, const uint8_t* __restrict__ norms, const float* __restrict__ cache,
                                                     const uint8_t* __restrict__ rank_to_norm, uint2* __restrict__ words,
                                                     uint8_t* __restrict__ freq8, uint32_t* __restrict__ ovf, BitmapStats* __restrict__ stats,
                                                     uint32_t* __restrict__ nib, uint32_t* __restrict__ memb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = i < df;
  const int32_t d = ok ? docs[i] : 0;
  const uint32_t f = ok ? (uint32_t)freqs[i] : 0u;
  const bool in = ok && (uint32_t)d < (uint32_t)max_doc;
  if (in) {
    atomicOr(&words[d >> 5].x, 1u << (d & 31));
    atomicOr(&memb[d >> 5], 1u << (d & 31));
    bool hi = true;  // no norms: every posting scores weight * (k1 + 1) * freq / (freq + k1) — no sketch, everything is "hi"
    if (norms != nullptr) {
      const uint32_t nb = norms[d];
      const float cv = cache[rank_to_norm != nullptr ? rank_to_norm[nb] : nb];
      const float qf = (float)(int32_t)f;
      hi = !(qf / (qf + cv) < BITMAP_HI_CUT);  // (a NaN — 0 / 0 — counts as hi: never under-estimate)
    }
    if (hi) atomicOr(&words[d >> 5].y, 1u << (d & 31));
    if (nib != nullptr) atomicOr(&nib[d >> 3], (f >= 1u && f <= 14u ? f : 15u) << (4 * (d & 7)));  // (a freq of 0 — corrupt — goes the long way too)
  }
  if (ok) freq8[i] = (uint8_t)(f < 255u ? f : 255u);
  if (ok && f >= 255u) {
    const unsigned int at = atomicAdd(&stats->n_ovf, 1u);
    if (at < (unsigned)BITMAP_OVF_CAP) { ovf[2 * at] = (uint32_t)i; ovf[2 * at + 1] = f; }
  }
  // one atomic per wavefront, not per posting (same-address atomics serialise)
  const uint32_t fmax = wave_reduce_max_u32(f);
  const uint64_t bad = __ballot(ok && !in);
  if (lane_id() == 0) {
    atomicMax(&stats->max_freq, fmax);
    if (bad) atomicAdd(&stats->bad_docs, (unsigned)__popcll(bad));
  }
}

// Membership bits ALONE for a list too sparse for a full bitmap (round 6): the conjunction kernel's batched first probe only asks
// "is the candidate in the list" (one bit per doc, max_doc / 8 bytes whatever the list's length); the few survivors then find their
// freq through the list's block directory. stats->bad_docs counts doc ids outside [0, max_doc) AND repeated ones (a bit that was
// already set): either makes the list unusable for the probe — its clause is simply walked.
__global__ __launch_bounds__(256) void k_bitmap_memb(const int32_t* __restrict__ docs, int64_t df, int32_t max_doc, uint32_t* __restrict__ memb,
                                                     BitmapStats* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = i < df;
  const int32_t d = ok ? docs[i] : 0;
  const bool in = ok && (uint32_t)d < (uint32_t)max_doc;
  bool dup = false;
  if (in) dup = (atomicOr(&memb[d >> 5], 1u << (d & 31)) >> (d & 31)) & 1u;
  const uint64_t bad = __ballot((ok && !in) || dup);
  if (lane_id() == 0 && bad) atomicAdd(&stats->bad_docs, (unsigned)__popcll(bad));
}

__global__ __launch_bounds__(256) void k_bitmap_popc(const uint2* __restrict__ words, int64_t n_words, uint32_t* __restrict__ ranks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) ranks[i] = (uint32_t)__popc(words[i].x);
}

}  // namespace rgpu
