// Materialising decode of whole terms: one wavefront per chunk of 128-posting blocks (+ VInt tail / singleton),
// docs and freqs written to HBM. GPU counterpart of BlockDocIterator::{refill_docs, next}
// (codec/postings/posting_reader.rs:501-561, 612-647) driven to exhaustion, and the block-decode microbenchmark.
// Also k_advance: BlockDocIterator::advance (posting_reader.rs:649-789) for independent probes.
#pragma once
#include "decode.hpp"
#include "types.hpp"

namespace rgpu {

#ifndef RGPU_DECODE_DEPTH
#define RGPU_DECODE_DEPTH 3
#endif
constexpr int DECODE_PREFETCH_DEPTH = RGPU_DECODE_DEPTH;
#ifndef RGPU_DECODE_BURST  // blocks decoded into registers before their stores go out together (0 / 1: block by block). Measured on
#define RGPU_DECODE_BURST 4  // the 100 M-doc shard, one box, same session: 0.735 ms block by block, 0.656 (2), 0.584 (4), 0.611 (8)
#endif
// (Round 4: the burst as 16-byte stores — values transposed through the slab so that every lane holds four consecutive output
// dwords, the term's misalignment folded into the LDS index — measured SLOWER: 0.057 vs 0.053 ms at 10 M docs, 0.726 vs 0.694
// at 100 M on one box; the same for k_prepare_blocks' fused output, 1.23 vs 1.14 ms. The compiler already merges a lane's two
// dwords into one 8-byte store; the stream is not issue-bound, and the LDS round trip costs more than the wider stores save.
// scripts/experiments/stores_x4.patch.)
#ifndef RGPU_DECODE_PLAIN_STORES
#define RGPU_DECODE_PLAIN_STORES 0
#endif
constexpr int WG_THREADS = 256;
constexpr int WG_WAVES = WG_THREADS / 64;

// largest t in [0, n) with prefix[t] <= x   (prefix[0] == 0, prefix is non-decreasing, x < prefix[n])
__device__ __forceinline__ int upper_slot(const int64_t* __restrict__ prefix, int n, int64_t x) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// largest t in [0, n) with prefix[t] <= x, searched by the whole wavefront: 64 probes per step — two dependent loads for 1024
// queries instead of ten, three for the 150 k terms of a big prepare / decode call instead of eighteen (at ~1.5 us per
// dependent load under load, the one-lane search was most of what k_decode_terms / k_prepare_blocks waited for)
__device__ __forceinline__ int upper_slot_wave(const int64_t* __restrict__ prefix, int n, int64_t x, int lane) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const bool ok = idx < hi && prefix[idx] <= x;  // true on a prefix of the lanes, lane 0 included
    const int cnt = __popcll(__ballot(ok));
    lo += (cnt - 1) * step;
    hi = min(hi, lo + step);
  }
  return lo;
}

// Items = (term, chunk of `blocks_per_item` blocks); the last chunk of a term also decodes its VInt tail or
// singleton. Directory entries of a chunk come in with one coalesced load and block i+1's payload rows are in
// flight while block i is unpacked and stored.
template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_decode_terms(SegView seg, const DevTerm* __restrict__ terms,
                                                             const int64_t* __restrict__ item_prefix,
                                                             const int64_t* __restrict__ out_prefix, int n_terms,
                                                             int64_t n_items, int blocks_per_item,
                                                             int32_t* __restrict__ docs_out,
                                                             int32_t* __restrict__ freqs_out) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (item >= n_items) return;
  const int t = upper_slot_wave(item_prefix, n_terms, item, lane);
  const int chunk = (int)(item - item_prefix[t]);
  const DevTerm T = terms[t];
  const int64_t out = out_prefix[t];
  uint8_t* slab = slabs[wave];
  const int b0 = chunk * blocks_per_item;
  const int b1 = min(T.nblocks, b0 + blocks_per_item);
  int32_t base = b0 == 0 ? 0 : seg.dir_last[T.dir_base + b0 - 1];
  const uint8_t* term_rows = seg.bstore + T.bs_base;
#if RGPU_DECODE_BURST > 1
  // Blocks go through in groups of RGPU_DECODE_BURST: decoded into registers one after the other, then stored back to
  // back — 2 KB of doc ids and 2 KB of freqs leave the wavefront within a few hundred cycles instead of 512 B every few
  // thousand, which is what the DRAM pages behind the output want (thousands of wavefronts stream into regions 8 KB apart)
  constexpr int NB = RGPU_DECODE_BURST;
  int b_done = b0;
  for (int c0 = b0; c0 + NB <= b1; c0 += 64) {
    const int nb = min(64, b1 - c0) / NB * NB;  // whole groups of this 64-entry directory chunk
    DirChunk dir;
    dir.load(seg.dir_row, seg.dir_hdr, T.dir_base, c0, nb, lane);
    const int last = nb - 1;
    uint4 ring[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(min(j, last))), dir.hdr_at(min(j, last)), lane);
    for (int i = 0; i < nb; i += NB) {
      int32_t D0[NB], D1[NB];
      uint32_t F0[NB], F1[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const uint4 rows = ring[j];
        const int pj = min(i + j + NB, last);  // clamped, not guarded: a redundant reload beats a load behind a branch
        ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(pj)), dir.hdr_at(pj), lane);
        const BlockPair bp = block_rows_decode<LEGACY>(rows, dir.hdr_at(i + j), slab, lane);
        deltas_to_docs(bp.d0, bp.d1, base, D0[j], D1[j]);
        base = readlane(D1[j], 63);
        F0[j] = bp.f0; F1[j] = bp.f1;
      }
      const int64_t o = out + 128 * (int64_t)(c0 + i) + 2 * lane;
#if RGPU_DECODE_PLAIN_STORES
#pragma unroll
      for (int j = 0; j < NB; ++j) { docs_out[o + 128 * j] = D0[j]; docs_out[o + 128 * j + 1] = D1[j]; }
#pragma unroll
      for (int j = 0; j < NB; ++j) { freqs_out[o + 128 * j] = (int32_t)F0[j]; freqs_out[o + 128 * j + 1] = (int32_t)F1[j]; }
#else
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        __builtin_nontemporal_store(D0[j], docs_out + o + 128 * j);
        __builtin_nontemporal_store(D1[j], docs_out + o + 128 * j + 1);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        __builtin_nontemporal_store((int32_t)F0[j], freqs_out + o + 128 * j);
        __builtin_nontemporal_store((int32_t)F1[j], freqs_out + o + 128 * j + 1);
      }
#endif
    }
    b_done = c0 + nb;
    if (nb < 64) break;
  }
  // (what is left: fewer than a group's blocks at the end of the chunk)
  stream_blocks<LEGACY, false, 1>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, nullptr, b_done, b1, slab, lane, base,
                        [&](int blk, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t, uint32_t) {
                          const int64_t o = out + 128 * (int64_t)blk + 2 * lane;
                          __builtin_nontemporal_store(d0, docs_out + o);
                          __builtin_nontemporal_store(d1, docs_out + o + 1);
                          __builtin_nontemporal_store((int32_t)f0, freqs_out + o);
                          __builtin_nontemporal_store((int32_t)f1, freqs_out + o + 1);
                        });
#else
  stream_blocks<LEGACY, false, DECODE_PREFETCH_DEPTH>(term_rows, seg.dir_row, seg.dir_hdr, T.dir_base, nullptr, b0, b1, slab, lane, base,
                        [&](int blk, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t, uint32_t) {
                          const int64_t o = out + 128 * (int64_t)blk + 2 * lane;
                          // streaming output, never re-read by this launch: nontemporal stores keep it from
                          // churning the L2 (measured 0.84-0.92 ms vs 1.07-1.21 ms on the 324 M-posting launch)
                          __builtin_nontemporal_store(d0, docs_out + o);
                          __builtin_nontemporal_store(d1, docs_out + o + 1);
                          __builtin_nontemporal_store((int32_t)f0, freqs_out + o);
                          __builtin_nontemporal_store((int32_t)f1, freqs_out + o + 1);
                        });
#endif
  if (b1 == T.nblocks) {
    if (T.df == 1) {
      if (lane == 0) { docs_out[out] = T.singleton_doc; freqs_out[out] = T.singleton_freq; }
    } else if (T.tail_n > 0) {
      int32_t d0, d1;
      uint32_t f0, f1;
      tail_load(term_rows, seg.dir_row[T.dir_base + T.nblocks], lane, d0, d1, f0, f1);  // decoded and validated at prepare time
      const int64_t o = out + 128 * (int64_t)T.nblocks + 2 * lane;
      if (2 * lane < T.tail_n) { docs_out[o] = d0; freqs_out[o] = (int32_t)f0; }
      if (2 * lane + 1 < T.tail_n) { docs_out[o + 1] = d1; freqs_out[o + 1] = (int32_t)f1; }
    }
  }
}

// first directory slot whose last doc >= target, in [0, nblocks]; nblocks == "the tail (or nothing)"
__device__ __forceinline__ int find_block(const int32_t* __restrict__ dir_last, uint32_t dir_base, int nblocks, int32_t target) {
  int lo = 0, hi = nblocks;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (dir_last[dir_base + mid] >= target) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// One wavefront per probe: binary search over the block directory (what skip_to + seek achieve, skip_reader.rs:
// 554-584), decode that block, then the in-block position of the first doc >= target as a count-of-less-than
// over the 128 docs (simd_block_decoder.rs:100-128) via ballot.
template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_advance(SegView seg, DevTerm T, const int32_t* __restrict__ targets,
                                                        int64_t n_targets, int32_t* __restrict__ out_docs,
                                                        int32_t* __restrict__ out_freqs) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t probe = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (probe >= n_targets) return;
  const int32_t target = targets[probe];
  uint8_t* slab = slabs[wave];
  int32_t res_doc = 0x7fffffff, res_freq = 0;
  if (T.df == 1) {
    if (T.singleton_doc >= target) { res_doc = T.singleton_doc; res_freq = T.singleton_freq; }
  } else {
    int blk = find_block(seg.dir_last, T.dir_base, T.nblocks, target);
    // a block whose directory entry is the sentinel (df % 128 == 0) may turn out not to hold the target
    for (; blk <= T.nblocks; ++blk) {
      int32_t d0, d1;
      uint32_t f0, f1;
      int n;
      if (blk < T.nblocks) {
        const int32_t base = blk == 0 ? 0 : seg.dir_last[T.dir_base + blk - 1];
        const BlockPair bp = decode_block<LEGACY>(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + blk],
                                                   seg.dir_hdr[T.dir_base + blk], slab, lane);
        deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
        f0 = bp.f0; f1 = bp.f1; n = 128;
      } else {
        if (T.tail_n == 0) break;
        tail_load(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + T.nblocks], lane, d0, d1, f0, f1);
        n = T.tail_n;
      }
      const bool lt0 = 2 * lane < n && d0 < target;
      const bool lt1 = 2 * lane + 1 < n && d1 < target;
      const int pos = __popcll(__ballot(lt0)) + __popcll(__ballot(lt1));  // docs are sorted: count == index
      if (pos < n) {
        const int src = pos >> 1;
        res_doc = (pos & 1) ? readlane(d1, src) : readlane(d0, src);
        res_freq = (pos & 1) ? readlane((int)f1, src) : readlane((int)f0, src);
        break;
      }
    }
  }
  if (lane == 0) { out_docs[probe] = res_doc; out_freqs[probe] = res_freq; }
}

}  // namespace rgpu
