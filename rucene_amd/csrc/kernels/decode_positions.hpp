// Materialising decode of a term's POSITIONS: the GPU counterpart of BlockPostingIterator::{next, next_position} driven to
// exhaustion (codec/postings/posting_reader.rs:1285-1324 refill_positions, :1357-1380 next_position, :1400-1437 next) — every
// position of every doc, doc after doc, a doc's `freq` positions ascending, into one i32 array. The decode surface of the
// ".pos" stream (the phrase kernels, search_phrase.hpp, only ever fetch the positions of one candidate doc) and the parity /
// throughput probe that rgpu_decode_terms is for ".doc". Fields that store payloads or offsets included: their position
// blocks are the plain ones, their trailing VInt block is walked past the payload bytes (decode_vint_block_everything).
//
// Work item = one directory slot of a term: a FullBlock of 128 docs, the prepared tail, or a singleton. Where the position
// stream stands at the slot's first doc is in the directory (dir_pos: the skip entry's posFP / posBufferUpto); where the slot's
// positions go in the output is the exclusive prefix sum of the slots' position counts (sum of the block's freqs) over the
// call: k_pos_counts -> the prepare path's scan kernels -> k_decode_positions. Inside a slot the wavefront walks the position
// blocks in order, 128 deltas at a time (two per lane), and turns deltas into positions with a SEGMENTED prefix sum — a doc's
// first delta is its first position (posting_writer.rs:363-380: last_position restarts at 0 with every doc): heads are
// scattered into a 128-bit mask from the docs' start indices, and since deltas are >= 0 the prefix sum at the latest head is a
// running maximum — position[i] = P[i] - max over heads h <= i of Pexcl[h] (or P[i] + the carry of a doc that began in an
// earlier window).
#pragma once
#include "search_phrase.hpp"

namespace rgpu {

// inclusive running maximum over the 64 lanes (values >= -1; -1 = "none")
__device__ __forceinline__ int wave_incl_max_scan(int v) {
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false));
  return v;
}

// the freqs of a directory slot's docs, two per lane (0 past the slot's end); returns the number of docs
template <bool LEGACY>
__device__ __forceinline__ int slot_freqs(const SegView& seg, const DevTerm& T, int blk, uint8_t* slab, int lane, uint32_t& f0, uint32_t& f1) {
  if (T.df == 1) {
    f0 = lane == 0 ? (uint32_t)T.singleton_freq : 0u;
    f1 = 0u;
    return 1;
  }
  if (blk < T.nblocks) {
    const BlockPair bp = decode_block<LEGACY>(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + blk], seg.dir_hdr[T.dir_base + blk], slab, lane);
    f0 = bp.f0; f1 = bp.f1;
    return 128;
  }
  int32_t d0, d1;
  tail_load(seg.bstore + T.bs_base, seg.dir_row[T.dir_base + T.nblocks], lane, d0, d1, f0, f1);
  if (2 * lane >= T.tail_n) f0 = 0u;
  if (2 * lane + 1 >= T.tail_n) f1 = 0u;
  return T.tail_n;
}

// items = directory slots of the call's terms (item_prefix per term): counts[item] = positions of the slot's docs
template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_pos_counts(SegView seg, const DevTerm* __restrict__ terms, const int64_t* __restrict__ item_prefix,
                                                           int n_terms, int64_t n_items, uint32_t* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  const int lane = lane_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave_id();
  if (item >= n_items) return;
  const int t = upper_slot_wave(item_prefix, n_terms, item, lane);
  const DevTerm T = terms[t];
  uint32_t f0, f1;
  (void)slot_freqs<LEGACY>(seg, T, (int)(item - item_prefix[t]), slabs[wave_id()], lane, f0, f1);
  const int total = wave_reduce_add((int)(f0 + f1));
  if (lane == 0) counts[item] = (uint32_t)total;
}

// offsets[item]: the exclusive prefix sum of counts over the call = the slot's first position in positions_out
template <bool LEGACY>
__global__ __launch_bounds__(WG_THREADS) void k_decode_positions(SegView seg, const DevTerm* __restrict__ terms, const PosTerm* __restrict__ pterms,
                                                                 const int64_t* __restrict__ item_prefix, int n_terms, int64_t n_items,
                                                                 const uint32_t* __restrict__ offsets, int64_t pos_len,
                                                                 int32_t* __restrict__ positions_out, int* err) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[WG_WAVES][SLAB_BYTES];
  __shared__ uint32_t heads_all[WG_WAVES][4];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * WG_WAVES + wave;
  if (item >= n_items || *err != 0) return;  // (the scan flags freqs that add up to more than total_term_freq: nothing may be written then)
  const int t = upper_slot_wave(item_prefix, n_terms, item, lane);
  const DevTerm T = terms[t];
  const PosTerm P = pterms[t];
  const int blk = (int)(item - item_prefix[t]);
  uint8_t* slab = slabs[wave];
  uint32_t* heads = heads_all[wave];
  uint32_t f0, f1;
  (void)slot_freqs<LEGACY>(seg, T, blk, slab, lane, f0, f1);
  // where each doc's positions start, counted from the slot's first position
  const int pair = (int)(f0 + f1);
  const int incl = wave_incl_scan(pair);
  const int s0 = incl - pair, s1 = s0 + (int)f0;  // doc 2 lane / 2 lane + 1
  const int total = readlane(incl, 63);
  if (total == 0) return;
  // the position stream at the slot's first doc (slot 0: the term's start)
  int64_t fp = (int64_t)P.pos_start_fp;
  int skip = 0;
  if (T.df > 1) {
    const uint64_t st = seg.dir_pos[T.dir_base + blk];
    fp += (int64_t)(uint32_t)st;
    skip = (int)(st >> 32);
  }
  int32_t* out = positions_out + offsets[item];
  auto give_up = [&](int code) { if (lane == 0) atomicMin(err, code); };
  int got = 0;      // positions of the slot written so far
  int32_t carry = 0;  // the position reached by the doc that runs across the window's start
  while (got < total) {
    uint32_t x0, x1;
    int nvals = 128;
    if (fp < 0 || fp + 2 > pos_len) { give_up(-4); return; }
    if (fp == P.last_pos_block_fp) {
      nvals = (int)(P.total_term_freq % 128);
      if (seg.pos_tail_flags == 0) decode_vint_block(seg.pos + fp, slab, lane, x0, x1);
      else if (!decode_vint_block_everything(seg.pos + fp, pos_len - fp, nvals, seg.pos_tail_flags, slab, lane, x0, x1)) { give_up(-4); return; }
      fp = -2;  // nothing follows the trailing block
    } else {
      const uint32_t b = seg.pos[fp];
      if (b > 32u) { give_up(-4); return; }
      if (b == 0) {
        int vlen;
        x0 = x1 = read_vint_uniform(seg.pos + fp + 1, &vlen);
        fp += 1 + vlen;
      } else {
        if (lane < 32) *reinterpret_cast<uint4*>(slab + 16 * lane) = load16_unaligned(seg.pos + fp + 1 + 16 * lane);
        wave_sync();
        extract_pair<LEGACY>(slab, (int)b, lane, x0, x1);
        wave_sync();
        fp += 1 + 16 * (int64_t)b;
      }
    }
    const int take = min(nvals - skip, total - got);
    if (take <= 0) { give_up(-4); return; }  // the stream ends before the slot's positions do
    // heads: docs whose first position lies in this window [got, got + take) -> window index start - got + skip
    if (lane < 4) heads[lane] = 0u;
    wave_sync();
    if (f0 != 0u && s0 >= got && s0 < got + take) atomicOr(&heads[(s0 - got + skip) >> 5], 1u << ((s0 - got + skip) & 31));
    if (f1 != 0u && s1 >= got && s1 < got + take) atomicOr(&heads[(s1 - got + skip) >> 5], 1u << ((s1 - got + skip) & 31));
    wave_sync();
    const int i0 = 2 * lane, i1 = 2 * lane + 1;
    const bool in0 = i0 >= skip && i0 < skip + take, in1 = i1 >= skip && i1 < skip + take;
    const uint32_t hw = heads[lane >> 4];
    const bool h0 = in0 && ((hw >> (i0 & 31)) & 1u), h1 = in1 && ((hw >> (i1 & 31)) & 1u);
    const int d0 = in0 ? (int)x0 : 0, d1 = in1 ? (int)x1 : 0;
    const int pr = d0 + d1;
    const int pin = wave_incl_scan(pr);
    const int p1 = pin, p0 = pin - d1;            // inclusive prefix sums at i0, i1
    const int e0 = p0 - d0, e1 = p0;              // exclusive ones
    // the prefix sum just in front of the latest head at or before each index (-1: none in this window yet)
    const int m_pair = h1 ? e1 : (h0 ? e0 : -1);  // the later head of the pair wins (prefix sums do not fall)
    const int m_incl = wave_incl_max_scan(m_pair);
    const int m_before = __builtin_amdgcn_update_dpp(-1, m_incl, 0x138, 0xf, 0xf, false);  // wave_shr:1 — the lanes in front
    const int b0 = h0 ? e0 : m_before;
    const int b1 = h1 ? e1 : b0;
    const int32_t v0 = b0 >= 0 ? p0 - b0 : p0 + carry;
    const int32_t v1 = b1 >= 0 ? p1 - b1 : p1 + carry;
    if (in0) __builtin_nontemporal_store(v0, out + got + i0 - skip);
    if (in1) __builtin_nontemporal_store(v1, out + got + i1 - skip);
    // the position the window's last value reached: the start of the next window's first doc if that doc goes on
    {
      const int last = skip + take - 1;
      carry = (last & 1) ? readlane(v1, last >> 1) : readlane(v0, last >> 1);
    }
    wave_sync();  // heads[] is rewritten by the next window
    got += take;
    skip = 0;
  }
}

}  // namespace rgpu
