// Device-side decoders for one 128-posting FullBlock (doc deltas + freqs) and for a VInt tail, one wavefront
// per block. Replaces, for the GPU path (paths relative to /root/reference/src/core):
//   codec/postings/for_util.rs:187-243            ForUtilInstance::read_block (header byte, all-equal, payload)
//   util/packed/packed_simd.rs:126-163            SIMD128Packer::unpack   (.doc version 1, BP128 vertical layout)
//   util/packed/packed_misc.rs:2655-2680,2829-2841 BulkOperationPacked / PackedSingleBlock::decode_byte_to_int (v0)
//   codec/postings/posting_reader.rs:308-333      read_vint_block (tail)
//   codec/postings/posting_reader.rs:622-646      the sequential `accum + delta` prefix sum in next()
//
// Lane mapping: lane t owns postings 2t and 2t+1 of the block. In the BP128 layout value i = 4r + l lives at
// bits [r*b, (r+1)*b) of stream l, stream word w at dword 4w + l, so postings (2t, 2t+1) share row r = t/2 and
// sit in adjacent streams l = 2(t&1), l+1: one ds_read_b64 fetches both low words, one more the straddle words.
// Payload rows (16 B each) come from the block store (SegView::bstore, 16-byte aligned copies made once per term by
// k_prepare_terms): one aligned global_load_dwordx4 per lane — lanes 0..31 take the doc stream, lanes 32..63 the
// freq stream — staged in a wave-private LDS slab. In the file itself a stream starts right after a 1-byte header,
// and byte-misaligned dwordx4 loads run at 1/3 of the aligned rate on gfx950 (scripts/microbench/unaligned_rows.hip:
// 34.6 vs 11.5 ns per block per CU), which bounded every scoring kernel; shifting the bytes into place in
// registers (v_alignbyte) or through misaligned LDS reads was measured too and costs more than it saves.
#pragma once
#include "wave.hpp"

namespace rgpu {

// wave-private LDS slab (bytes): two 528-byte stream areas, reused as byte / value scratch by the tail decoder
constexpr int SLAB_STREAM = 528;
constexpr int SLAB_BYTES = 2432;  // >= 1296 tail bytes + 1024 value words, multiple of 16
constexpr int TAIL_MAX_BYTES = 1280;
constexpr int DIR_SENTINEL_DOC = 0x7ffffffe;

// per-block directory header word: b_doc(6) | vint_len_doc(3) << 6 | b_freq(6) << 9
__device__ __host__ __forceinline__ int hdr_bdoc(uint32_t h) { return (int)(h & 63); }
__device__ __host__ __forceinline__ int hdr_vlen(uint32_t h) { return (int)((h >> 6) & 7); }
__device__ __host__ __forceinline__ int hdr_bfreq(uint32_t h) { return (int)((h >> 9) & 63); }

// bytes of a FullBlock as the .doc file frames it: two header bytes + each stream's payload (16 * b, or its VInt)
__device__ __host__ __forceinline__ uint32_t encoded_block_bytes(uint32_t hdr) {
  return 2u + (hdr_bdoc(hdr) ? 16u * (uint32_t)hdr_bdoc(hdr) : (uint32_t)hdr_vlen(hdr)) + (hdr_bfreq(hdr) ? 16u * (uint32_t)hdr_bfreq(hdr) : 1u);
}

struct __attribute__((packed, aligned(1))) UnalignedU4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) UnalignedU32 { uint32_t v; };

__device__ __forceinline__ uint4 load16_unaligned(const uint8_t* p) {
  UnalignedU4 v = *reinterpret_cast<const UnalignedU4*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t load4_unaligned(const uint8_t* p) {
  return reinterpret_cast<const UnalignedU32*>(p)->v;
}

// data_input.rs:78-111 read_vint, all lanes on the same address (broadcast load); *len receives the byte count
__device__ __forceinline__ uint32_t read_vint_uniform(const uint8_t* p, int* len) {
  uint32_t v = 0;
  int i = 0;
  for (; i < 5; ++i) {
    uint32_t b = p[i];
    v |= (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) { ++i; break; }
  }
  *len = i;
  return v;
}

// staged streams are dword streams (16-byte aligned in the slab)
__device__ __forceinline__ uint32_t lds_u32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
__device__ __forceinline__ uint2 lds_u2(const uint8_t* p) { return *reinterpret_cast<const uint2*>(p); }

// ---- BP128 (version 1) ----------------------------------------------------------------------------------------
__device__ __forceinline__ void extract_pair_bp128(const uint8_t* stream, int b, int lane, uint32_t& v0, uint32_t& v1) {
  const int r = lane >> 1;
  const int l = (lane & 1) << 1;
  const int p = r * b;
  const int w = p >> 5;
  const int s = p & 31;
  const uint8_t* at = stream + 4 * l + 16 * w;  // one address: the pair of reads becomes one ds_read2_b64
  const uint2 lo = lds_u2(at);
  const uint2 hi = lds_u2(at + 16);
  const uint32_t mask = 0xffffffffu >> (32 - b);  // callers pass 1 <= b <= 32 (b == 0 is the all-equal form)
  v0 = (uint32_t)((((uint64_t)hi.x << 32) | lo.x) >> s) & mask;
  v1 = (uint32_t)((((uint64_t)hi.y << 32) | lo.y) >> s) & mask;
}

// ---- legacy PackedInts (version 0) ----------------------------------------------------------------------------
// Packed: value i at bits [i*b, (i+1)*b) of one MSB-first big-endian stream; PackedSingleBlock (b in 1,2,4):
// big-endian u64 blocks holding 64/b values each, value j of a block at bits [j*b, (j+1)*b) from the LSB.
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
__device__ __forceinline__ uint32_t extract_packed_be(const uint8_t* stream, int b, int i) {
  const int p = i * b;
  const int w = p >> 5;
  const int s = p & 31;
  const uint64_t hi = bswap32(lds_u32(stream + 4 * w));
  const uint64_t lo = bswap32(lds_u32(stream + 4 * w + 4));
  const uint64_t win = (hi << 32) | lo;  // bits p.. start at the top
  const uint32_t mask = b >= 32 ? 0xffffffffu : ((1u << b) - 1u);
  return (uint32_t)(win >> (64 - s - b)) & mask;
}
__device__ __forceinline__ uint32_t extract_psb(const uint8_t* stream, int b, int i) {
  const int per = 64 / b;
  const int blk = i / per;
  const int j = i - blk * per;
  const uint64_t w = ((uint64_t)bswap32(lds_u32(stream + 8 * blk)) << 32) | bswap32(lds_u32(stream + 8 * blk + 4));
  return (uint32_t)(w >> (j * b)) & ((1u << b) - 1u);
}
__device__ __forceinline__ void extract_pair_legacy(const uint8_t* stream, int b, int lane, uint32_t& v0, uint32_t& v1) {
  if (b == 1 || b == 2 || b == 4) {  // FormatAndBits::fastest with COMPACT (packed_misc.rs:474-531)
    v0 = extract_psb(stream, b, 2 * lane);
    v1 = extract_psb(stream, b, 2 * lane + 1);
  } else {
    v0 = extract_packed_be(stream, b, 2 * lane);
    v1 = extract_packed_be(stream, b, 2 * lane + 1);
  }
}
template <bool LEGACY>
__device__ __forceinline__ void extract_pair(const uint8_t* stream, int b, int lane, uint32_t& v0, uint32_t& v1) {
  if (LEGACY) extract_pair_legacy(stream, b, lane, v0, v1);
  else extract_pair_bp128(stream, b, lane, v0, v1);
}

struct BlockPair {
  uint32_t d0, d1;  // doc deltas of postings 2*lane, 2*lane+1
  uint32_t f0, f1;  // freqs
};

// ---- the .doc file's own framing (prepare time only) ------------------------------------------------------------
// [hdr byte][doc payload 16*b | vint][hdr byte][freq payload 16*b | vint]; this lane's 16-byte row of its stream,
// fetched byte-exact (the slow TA path — once per block per term lifetime). Row 0 of an all-equal stream (b == 0)
// starts with the stream's VInt.
// (a docs-only field has no freq stream: its freq half reads bytes past the block, replaced by store_rows_from_file)
__device__ __forceinline__ uint4 file_rows_load(const uint8_t* __restrict__ blk, uint32_t hdr, int lane) {
  const int bd = hdr_bdoc(hdr);
  const int doc_sz = bd ? 16 * bd : hdr_vlen(hdr);
  const uint32_t voff = 1u + 16u * (uint32_t)(lane & 31) + __umul24((uint32_t)(lane >> 5), (uint32_t)doc_sz + 1u);
  return load16_unaligned(blk + voff);
}
// VInt (data_input.rs:78-111) out of two dwords holding its first 8 bytes
__device__ __forceinline__ uint32_t vint_from_words(uint32_t w0, uint32_t w1) {
  uint32_t v = w0 & 0x7fu;
  if (w0 & 0x80u) {
    v |= ((w0 >> 8) & 0x7fu) << 7;
    if (w0 & 0x8000u) {
      v |= ((w0 >> 16) & 0x7fu) << 14;
      if (w0 & 0x800000u) {
        v |= ((w0 >> 24) & 0x7fu) << 21;
        if (w0 & 0x80000000u) v |= (w1 & 0x0fu) << 28;
      }
    }
  }
  return v;
}
// file rows -> block-store rows: the VInt of an all-equal stream becomes its plain u32 value (for_util.rs:203-207)
__device__ __forceinline__ uint4 store_rows_from_file(uint4 rows, uint32_t hdr, int lane, bool has_freqs = true) {
  const int b = (lane >> 5) ? hdr_bfreq(hdr) : hdr_bdoc(hdr);
  if (b == 0) rows = make_uint4(vint_from_words(rows.x, rows.y), 0u, 0u, 0u);
  if (!has_freqs && (lane >> 5)) rows = make_uint4(1u, 0u, 0u, 0u);  // IndexOptions::Docs: "all freqs equal 1"
  return rows;
}
__device__ __host__ __forceinline__ int store_doc_rows(uint32_t hdr) { return hdr_bdoc(hdr) ? hdr_bdoc(hdr) : 1; }  // (a flagged non-PF block says 32)
__device__ __host__ __forceinline__ int store_freq_rows(uint32_t hdr) { return hdr_bfreq(hdr) ? hdr_bfreq(hdr) : 1; }

// ---- doc blocks that are not packed deltas (prepare time only) ---------------------------------------------------
// ForUtil::read_other_encode_block (for_util.rs:337-372): a doc block's header byte carries its encode type in bits 6-7
// — 0 PF (packed deltas), 1 EF (Elias-Fano over doc - base - 1; util/packed/elias_fano_encoder.rs), 2 BITSET (a bitmap
// from min_doc), 3 FULL (unimplemented in the reference itself). No Rucene build writes 1 or 2 (posting_writer.rs:46:
// use_ef = false, never set), the reader handles them, and so does k_prepare_blocks: such a block is decoded ONCE, its
// doc ids re-expressed as deltas and packed into the block store as ordinary BP128 rows — the query kernels never see
// anything but packed-delta blocks. In the directory header word such a block is flagged (bit 15, encode type in the
// vint-length field, 32 doc rows reserved) until k_prepare_blocks has rewritten it.
constexpr uint32_t HDR_NONPF = 0x8000u;
__device__ __host__ __forceinline__ bool hdr_nonpf(uint32_t h) { return (h & HDR_NONPF) != 0u; }
__device__ __forceinline__ int lz64(uint64_t v) { return v == 0 ? 64 : __builtin_clzll(v); }
struct EfShape {  // EliasFanoEncoder::new for 128 values (elias_fano_encoder.rs:47-146)
  int64_t upper_bound;
  int low_bits, upper_longs, lower_longs, index_longs, vlen;
};
__device__ __forceinline__ EfShape ef_shape(const uint8_t* p /* at the vlong */) {
  EfShape e;
  uint64_t ub = 0;
  int i = 0;
  for (; i < 9; ++i) { const uint64_t b = p[i]; ub |= (b & 0x7f) << (7 * i); if (!(b & 0x80)) { ++i; break; } }
  e.vlen = i;
  e.upper_bound = (int64_t)ub;
  const uint64_t fac = ub / 128u;
  e.low_bits = fac > 0 ? 63 - lz64(fac) : 0;
  const uint64_t max_high = ub >> e.low_bits;
  e.upper_longs = (int)((max_high + 128u + 63u) >> 6);
  e.lower_longs = (int)((128u * (uint64_t)e.low_bits + 63u) >> 6);
  const uint64_t n_index = max_high / 256u, max_entry = max_high + 127u;
  const int entry_bits = max_entry == 0 ? 0 : 64 - lz64(max_entry);
  e.index_longs = (int)((n_index * (uint64_t)entry_bits + 63u) >> 6);
  return e;
}
// bytes of a non-PF doc block after its header byte; < 0: malformed
__device__ __forceinline__ int nonpf_doc_bytes(const uint8_t* p /* at the header byte */, int etype) {
  if (etype == 2) {  // vint min_doc | u8 num_words | num_words x i64
    int v = 1;
    while (v < 5 && (p[v] & 0x80)) ++v;
    const int nw = p[1 + v];
    return nw <= 64 ? v + 1 + 8 * nw : -1;
  }
  const EfShape e = ef_shape(p + 1);
  if (e.upper_bound < 127 || e.upper_bound > 0x7fffffffLL || e.upper_longs > 6) return -1;
  return e.vlen + 8 * (e.upper_longs + e.lower_longs + e.index_longs);
}

// ---- the block store (query time) -------------------------------------------------------------------------------
// Phase 1 of a FullBlock decode: issue this lane's aligned 16-byte row load (lanes 0..31 -> doc rows, lanes 32..63
// -> freq rows) of the block whose rows start at `rows0`. Split from phase 2 so callers can issue later blocks'
// loads before decoding block i. The load is unconditional (rows past a stream read what follows — the store is
// padded) so that it never sits behind a branch: the waitcnt pass can then keep several blocks in flight.
// Address = uniform base + one 32-bit lane offset (SGPR-base form).
__device__ __forceinline__ uint4 block_rows_load(const uint8_t* __restrict__ rows0, uint32_t hdr, int lane) {
  const uint32_t voff = 16u * (uint32_t)(lane & 31) + __umul24(16u * (uint32_t)(lane >> 5), (uint32_t)store_doc_rows(hdr));  // one v_mad_u32_u24
  return *reinterpret_cast<const uint4*>(rows0 + voff);
}
__device__ __forceinline__ const uint8_t* block_rows_at(const uint8_t* __restrict__ term_rows, uint32_t row) {
  return term_rows + 16 * (size_t)row;
}

// Staging: every lane stores its row — rows past a stream's payload hold over-read bytes that extraction never
// uses (a value's straddle word lies in the next row only when that row is still payload), and an unconditional
// 1 KB wave store costs less issue time than a per-lane `row < b` test.
__device__ __forceinline__ void stage_rows(const uint4& rows, uint8_t* slab, int lane) {
  *reinterpret_cast<uint4*>(slab + (lane >> 5) * SLAB_STREAM + 16 * (lane & 31)) = rows;
}
template <bool LEGACY>
__device__ __forceinline__ void staged_doc_deltas(const uint8_t* slab, const uint4& rows, uint32_t hdr, int lane, uint32_t& d0, uint32_t& d1) {
  const int bd = hdr_bdoc(hdr);
  if (bd) extract_pair<LEGACY>(slab, bd, lane, d0, d1);
  else d0 = d1 = (uint32_t)readlane((int)rows.x, 0);
}
template <bool LEGACY>
__device__ __forceinline__ void staged_freqs(const uint8_t* slab, const uint4& rows, uint32_t hdr, int lane, uint32_t& f0, uint32_t& f1) {
  const int bf = hdr_bfreq(hdr);
  if (bf) extract_pair<LEGACY>(slab + SLAB_STREAM, bf, lane, f0, f1);
  else f0 = f1 = (uint32_t)readlane((int)rows.x, 32);
}

// Phase 2: stage the rows in the wave's LDS slab and extract postings 2*lane, 2*lane+1. `hdr` from the block
// directory. Wave-uniform control flow, no global loads.
template <bool LEGACY>
__device__ __forceinline__ BlockPair block_rows_decode(const uint4& rows, uint32_t hdr, uint8_t* slab, int lane) {
  stage_rows(rows, slab, lane);
  wave_sync();
  BlockPair out;
  staged_doc_deltas<LEGACY>(slab, rows, hdr, lane, out.d0, out.d1);
  staged_freqs<LEGACY>(slab, rows, hdr, lane, out.f0, out.f1);
  wave_sync();  // slab is free for the next block
  return out;
}

// one block of a term: `term_rows` = SegView::bstore + DevTerm::bs_base, `row` = SegView::dir_row of the block
template <bool LEGACY>
__device__ __forceinline__ BlockPair decode_block(const uint8_t* __restrict__ term_rows, uint32_t row, uint32_t hdr, uint8_t* slab, int lane) {
  return block_rows_decode<LEGACY>(block_rows_load(block_rows_at(term_rows, row), hdr, lane), hdr, slab, lane);
}

// A wave's view of up to 64 consecutive directory entries (one coalesced load), read back with readlane so
// the per-block loop carries no dependent directory loads.
struct DirChunk {
  uint32_t row, hdr;
  __device__ __forceinline__ void load(const uint32_t* __restrict__ dir_row, const uint16_t* __restrict__ dir_hdr, uint32_t base, int first,
                                       int count, int lane) {
    const bool ok = lane < count;
    row = ok ? dir_row[base + first + lane] : 0u;
    hdr = ok ? (uint32_t)dir_hdr[base + first + lane] : 0u;
  }
  __device__ __forceinline__ uint32_t row_at(int i) const { return (uint32_t)readlane((int)row, i); }
  __device__ __forceinline__ uint32_t hdr_at(int i) const { return (uint32_t)readlane((int)hdr, i); }
};

// deltas -> absolute doc ids (the `accum + delta` chain of posting_reader.rs:622-646, as a wave scan)
__device__ __forceinline__ void deltas_to_docs(uint32_t d0, uint32_t d1, int32_t base, int32_t& doc0, int32_t& doc1) {
  const int incl = wave_incl_scan((int)(d0 + d1));  // sum of the deltas up to and including this lane's pair
  doc1 = base + incl;
  doc0 = doc1 - (int)d1;
}

// Streams the FullBlocks [b0, b1) of one term through `body(block_index, doc0, doc1, freq0, freq1)`, keeping
// PREFETCH_DEPTH blocks' payload rows in flight per wavefront. One row load per block in flight is latency bound
// (Little: 7 waves/SIMD x ~150 B / ~0.7 us ~ 1.5 TB/s chip-wide — what a depth-1 pipeline measured); four deep
// covers the HBM/Infinity-Cache latency with the decode work of the blocks in between. `base` carries the
// running doc id (last doc of the previous block) in and out.
#ifndef RGPU_PREFETCH_DEPTH
#define RGPU_PREFETCH_DEPTH 4
#endif
constexpr int PREFETCH_DEPTH = RGPU_PREFETCH_DEPTH;  // default; the store-bound materialising decode runs shallower (see k_decode_terms)
// HAS_PN: also stream the term's posting-order norms (2 bytes per lane per block, SegView::pnorm) through the
// same ring; body(block_index, doc0, doc1, freq0, freq1, norm0, norm1) — norms are 0 without HAS_PN.
template <bool LEGACY, bool HAS_PN, int DEPTH = PREFETCH_DEPTH, typename Body>
__device__ __forceinline__ void stream_blocks(const uint8_t* __restrict__ term_rows, const uint32_t* __restrict__ dir_row,
                                              const uint16_t* __restrict__ dir_hdr, uint32_t dir_base,
                                              const uint8_t* __restrict__ pn, int b0, int b1, uint8_t* slab, int lane,
                                              int32_t& base, Body body) {
  for (int c0 = b0; c0 < b1; c0 += 64) {
    const int nb = min(64, b1 - c0);
    DirChunk dir;
    dir.load(dir_row, dir_hdr, dir_base, c0, nb, lane);
    auto step = [&](int idx, const uint4& rows, uint32_t nn) {
      const BlockPair bp = block_rows_decode<LEGACY>(rows, dir.hdr_at(idx), slab, lane);
      int32_t d0, d1;
      deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
      base = readlane(d1, 63);
      body(c0 + idx, d0, d1, bp.f0, bp.f1, nn & 0xffu, nn >> 8);
    };
    auto norms_of = [&](int idx) -> uint32_t {
      if (!HAS_PN) return 0u;
      // uniform term base + one 32-bit offset (a term's posting-order norms span < 2^31 bytes)
      return *reinterpret_cast<const uint16_t*>(pn + (128u * (uint32_t)(c0 + idx) + 2u * (uint32_t)lane));
    };
    // prefetch indices are clamped to the chunk's last block instead of being guarded: a redundant reload of
    // that block near the end is cheaper than a load behind a branch
    const int last = nb - 1;
    uint4 ring[DEPTH];
    uint32_t nring[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int pj = min(j, last);
      ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(pj)), dir.hdr_at(pj), lane);
      nring[j] = norms_of(pj);
    }
    int i = 0;
    for (; i + DEPTH <= nb; i += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {  // static ring slot j <-> block i + j
        const uint4 rows = ring[j];
        const uint32_t nn = nring[j];
        const int pj = min(i + j + DEPTH, last);
        ring[j] = block_rows_load(block_rows_at(term_rows, dir.row_at(pj)), dir.hdr_at(pj), lane);
        nring[j] = norms_of(pj);
        step(i + j, rows, nn);
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j)
      if (i + j < nb) step(i + j, ring[j], nring[j]);
  }
}

// ---- a term's tail as the query kernels see it -------------------------------------------------------------------
// k_prepare_blocks decodes every term's VInt tail ONCE (decode_tail below), validates it and leaves it in the block
// store behind the term's FullBlock rows as 64 cells of 16 bytes — cell `lane` = {doc 2*lane, doc 2*lane+1, freq 2*lane,
// freq 2*lane+1}, absolute doc ids, INT_MAX / 0 past the tail's end (TAIL_STORE_ROWS rows; the tail starts at row
// dir_row[nblocks] of the term). Query kernels read their two postings with one 16-byte load: no byte-serial VInt
// parsing and none of its ~15 registers in any scoring kernel.
constexpr int TAIL_STORE_ROWS = 64;  // 64 lanes x 16 bytes
__device__ __forceinline__ void tail_load(const uint8_t* __restrict__ term_rows, uint32_t tail_row, int lane, int32_t& doc0, int32_t& doc1,
                                          uint32_t& f0, uint32_t& f1) {
  const uint4 c = *reinterpret_cast<const uint4*>(term_rows + 16 * (size_t)tail_row + 16 * lane);
  doc0 = (int32_t)c.x; doc1 = (int32_t)c.y;
  f0 = c.z; f1 = c.w;
}

// ---- VInt tail (< 128 postings) -------------------------------------------------------------------------------
#ifndef RGPU_TAIL_INLINE
#define RGPU_TAIL_INLINE __forceinline__
#endif
// posting i: code = vint; delta = code >>> 1; freq = (code & 1) ? 1 : vint        (posting_reader.rs:316-325)
// Parallel formulation: (1) every lane scans 20 bytes for VInt terminators, a wave scan turns terminator counts
// into value indices and each terminator lane assembles its value by looking back over continuation bytes;
// (2) "is this value a code or a freq" is the recurrence s[k+1] = !(s[k] && even(v[k])), a scan over the
// 4-element function monoid {const0, const1, id, not}; (3) a wave scan over the code deltas gives doc ids.
// Returns postings 2*lane, 2*lane+1 (valid while index < n).
__device__ __forceinline__ uint32_t compose_fn(uint32_t first, uint32_t then) {
  // a function {0,1}->{0,1} is 2 bits: bit0 = f(0), bit1 = f(1); result = then o first
  const uint32_t r0 = (then >> (first & 1)) & 1;
  const uint32_t r1 = (then >> ((first >> 1) & 1)) & 1;
  return r0 | (r1 << 1);
}

// Stages (0) + (1) of the tail decoder: TAIL_MAX_BYTES of the stream are parsed as VInts in parallel and the first 256
// values are left, in order, in `slab` + 1296 (u32 each; unwritten slots read 1).
__device__ __forceinline__ uint32_t* stage_vints(const uint8_t* __restrict__ tail, uint8_t* slab, int lane) {
  uint8_t* bytes = slab;                                              // [0, 1296)
  uint32_t* vals = reinterpret_cast<uint32_t*>(slab + 1296);         // 256 values (+ 8 pad words)
  // (0) stage TAIL_MAX_BYTES: 80 x 16 B
  for (int c = lane; c < TAIL_MAX_BYTES / 16; c += WAVE)
    *reinterpret_cast<uint4*>(bytes + 16 * c) = load16_unaligned(tail + 16 * c);
  for (int i = lane; i < 264; i += WAVE) vals[i] = 1;  // unwritten values read as odd codes (harmless)
  wave_sync();
  // (1) terminators in my 20-byte span
  const int span = 20 * lane;
  uint32_t w[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) w[j] = *reinterpret_cast<const uint32_t*>(bytes + span + 4 * j);
  uint32_t term = 0;
#pragma unroll
  for (int j = 0; j < 20; ++j) term |= (((w[j >> 2] >> (8 * (j & 3) + 7)) & 1u) ^ 1u) << j;
  const int cnt = __popc(term);
  int vi = wave_incl_scan(cnt) - cnt;
  while (term) {
    const int j = __builtin_ctz(term);
    term &= term - 1;
    int p = span + j;
    uint32_t v = bytes[p];
    for (int back = 0; back < 4 && p > 0 && (bytes[p - 1] & 0x80); ++back) {
      --p;
      v = (v << 7) | (bytes[p] & 0x7f);
    }
    if (vi < 256) vals[vi] = v;
    ++vi;
  }
  wave_sync();
  return vals;
}
// A block of up to 128 plain VInts (the trailing position block of a term, posting_reader.rs:1285-1324 without payloads
// or offsets): values 2*lane, 2*lane+1, as stored.
__device__ __forceinline__ void decode_vint_block(const uint8_t* __restrict__ src, uint8_t* slab, int lane, uint32_t& v0, uint32_t& v1) {
  const uint32_t* vals = stage_vints(src, slab, lane);
  v0 = vals[2 * lane];
  v1 = vals[2 * lane + 1];
  wave_sync();
}

// The same block of a field that stores payloads (flags bit 0) and / or offsets (bit 1) — posting_reader.rs:1285-1312: per
// position  code = vint [payload length = vint when code & 1] [that many payload bytes] [offset code = vint [offset length =
// vint when code & 1]], position delta = code >>> 1 with payloads, code without. Payload bytes are raw, so where a value
// starts depends on every length in front of it: the walk is serial. Every lane walks (uniform control flow, LDS broadcast
// reads) through a 512-byte window of the stream that the lanes refill together whenever the walk is about to leave it — a
// long payload is jumped over, never read — and keeps the two deltas that are its own. `count` = total_term_freq % 128.
constexpr int POS_TAIL_PAYLOADS = 1, POS_TAIL_OFFSETS = 2;
// `limit`: bytes of the file from src on (the device copy is zero-padded for 8 KB behind them). A length that leads out of
// the file — a corrupt payload length is up to 4 GiB — ends the walk: false.
__device__ __forceinline__ bool decode_vint_block_everything(const uint8_t* __restrict__ src, int64_t limit, int count, int flags, uint8_t* slab,
                                                             int lane, uint32_t& v0, uint32_t& v1) {
  constexpr int WIN = 512;
  int64_t wb = 0, at = 0;  // window base and walk position, bytes from src (wave-uniform)
  bool ok = true;
  auto refill = [&]() {
    wave_sync();
    ok = ok && at >= 0 && at <= limit;
    wb = ok ? at : 0;  // (a walk that left the file reads its first window again: harmless, and reported)
    at = wb;
    if (lane < WIN / 16) *reinterpret_cast<uint4*>(slab + 16 * lane) = load16_unaligned(src + wb + 16 * lane);
    wave_sync();
  };
  // the four bytes at the walk position (two aligned LDS dwords shifted into place)
  auto peek = [&]() -> uint32_t {
    const uint32_t o = (uint32_t)(at - wb);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(slab + (o & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u);
  };
  auto vint = [&]() -> uint32_t {  // data_input.rs:78-111; at most 5 bytes, the walk stays 12 bytes inside the window
    uint32_t b = peek(), v = b & 0x7fu;
    int n = 1;
    if (b & 0x80u) { v |= ((b >> 8) & 0x7fu) << 7; n = 2;
      if (b & 0x8000u) { v |= ((b >> 16) & 0x7fu) << 14; n = 3;
        if (b & 0x800000u) { v |= ((b >> 24) & 0x7fu) << 21; n = 4;
          if (b & 0x80000000u) { at += 4; v |= (peek() & 0x0fu) << 28; at -= 4; n = 5; } } } }
    at += n;
    return v;
  };
  refill();
  uint32_t payload_length = 0;
  v0 = v1 = 0u;
  for (int i = 0; i < count; ++i) {
    if (at - wb > WIN - 24) refill();  // a code and a length: <= 10 bytes (+ the peek's over-read)
    uint32_t code = vint();
    if (flags & POS_TAIL_PAYLOADS) {
      if (code & 1u) payload_length = vint();
      code >>= 1;
      at += payload_length;  // posting_reader.rs:1299-1302: seek past the bytes
    }
    if (flags & POS_TAIL_OFFSETS) {
      if (at - wb > WIN - 24 || at < wb) refill();
      if (vint() & 1u) (void)vint();  // the offset length changed
    }
    v0 = i == 2 * lane ? code : v0;
    v1 = i == 2 * lane + 1 ? code : v1;
    if (!ok) break;
  }
  wave_sync();
  return ok && at <= limit;
}

// has_freqs == false (IndexOptions::Docs, posting_reader.rs:326-331): every value is a doc delta, every freq is 1.
__device__ RGPU_TAIL_INLINE void decode_tail(const uint8_t* __restrict__ tail, int n, int32_t base, uint8_t* slab, int lane,
                                            int32_t& doc0, int32_t& doc1, uint32_t& f0, uint32_t& f1, bool has_freqs = true) {
  uint32_t* vals = stage_vints(tail, slab, lane);
  if (!has_freqs) {  // wave-uniform: value i is posting i's delta
    const uint32_t d0 = (2 * lane < n) ? vals[2 * lane] : 0u;
    const uint32_t d1 = (2 * lane + 1 < n) ? vals[2 * lane + 1] : 0u;
    f0 = f1 = 1u;
    deltas_to_docs(d0, d1, base, doc0, doc1);
    wave_sync();
    return;
  }
  // (2) classify 4 values per lane: code / freq
  uint32_t v4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v4[j] = vals[4 * lane + j];
  uint32_t F = 2;  // identity
#pragma unroll
  for (int j = 0; j < 4; ++j) F = compose_fn(F, (v4[j] & 1) ? 3u /*const 1*/ : 1u /*not*/);
  // inclusive scan of F under composition (Hillis-Steele over lanes)
  uint32_t S = F;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t prev = (uint32_t)__shfl_up((int)S, d);
    if (lane >= d) S = compose_fn(prev, S);
  }
  uint32_t before = (uint32_t)__shfl_up((int)S, 1);  // composition of all earlier lanes
  uint32_t state = lane == 0 ? 1u : ((before >> 1) & 1u);  // apply to s[0] = 1 (first value is a code)
  uint32_t is_code[4];
  int ncodes = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    is_code[j] = state;
    ncodes += (int)state;
    state = ((state & 1u) && !(v4[j] & 1u)) ? 0u : 1u;
  }
  int pi = wave_incl_scan(ncodes) - ncodes;
  wave_sync();
  uint32_t* pd = reinterpret_cast<uint32_t*>(slab);        // posting deltas [128]  (byte area is dead now)
  uint32_t* pf = reinterpret_cast<uint32_t*>(slab + 512);  // posting freqs  [128]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (is_code[j]) {
      if (pi < 128) {
        pd[pi] = v4[j] >> 1;
        pf[pi] = (v4[j] & 1) ? 1u : vals[4 * lane + j + 1];
      }
      ++pi;
    }
  }
  wave_sync();
  // (3) two postings per lane + prefix sum
  const uint32_t d0 = (2 * lane < n) ? pd[2 * lane] : 0u;
  const uint32_t d1 = (2 * lane + 1 < n) ? pd[2 * lane + 1] : 0u;
  f0 = pf[2 * lane];
  f1 = pf[2 * lane + 1];
  deltas_to_docs(d0, d1, base, doc0, doc1);
  wave_sync();
}

}  // namespace rgpu
