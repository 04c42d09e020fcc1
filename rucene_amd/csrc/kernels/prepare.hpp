// Term preparation on the GPU, two launches: k_prepare_terms — one workgroup per term turns the term's level-0 skip
// entries into a flat block directory {last doc id, byte offset, header word, store row} in HBM; k_prepare_blocks —
// one wavefront per chunk of blocks copies the FullBlock payloads into the 16-byte aligned block store and lays the
// docs' norms out in posting order. GPU counterpart of (paths relative to
// /root/reference/src/core):
//   codec/postings/skip_reader.rs:460-511   load_skip_levels  (vlong length + bytes for levels L-1..1, then level 0)
//   codec/postings/skip_reader.rs:431-453   read_skip_data    (vint docDelta, vlong docFpDelta per entry)
//   codec/postings/skip_reader.rs:513-539   load_next_skip    (the running sums skip_doc / doc_pointer)
//   codec/postings/for_util.rs:196-223      block header byte (encode type, num_bits, all-equal vint)
// The higher skip levels are subsampled copies of level 0 with child pointers; a flat level-0 directory plus
// binary search gives the same `advance` answers, so only their byte lengths are parsed (to find level 0).
//
// VInt streams are decoded data-parallel: each thread inspects 4 bytes, a workgroup scan over terminator
// counts yields each value's index, and the thread owning a terminator assembles the value by looking back
// over its continuation bytes.
#pragma once
#include "decode.hpp"
#include "types.hpp"

namespace rgpu {

constexpr int PREP_THREADS = 256;

// err[0] = the most severe status (rgpu_status, or -101 "plan again with worst-case rows"), err[1] = the highest-numbered
// check that failed — which of this file's consistency checks it was ends up in the host's error message
__device__ __forceinline__ void flag_err(int* err, int status, int site) {
  atomicMin(err, status);
  atomicMax(err + 1, site);
}

// workgroup exclusive scan for PREP_THREADS threads; `total` is uniform on return
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_sums, uint32_t& total) {
  const int lane = lane_id();
  const int wave = wave_id();
  const uint32_t incl = (uint32_t)wave_incl_scan((int)v);
  __syncthreads();  // protect wave_sums reuse
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PREP_THREADS / 64; ++w) {
    const uint32_t s = wave_sums[w];
    if (w < wave) off += s;
    tot += s;
  }
  total = tot;
  return off + incl - v;
}

__device__ __forceinline__ uint64_t read_vlong_serial(const uint8_t* p, int* len) {
  uint64_t v = 0;
  int i = 0;
  for (; i < 9; ++i) {
    uint64_t b = p[i];
    v |= (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) { ++i; break; }
  }
  *len = i;
  return v;
}

__device__ __forceinline__ int vint_len_serial(const uint8_t* p) {
  int i = 0;
  while (i < 4 && (p[i] & 0x80)) ++i;
  return i + 1;
}

template <bool LEGACY>
__global__ __launch_bounds__(PREP_THREADS) void k_prepare_terms(const uint8_t* __restrict__ doc, int64_t doc_len,
                                                                 const PrepTerm* __restrict__ terms, int32_t* dir_last,
                                                                 uint32_t* dir_off, uint32_t* dir_row, uint16_t* dir_hdr,
                                                                 uint64_t* dir_pos, int has_freqs, int* err) {
  const PrepTerm t = terms[blockIdx.x];
  const int tid = (int)threadIdx.x;
  __shared__ uint32_t s_ws[PREP_THREADS / 64];
  __shared__ int64_t s_l0;
  __shared__ int s_nonpf;  // some block of this term is EF / BITSET encoded
  if (tid == 0) s_nonpf = 0;
  __syncthreads();

  if (t.n_entries > 0) {
    // ---- where does level 0 start? (skip_reader.rs:481-509)
    if (tid == 0) {
      int64_t p = t.skip_fp;
      for (int lvl = t.n_levels - 1; lvl >= 1; --lvl) {
        int n;
        uint64_t len = read_vlong_serial(doc + p, &n);
        p += n + (int64_t)len;
        if (p >= doc_len) { flag_err(err, -4, 1); p = t.skip_fp; break; }
      }
      s_l0 = p;
    }
    __syncthreads();
    const uint8_t* l0 = doc + s_l0;
    const int64_t l0_room = doc_len - s_l0;  // bytes of the file from level 0 on (the device copy is padded by 8 KiB of zeros)
    // ---- parallel VInt decode of the level-0 entries: docDelta (vint), docFpDelta (vlong) and, for a positions field
    // (dir_pos != null; skip_writer.rs:261-289), posFpDelta (vlong), posBufferUpto (vint)
    const uint32_t vals = dir_pos ? 4u : 2u;
    const uint32_t need = vals * (uint32_t)t.n_entries;
    uint32_t done = 0;
    int64_t chunk = 0;
    while (done < need) {
      const int64_t my = chunk + 16 * tid;  // 16 bytes per thread, 4 KiB per round (a 2 M-posting term has ~80 KB of level 0)
      if (chunk >= l0_room) { if (tid == 0) flag_err(err, -4, 2); break; }  // ran off the file looking for skip entries (uniform)
      const uint4 w4 = load16_unaligned(l0 + my);
      const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
      uint32_t term = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) term |= (((ws[j >> 2] >> (8 * (j & 3) + 7)) & 1u) ^ 1u) << j;
      uint32_t total;
      uint32_t vi = done + block_excl_scan((uint32_t)__popc(term), s_ws, total);
      while (term) {
        const int j = __builtin_ctz(term);
        term &= term - 1;
        int64_t p = my + j;
        uint64_t v = l0[p];
        for (int back = 0; back < 9 && p > 0 && (l0[p - 1] & 0x80); ++back) {
          --p;
          v = (v << 7) | (uint64_t)(l0[p] & 0x7f);
        }
        if (vi < need) {
          const uint32_t e = vi / vals, f = vi % vals;
          if (f == 0) dir_last[t.dir_base + e] = (int32_t)v;
          else if (f == 1) dir_off[t.dir_base + e + 1] = (uint32_t)v;
          else reinterpret_cast<uint32_t*>(dir_pos + t.dir_base + e + 1)[f - 2] = (uint32_t)v;  // [0] posFpDelta, [1] upto
        }
        ++vi;
      }
      if (total == 0) { if (tid == 0) flag_err(err, -4, 3); break; }  // 4 KiB without a terminator: corrupt
      done += total;
      chunk += 16 * PREP_THREADS;
    }
    __syncthreads();
    // ---- deltas -> running sums (skip_doc[0] += delta ; doc_pointer[0] += delta, skip_reader.rs:530, 434)
    uint32_t carry_doc = 0, carry_off = 0, carry_pos = 0;
    for (int e0 = 0; e0 < t.n_entries; e0 += PREP_THREADS) {
      const int e = e0 + tid;
      const bool ok = e < t.n_entries;
      const uint32_t dd = ok ? (uint32_t)dir_last[t.dir_base + e] : 0u;
      const uint32_t fo = ok ? dir_off[t.dir_base + e + 1] : 0u;
      uint32_t tot_d, tot_o;
      const uint32_t sd = block_excl_scan(dd, s_ws, tot_d) + dd + carry_doc;
      const uint32_t so = block_excl_scan(fo, s_ws, tot_o) + fo + carry_off;
      if (ok) {
        dir_last[t.dir_base + e] = (int32_t)sd;
        dir_off[t.dir_base + e + 1] = so;
      }
      carry_doc += tot_d;
      carry_off += tot_o;
      if (dir_pos) {  // uniform: the position pointer is a running sum too, the buffered count is absolute
        uint32_t* pp = reinterpret_cast<uint32_t*>(dir_pos + t.dir_base + e + 1);
        const uint32_t po = ok ? pp[0] : 0u;
        uint32_t tot_p;
        const uint32_t sp = block_excl_scan(po, s_ws, tot_p) + po + carry_pos;
        if (ok) { pp[0] = sp; if (pp[1] >= 128u) flag_err(err, -4, 4); }
        carry_pos += tot_p;
      }
    }
  }
  if (tid == 0) {
    if (dir_pos) dir_pos[t.dir_base] = 0ull;
    dir_off[t.dir_base] = 0;
    if (t.nblocks > t.n_entries) dir_last[t.dir_base + t.nblocks - 1] = DIR_SENTINEL_DOC;  // df % 128 == 0
  }
  __syncthreads();
  // ---- block headers (for_util.rs:196-223) + consistency of every skip pointer with the block sizes
  for (int i = tid; i < t.nblocks; i += PREP_THREADS) {
    const uint32_t off = dir_off[t.dir_base + i];
    // offsets are running sums of deltas nobody has checked yet: a block (<= 2 + 2 * 512 bytes) must start inside the file
    if ((uint64_t)t.start_fp + (uint64_t)off + 1030u > (uint64_t)doc_len + 4096u) { flag_err(err, -4, 5); dir_hdr[t.dir_base + i] = 0; continue; }
    const uint8_t* p = doc + t.start_fp + off;
    const uint32_t h = p[0];
    int bd = (int)(h & 63);
    int vlen = 0;
    const int etype = (int)(h >> 6);
    int doc_sz = 16 * bd;
    uint32_t flag = 0;
    if (etype != 0) {  // EF / BITSET doc block (decode.hpp): sized here, decoded and re-packed by k_prepare_blocks
      if (etype == 3 || LEGACY) flag_err(err, -5, 6);  // FULL is unimplemented in the reference; EF + the legacy layout: not served
      doc_sz = etype == 3 ? 0 : nonpf_doc_bytes(p, etype);
      if (doc_sz < 0) { flag_err(err, -4, 7); doc_sz = 0; }
      bd = 32;       // doc rows reserved in the block store
      vlen = etype;  // the vint-length field is free in a flagged word
      flag = HDR_NONPF;
      s_nonpf = 1;
    } else {
      if (bd > 32) flag_err(err, -4, 8);
      if (bd == 0) { vlen = vint_len_serial(p + 1); doc_sz = vlen; }
    }
    // the freq block (absent for IndexOptions::Docs: posting_writer.rs:334-351 writes it only when the field has freqs;
    // the directory then says "all-equal freq stream" and the block store supplies the value 1)
    int bf = 0;
    uint32_t end = off + 1u + (uint32_t)doc_sz;
    if (has_freqs) {
      const uint32_t h2 = p[1 + doc_sz];
      bf = (int)(h2 & 63);
      if (bf > 32) flag_err(err, -4, 9);
      const int freq_sz = bf ? 16 * bf : vint_len_serial(p + 1 + doc_sz + 1);
      end += 1u + (uint32_t)freq_sz;
    }
    if (i < t.n_entries && end != dir_off[t.dir_base + i + 1]) flag_err(err, -4, 10);
    if (i > 0 && i < t.n_entries && dir_last[t.dir_base + i] <= dir_last[t.dir_base + i - 1]) flag_err(err, -4, 11);
    dir_hdr[t.dir_base + i] = (uint16_t)((uint32_t)bd | ((uint32_t)vlen << 6) | ((uint32_t)bf << 9) | flag);
  }
  __syncthreads();
  // ---- block store rows: block i takes max(b_doc,1) + max(b_freq,1) rows; exclusive prefix sum over the blocks
  uint32_t carry_rows = 0;
  for (int i0 = 0; i0 < t.nblocks; i0 += PREP_THREADS) {
    const int i = i0 + tid;
    const bool ok = i < t.nblocks;
    const uint32_t h = ok ? (uint32_t)dir_hdr[t.dir_base + i] : 0u;
    const uint32_t r = ok ? (uint32_t)(store_doc_rows(h) + store_freq_rows(h)) : 0u;
    uint32_t tot;
    const uint32_t at = block_excl_scan(r, s_ws, tot) + carry_rows;
    if (ok) dir_row[t.dir_base + i] = at;
    carry_rows += tot;
  }
  if (tid == 0) dir_row[t.dir_base + t.nblocks] = carry_rows;  // where the term's decoded tail goes (k_prepare_blocks)
  const uint32_t tail_rows = (t.df > 1 && t.df % 128 != 0) ? (uint32_t)TAIL_STORE_ROWS : 0u;
  if (carry_rows + tail_rows > t.bs_rows) {
    // more rows than the framing the host sized the store from: corrupt — unless the term has EF / BITSET blocks, whose
    // re-packed deltas may outgrow their file bytes: -101 asks the host to plan this call again with worst-case sizes
    if (tid == 0) flag_err(err, s_nonpf ? -101 : -4, 12);
    return;
  }
}

// Second half of term preparation, spread over the whole GPU (a 2 M-posting term has 15 k blocks: far too many for the
// four wavefronts of its directory workgroup). Items = (term, chunk of PREP_BLOCKS_PER_ITEM blocks), one wavefront
// each: copy the payload rows into the block store, then — with norms — decode the block from those rows and gather
// its docs' norm bytes in posting order (SegView::pnorm).
constexpr int PREP_BLOCKS_PER_ITEM = 32;

template <bool LEGACY>
__global__ __launch_bounds__(PREP_THREADS) void k_prepare_blocks(const uint8_t* __restrict__ doc, const PrepTerm* __restrict__ terms,
                                                                  const int64_t* __restrict__ item_prefix, int n_terms,
                                                                  int64_t n_items, int32_t* dir_last,
                                                                  const uint32_t* __restrict__ dir_off,
                                                                  const uint32_t* __restrict__ dir_row,
                                                                  uint16_t* dir_hdr, uint8_t* bstore,
                                                                  const uint8_t* __restrict__ norms, uint8_t* pnorm,
                                                                  uint64_t* __restrict__ dir_bmax, int ranked, int has_freqs,
                                                                  int32_t max_doc, int* err) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[PREP_THREADS / 64][SLAB_BYTES];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * (PREP_THREADS / 64) + wave;
  if (item >= n_items || *err != 0) return;  // a term whose framing did not check out must not be walked
  int ti = 0;
  {
    int lo = 0, hi = n_terms;  // largest t with item_prefix[t] <= item
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (item_prefix[mid] <= item) lo = mid; else hi = mid;
    }
    ti = lo;
  }
  const PrepTerm t = terms[ti];
  const int b0 = (int)(item - item_prefix[ti]) * PREP_BLOCKS_PER_ITEM;
  const int b1 = min(t.nblocks, b0 + PREP_BLOCKS_PER_ITEM);
  uint8_t* term_rows = bstore + t.bs_base;
  // the term's last item also takes its VInt tail (posting_reader.rs:308-333): decoded here once, checked like the blocks
  // (doc ids strictly increasing from the last FullBlock's last doc, inside the segment) and stored as 16-byte cells (tail_load)
  const int tail_n = t.df > 1 ? t.df % 128 : 0;
  if (tail_n > 0 && b0 + PREP_BLOCKS_PER_ITEM >= t.nblocks) {
    const uint32_t toff = t.nblocks ? dir_off[t.dir_base + t.nblocks] : 0u;
    const int32_t tbase = t.nblocks ? dir_last[t.dir_base + t.nblocks - 1] : 0;
    int32_t d0, d1;
    uint32_t f0, f1;
    decode_tail(doc + t.start_fp + toff, tail_n, tbase, slabs[wave], lane, d0, d1, f0, f1, has_freqs != 0);
    const bool v0 = 2 * lane < tail_n, v1 = 2 * lane + 1 < tail_n;
    const int32_t prev = __builtin_amdgcn_update_dpp(tbase, d1, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 <- the base doc
    const bool first_ok = (t.nblocks == 0 && lane == 0) ? d0 >= 0 : d0 > prev;  // a term's very first doc may be doc 0
    const bool bad = (v0 && (!first_ok || d0 >= max_doc)) || (v1 && (d1 <= d0 || d1 >= max_doc));
    if (__ballot(bad)) { if (lane == 0) flag_err(err, -4, 13); return; }
    uint8_t* tp = term_rows + 16 * (size_t)dir_row[t.dir_base + t.nblocks];
    *reinterpret_cast<uint4*>(tp + 16 * lane) = make_uint4(v0 ? (uint32_t)d0 : 0x7fffffffu, v1 ? (uint32_t)d1 : 0x7fffffffu, v0 ? f0 : 0u, v1 ? f1 : 0u);
    // the tail's directory slot gets its last doc, like a FullBlock's: the wide OR kernel walks tails as one more block
    const int32_t tail_last = readlane(((tail_n - 1) & 1) ? d1 : d0, (tail_n - 1) >> 1);
    if (lane == 0) dir_last[t.dir_base + t.nblocks] = tail_last;
    if (norms != nullptr) {  // ... and its posting-order norms continue the FullBlocks'
      const uint32_t n0 = v0 ? norms[d0] : 0u, n1 = v1 ? norms[d1] : 0u;
      *reinterpret_cast<uint16_t*>(pnorm + t.pn_base + 128 * (uint64_t)t.nblocks + 2 * lane) = (uint16_t)(n0 | (n1 << 8));
    }
  }
  for (int blk = b0; blk < b1; ++blk) {
    uint32_t hdr = dir_hdr[t.dir_base + blk];
    const uint32_t row0 = dir_row[t.dir_base + blk];
    uint4 rows;
    if (!hdr_nonpf(hdr)) {
      rows = store_rows_from_file(file_rows_load(doc + t.start_fp + dir_off[t.dir_base + blk], hdr, lane), hdr, lane, has_freqs != 0);
    } else {
      // ---- an EF / BITSET doc block: 128 doc ids -> deltas -> BP128 rows (decode.hpp; for_util.rs:337-372,
      // posting_reader.rs:622-637, elias_fano_decoder.rs:95-169, util/bit_set.rs:351-376)
      const uint8_t* p = doc + t.start_fp + dir_off[t.dir_base + blk];
      const int etype = hdr_vlen(hdr);
      int32_t* ids = reinterpret_cast<int32_t*>(slabs[wave]);            // 128 doc ids
      uint32_t* pack = reinterpret_cast<uint32_t*>(slabs[wave] + 512);   // 132 dwords of BP128 rows
      const int32_t pf_base = blk == 0 ? 0 : dir_last[t.dir_base + blk - 1];
      int doc_sz, total;
      if (etype == 2) {
        int v = 1;
        while (v < 5 && (p[v] & 0x80)) ++v;
        int vl;
        const int32_t min_doc = (int32_t)read_vint_uniform(p + 1, &vl);
        const int nw = p[1 + v];
        doc_sz = v + 1 + 8 * nw;
        uint64_t word = 0;
        if (lane < nw) { const uint8_t* wp = p + 2 + v + 8 * lane; word = (uint64_t)load4_unaligned(wp) | ((uint64_t)load4_unaligned(wp + 4) << 32); }
        const int cnt = __popcll(word);
        const int incl = wave_incl_scan(cnt);
        total = readlane(incl, 63);
        int at = incl - cnt;
        while (word) {
          if (at < 128) ids[at] = min_doc + 64 * lane + (int)__builtin_ctzll(word);
          ++at;
          word &= word - 1;
        }
      } else {
        const EfShape e = ef_shape(p + 1);
        doc_sz = e.vlen + 8 * (e.upper_longs + e.lower_longs + e.index_longs);
        const uint8_t* up = p + 1 + e.vlen;
        const uint8_t* lp = up + 8 * e.upper_longs;
        const int32_t ef_base = blk == 0 ? -1 : pf_base;  // refill_docs: ef_base_doc = accum when accum > 0, else -1
        uint64_t word = 0;
        if (lane < e.upper_longs) word = (uint64_t)load4_unaligned(up + 8 * lane) | ((uint64_t)load4_unaligned(up + 8 * lane + 4) << 32);
        const int cnt = __popcll(word);
        const int incl = wave_incl_scan(cnt);
        total = readlane(incl, 63);
        int i = incl - cnt;  // index of this lane's first value
        const uint64_t lmask = e.low_bits ? (~0ull >> (64 - e.low_bits)) : 0ull;
        while (word) {
          const int64_t high = 64 * lane + (int)__builtin_ctzll(word) - i;  // set bit position - index (current_high_value)
          uint64_t low = 0;
          if (e.low_bits) {  // unpack_value
            const int bit_pos = i * e.low_bits, at = bit_pos & 63;
            const uint8_t* wp = lp + 8 * (bit_pos >> 6);
            low = ((uint64_t)load4_unaligned(wp) | ((uint64_t)load4_unaligned(wp + 4) << 32)) >> at;
            if (at + e.low_bits > 64) low |= ((uint64_t)load4_unaligned(wp + 8) | ((uint64_t)load4_unaligned(wp + 12) << 32)) << (64 - at);
            low &= lmask;
          }
          if (i < 128) ids[i] = (int32_t)(((uint64_t)high << e.low_bits) | low) + 1 + ef_base;
          ++i;
          word &= word - 1;
        }
      }
      pack[lane] = 0u; pack[64 + lane] = 0u;
      if (lane < 4) pack[128 + lane] = 0u;
      wave_sync();
      if (total != 128) { if (lane == 0) flag_err(err, -4, 14); return; }
      const int32_t a = ids[2 * lane], b = ids[2 * lane + 1];
      const int32_t before = lane == 0 ? pf_base : ids[2 * lane - 1];
      const uint32_t d0 = (uint32_t)(a - before), d1 = (uint32_t)(b - a);
      const uint32_t top = wave_reduce_max_u32(d0 | d1);  // bits_required looks at the OR: the same most significant bit as the maximum
      const int bits = top == 0u ? 1 : 32 - __builtin_clz(top);
      auto put = [&](int i, uint32_t v) {  // SIMD128Packer::pack for one value (packed_simd.rs:81-108)
        const int bit = (i >> 2) * bits, w = bit >> 5, sft = bit & 31, l = i & 3;
        atomicOr(&pack[4 * w + l], v << sft);
        if (sft + bits > 32) atomicOr(&pack[4 * (w + 1) + l], v >> (32 - sft));
      };
      put(2 * lane, d0);
      put(2 * lane + 1, d1);
      wave_sync();
      hdr = (uint32_t)bits | (hdr & (63u << 9));  // from here on an ordinary packed-delta block
      if (lane == 0) dir_hdr[t.dir_base + blk] = (uint16_t)hdr;
      if (lane < 32) rows = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(pack) + 16 * lane);
      else rows = load16_unaligned(p + 1 + doc_sz + 1 + 16 * (lane & 31));
      rows = store_rows_from_file(rows, hdr, lane, has_freqs != 0);
      wave_sync();
    }
    const int half = lane >> 5, row = lane & 31;
    const int rd = store_doc_rows(hdr);
    if (row < (half ? store_freq_rows(hdr) : rd))
      *reinterpret_cast<uint4*>(term_rows + 16 * (size_t)(row0 + (uint32_t)(half ? rd : 0) + (uint32_t)row)) = rows;
    // Every FullBlock is decoded and validated here, once: doc ids inside the segment and strictly increasing (a zero
    // delta or a 32-bit wrap shows up as d1 <= d0 or d0 <= the previous lane's d1), and the block ending on the doc its
    // skip entry names. The query kernels then gather norms / live bits and index windows with these docs unchecked.
    const int32_t base = blk == 0 ? 0 : dir_last[t.dir_base + blk - 1];
    const BlockPair bp = block_rows_decode<LEGACY>(rows, hdr, slabs[wave], lane);
    int32_t d0, d1;
    deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
    // df % 128 == 0: no skip entry names the final block's last doc (k_prepare_terms left a sentinel there); the
    // decode does — windowed consumers (the OR kernel) can then tell that the term has ended
    if (blk == t.nblocks - 1 && t.nblocks > t.n_entries && lane == 63) dir_last[t.dir_base + blk] = d1;
    const int32_t prev = __builtin_amdgcn_update_dpp(base, d1, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 <- the block's base doc
    // (a term's very first delta is relative to doc 0 and may be 0: posting_writer.rs:298)
    const bool bad_first = (blk == 0 && lane == 0) ? d0 < 0 : d0 <= prev;
    const bool bad = bad_first || d1 <= d0 || d1 >= max_doc;
    const bool bad_last = blk < t.n_entries && lane == 63 && d1 != dir_last[t.dir_base + blk];
    if (__ballot(bad || bad_last)) { if (lane == 0) flag_err(err, -4, 15); return; }
    if (norms != nullptr) {
      const uint32_t n0 = norms[d0], n1 = norms[d1];
      *reinterpret_cast<uint16_t*>(pnorm + t.pn_base + 128 * (uint64_t)blk + 2 * lane) = (uint16_t)(n0 | (n1 << 8));
      // the block's (freq, norm rank) frontier word (SegView::dir_bmax)
      uint64_t w = 15ull;
      const uint32_t fmax = wave_reduce_max_u32(bp.f0 > bp.f1 ? bp.f0 : bp.f1);
      if (ranked && fmax <= 10u) {
        w = fmax;
#pragma unroll
        for (uint32_t f = 1; f <= 10; ++f) {
          const uint32_t r0 = bp.f0 == f ? n0 : 0u, r1 = bp.f1 == f ? n1 : 0u;
          w |= (uint64_t)(wave_reduce_max_u32(r0 > r1 ? r0 : r1) & 63u) << (4 + 6 * (f - 1));
        }
      }
      if (lane == 0) dir_bmax[t.dir_base + blk] = w;
    } else if (lane == 0) {
      dir_bmax[t.dir_base + blk] = 15ull;
    }
  }
}

}  // namespace rgpu
