// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's 128-value block bit-packers.
//
// Follows (paths relative to /root/reference/src/core):
//   util/packed/packed_simd.rs:81-108   pack_bits!   (SIMD-BP128 "vertical" layout, 4 interleaved u32 lanes)
//   util/packed/packed_simd.rs:126-163  unpack_bits!
//   util/packed/packed_simd.rs:347-374  trans_to_delta / trans_from_delta (4-lane prefix sum + carried base)
//   util/packed/packed_simd.rs:376-392  max_bits_num
//   util/packed/packed_misc.rs:365-454  Format::{Packed,PackedSingleBlock}, byte_count
//   util/packed/packed_misc.rs:474-531  FormatAndBits::fastest
//   util/packed/packed_misc.rs:2405-2440, 2556-2582, 2655-2680  BulkOperationPacked new / encode_int_to_byte / decode_byte_to_int
//   util/packed/packed_misc.rs:2686-2860  BulkOperationPackedSingleBlock
//   codec/postings/simd_block_decoder.rs:100-128  SIMDBlockDecoder::advance (count-of-less-than)
//
// The __m128i operations are restated on a plain 4 x u32 struct (one struct == one SSE register), so the
// instruction sequence of the macros is kept one-to-one; no x86 intrinsics are needed.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "store.hpp"

namespace orc {

constexpr int BLOCK_SIZE = 128;           // codec/postings/posting_format.rs:36
constexpr int MAX_ENCODED_SIZE = 128 * 4;  // for_util.rs:33

struct V4 {  // one __m128i viewed as 4 x u32
  uint32_t v[4];
};
inline V4 v4_load(const uint8_t* p) {  // _mm_lddqu_si128: little-endian u32 lanes
  V4 r;
  std::memcpy(r.v, p, 16);
  return r;
}
inline void v4_store(uint8_t* p, const V4& a) { std::memcpy(p, a.v, 16); }
inline V4 v4_set1(uint32_t x) { return V4{{x, x, x, x}}; }
inline V4 v4_or(V4 a, V4 b) { return V4{{a.v[0] | b.v[0], a.v[1] | b.v[1], a.v[2] | b.v[2], a.v[3] | b.v[3]}}; }
inline V4 v4_and(V4 a, V4 b) { return V4{{a.v[0] & b.v[0], a.v[1] & b.v[1], a.v[2] & b.v[2], a.v[3] & b.v[3]}}; }
inline V4 v4_add(V4 a, V4 b) { return V4{{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]}}; }
inline V4 v4_sub(V4 a, V4 b) { return V4{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}; }
// _mm_slli_epi32 / _mm_srli_epi32: a count >= 32 yields 0 (SSE semantics)
inline V4 v4_slli(V4 a, int n) {
  if (n >= 32) return v4_set1(0);
  return V4{{a.v[0] << n, a.v[1] << n, a.v[2] << n, a.v[3] << n}};
}
inline V4 v4_srli(V4 a, int n) {
  if (n >= 32) return v4_set1(0);
  return V4{{a.v[0] >> n, a.v[1] >> n, a.v[2] >> n, a.v[3] >> n}};
}
// _mm_slli_si128(a, 4*k): shift whole register left by k lanes (towards higher lane index)
inline V4 v4_shl_lanes(V4 a, int k) {
  V4 r = v4_set1(0);
  for (int i = k; i < 4; i++) r.v[i] = a.v[i - k];
  return r;
}

// ---- SIMD128Packer -------------------------------------------------------------------------------------------

struct Simd128Packer {
  uint32_t delta_base = 0;

  // packed_simd.rs:347-361
  V4 trans_to_delta(V4 data) {
    V4 prev = v4_shl_lanes(data, 1);
    prev.v[0] |= delta_base;  // _mm_set_epi32(0,0,0,base) puts base in lane 0
    V4 deltas = v4_sub(data, prev);
    delta_base = data.v[3];
    return deltas;
  }
  // packed_simd.rs:363-374
  V4 trans_from_delta(V4 delta) {
    V4 a_ab_bc_cd = v4_add(delta, v4_shl_lanes(delta, 1));
    V4 a_ab_abc_abcd = v4_add(a_ab_bc_cd, v4_shl_lanes(a_ab_bc_cd, 2));
    V4 value = v4_add(a_ab_abc_abcd, v4_set1(delta_base));
    delta_base = value.v[3];
    return value;
  }

  // packed_simd.rs:81-108 (pack_bits!) ; `delta` selects the $transfer arm
  void pack_impl(const uint32_t* data, uint8_t* encoded, int num, bool delta) {
    const uint8_t* input = (const uint8_t*)data;
    uint8_t* output = encoded;
    V4 buffer = v4_set1(0);
    for (int i = 0; i < 32; i++) {
      V4 input_data = v4_load(input);
      if (delta) input_data = trans_to_delta(input_data);
      const int inner_pos = i * num % 32;
      buffer = v4_or(buffer, v4_slli(input_data, inner_pos));
      const int new_pos = inner_pos + num;
      if (new_pos >= 32) {  // buffer is full, store
        v4_store(output, buffer);
        output += 16;
        buffer = (new_pos > 32) ? v4_srli(input_data, 32 - inner_pos) : v4_set1(0);
      }
      input += 16;
    }
  }

  // packed_simd.rs:126-163 (unpack_bits!)
  void unpack_impl(const uint8_t* encoded, uint32_t* data, int num, bool delta) {
    const uint8_t* input = encoded;
    uint8_t* output = (uint8_t*)data;
    const V4 mask = v4_set1((uint32_t)((1ull << num) - 1));
    V4 buffer = v4_load(input);
    for (int i = 0; i < 32; i++) {
      const int inner_pos = i * num % 32;
      const int new_pos = inner_pos + num;
      if (new_pos >= 32) {
        input += 16;
        if (new_pos == 32) {
          if (delta) buffer = trans_from_delta(buffer);
          v4_store(output, buffer);
          // The macro reads one vector past the payload after the last row; the reference hands it
          // a raw mmap pointer. Guard the restatement instead of over-reading.
          buffer = (i == 31) ? v4_set1(0) : v4_load(input);
        } else {
          const int remain = 32 - inner_pos;
          V4 temp = v4_load(input);
          buffer = v4_and(v4_or(buffer, v4_slli(temp, remain)), mask);
          if (delta) buffer = trans_from_delta(buffer);
          v4_store(output, buffer);
          buffer = v4_srli(temp, num - remain);
        }
      } else {
        V4 d = v4_and(buffer, mask);
        if (delta) d = trans_from_delta(d);
        v4_store(output, d);
        buffer = v4_srli(buffer, num);
      }
      output += 16;
    }
  }

  // packed_simd.rs:169-207
  static void pack(const uint32_t* data, uint8_t* encoded, int bits_num) {
    if (bits_num == 0) return;
    if (bits_num == 32) { std::memcpy(encoded, data, 512); return; }
    if (bits_num < 0 || bits_num > 32) throw OracleError(E_ILLEGAL_ARGUMENT, "unimplemented bit width");
    Simd128Packer p;
    p.pack_impl(data, encoded, bits_num, false);
  }
  // packed_simd.rs:209-252
  static void unpack(const uint8_t* encoded, uint32_t* data, int bits_num) {
    if (bits_num == 0) return;
    if (bits_num == 32) { std::memcpy(data, encoded, 512); return; }
    if (bits_num < 0 || bits_num > 32) throw OracleError(E_ILLEGAL_ARGUMENT, "unimplemented bit width");
    Simd128Packer p;
    p.unpack_impl(encoded, data, bits_num, false);
  }
  // packed_simd.rs:254-293
  void delta_pack(const uint32_t* data, uint8_t* encoded, uint32_t base, int bits_num) {
    delta_base = base;
    if (bits_num == 0) return;
    if (bits_num == 32) { std::memcpy(encoded, data, 512); return; }
    pack_impl(data, encoded, bits_num, true);
  }
  // packed_simd.rs:295-345
  void delta_unpack(const uint8_t* encoded, uint32_t* data, uint32_t base, int bits_num) {
    delta_base = base;
    if (bits_num == 0) return;
    if (bits_num == 32) { std::memcpy(data, encoded, 512); return; }
    unpack_impl(encoded, data, bits_num, true);
  }
  // packed_simd.rs:36-41 (trait default) and :376-392 (SSE version) agree: 32 - clz(OR of all)
  static int max_bits_num(const uint32_t* data, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; i++) r |= data[i];
    return r == 0 ? 0 : 32 - __builtin_clz(r);
  }
};

// simd_block_decoder.rs:100-128 — position of first element >= target in a sorted 128 block
// = number of elements < target (signed compare, _mm_cmplt_epi32).
inline int simd_block_advance(const int32_t* block, int32_t target) {
  int count = 0;
  for (int i = 0; i < BLOCK_SIZE; i++) count += (block[i] < target) ? 1 : 0;
  return count;
}

// ---- legacy Lucene PackedInts (".doc" version 0) --------------------------------------------------------------

enum PackedFormat { FMT_PACKED = 0, FMT_PACKED_SINGLE_BLOCK = 1 };  // packed_misc.rs:365-393

inline bool psb_is_supported(int bpv) {  // Packed64SingleBlock::is_supported
  static const int s[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
  for (int x : s) if (x == bpv) return true;
  return false;
}

// packed_misc.rs:395-409
inline int64_t format_byte_count(int format, int value_count, int bpv) {
  if (format == FMT_PACKED) return ((int64_t)value_count * bpv + 7) / 8;
  int values_per_block = 64 / bpv;
  return (int64_t)((value_count + values_per_block - 1) / values_per_block) * 8;
}

struct FormatAndBits { int format; int bits_per_value; };

// packed_misc.rs:474-531 (COMPACT = 0.0, FASTEST = 7.0)
inline FormatAndBits format_fastest(int value_count, int bits_per_value, float acceptable_overhead_ratio) {
  if (value_count == -1) value_count = INT32_MAX;
  acceptable_overhead_ratio = std::max(0.0f, acceptable_overhead_ratio);
  acceptable_overhead_ratio = std::min(7.0f, acceptable_overhead_ratio);
  float acceptable_overhead_per_value = acceptable_overhead_ratio * (float)bits_per_value;
  int max_bits_per_value = bits_per_value + (int)acceptable_overhead_per_value;
  int actual = -1;
  int format = FMT_PACKED;
  const int three_blocks_max = INT32_MAX / 3;
  if (bits_per_value <= 8 && max_bits_per_value >= 8) actual = 8;
  else if (bits_per_value <= 16 && max_bits_per_value >= 16) actual = 16;
  else if (bits_per_value <= 32 && max_bits_per_value >= 32) actual = 32;
  else if (bits_per_value <= 64 && max_bits_per_value >= 64) actual = 64;
  else if (value_count <= three_blocks_max && bits_per_value <= 24 && max_bits_per_value >= 24) actual = 24;
  else if (value_count <= three_blocks_max && bits_per_value <= 48 && max_bits_per_value >= 48) actual = 48;
  else {
    for (int bpv = bits_per_value; bpv <= max_bits_per_value; bpv++) {
      if (psb_is_supported(bpv)) {
        int vpb = 64 / bpv;
        float overhead = (float)(64 % bpv) / (float)vpb;
        float acceptable = acceptable_overhead_per_value + (float)bits_per_value - (float)bpv;
        if (overhead <= acceptable) { actual = bpv; format = FMT_PACKED_SINGLE_BLOCK; break; }
      }
    }
    if (actual < 0) actual = bits_per_value;
  }
  return FormatAndBits{format, actual};
}

// packed_misc.rs:2405-2440
struct BulkOperationPacked {
  int bits_per_value;
  int byte_block_count, byte_value_count;
  int32_t int_mask;
  explicit BulkOperationPacked(int bpv) : bits_per_value(bpv) {
    int blocks = bpv;
    while ((blocks & 1) == 0) blocks >>= 1;
    int long_value_count = 64 * blocks / bpv;
    byte_block_count = 8 * blocks;
    byte_value_count = long_value_count;
    while ((byte_block_count & 1) == 0 && (byte_value_count & 1) == 0) {
      byte_block_count >>= 1;
      byte_value_count >>= 1;
    }
    int64_t mask = (bpv == 64) ? -1 : (int64_t)((1ull << bpv) - 1);
    int_mask = (int32_t)mask;
  }
  // packed_misc.rs:2556-2582
  void encode_int_to_byte(const int32_t* values, uint8_t* blocks, int iterations) const {
    int32_t next_block = 0;
    int bits_left = 8;
    int vo = 0, bo = 0;
    for (int i = 0; i < byte_value_count * iterations; i++) {
      int32_t v = values[vo++];
      if (bits_per_value < bits_left) {
        next_block |= v << (bits_left - bits_per_value);
        bits_left -= bits_per_value;
      } else {
        int bits = bits_per_value - bits_left;
        blocks[bo++] = (uint8_t)(next_block | (int32_t)((uint32_t)v >> bits));
        while (bits >= 8) {
          bits -= 8;
          blocks[bo++] = (uint8_t)((uint32_t)v >> bits);
        }
        bits_left = 8 - bits;
        next_block = (v & ((1 << bits) - 1)) << bits_left;
      }
    }
  }
  // packed_misc.rs:2655-2680
  void decode_byte_to_int(const uint8_t* blocks, int32_t* values, int iterations) const {
    int32_t next_value = 0;
    int bits_left = bits_per_value;
    int vo = 0, bo = 0;
    for (int i = 0; i < iterations * byte_block_count; i++) {
      int32_t bytes = blocks[bo++];
      if (bits_left > 8) {
        bits_left -= 8;
        next_value |= bytes << bits_left;
      } else {
        int bits = 8 - bits_left;
        values[vo++] = next_value | (bytes >> bits);
        while (bits >= bits_per_value) {
          bits -= bits_per_value;
          values[vo++] = (bytes >> bits) & int_mask;
        }
        bits_left = bits_per_value - bits;
        next_value = (bytes & ((1 << bits) - 1)) << bits_left;
      }
    }
  }
};

// packed_misc.rs:2686-2860
struct BulkOperationPackedSingleBlock {
  int bits_per_value, value_count;
  int64_t mask;
  explicit BulkOperationPackedSingleBlock(int bpv)
      : bits_per_value(bpv), value_count(64 / bpv), mask((int64_t)((1ull << bpv) - 1)) {}
  int byte_block_count() const { return 8; }
  int byte_value_count() const { return value_count; }
  static int64_t read_long(const uint8_t* b, int off) {
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r = (r << 8) | b[off + i];
    return (int64_t)r;
  }
  void decode_byte_to_int(const uint8_t* blocks, int32_t* values, int iterations) const {
    int vo = 0;
    for (int i = 0; i < iterations; i++) {
      uint64_t block = (uint64_t)read_long(blocks, i * 8);
      values[vo++] = (int32_t)(block & (uint64_t)mask);
      for (int j = 1; j < value_count; j++) {
        block >>= bits_per_value;
        values[vo++] = (int32_t)(block & (uint64_t)mask);
      }
    }
  }
  void encode_int_to_byte(const int32_t* values, uint8_t* blocks, int iterations) const {
    int bo = 0;
    for (int i = 0; i < iterations; i++) {
      int off = i * value_count;
      uint64_t block = (uint32_t)values[off++];
      for (int j = 1; j < value_count; j++) block |= (uint64_t)(uint32_t)values[off++] << (j * bits_per_value);
      for (int k = 1; k < 9; k++) blocks[bo++] = (uint8_t)(block >> (64 - (k << 3)));
    }
  }
};

// for_util.rs:60-62 compute_iterations = ceil(BLOCK_SIZE / byte_value_count)
inline int compute_iterations(int byte_value_count) {
  return (int)std::ceil((float)BLOCK_SIZE / (float)byte_value_count);
}

// for_util.rs:64-97 max_data_size() — must equal the MAX_DATA_SIZE = 147 constant (for_util.rs:42,53-56)
inline int max_data_size() {
  int m = 0;
  for (int bpv = 1; bpv <= 32; bpv++) {
    BulkOperationPacked p(bpv);
    m = std::max(m, compute_iterations(p.byte_value_count) * p.byte_value_count);
  }
  for (int bpv = 1; bpv <= 32; bpv++) {
    BulkOperationPackedSingleBlock p(bpv);  // get_decoder (packed_misc.rs:2256-2259) never rejects a width
    m = std::max(m, compute_iterations(p.byte_value_count()) * p.byte_value_count());
  }
  return m;
}
constexpr int MAX_DATA_SIZE = 147;  // for_util.rs:42

}  // namespace orc
