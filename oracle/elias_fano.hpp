// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's Elias-Fano block codec as the postings use it for a 128-doc block (EncodeType::EF), and
// of the sequential half of its decoder. Paths relative to /root/reference/src/core/util/packed:
//   elias_fano_encoder.rs:47-146    EliasFanoEncoder::new (num_low_bits = floor(log2(upper_bound / num_values)) — "different
//                                   from lucene version"; upper / lower / index long counts)
//   elias_fano_encoder.rs:204-253   get_encoder (index interval 256), encode_size, encode_next
//   elias_fano_encoder.rs:255-287   serialize (type byte 0x40, vlong upper_bound, the three long arrays as raw native-endian
//                                   bytes), deserialize2
//   elias_fano_encoder.rs:309-346   num_longs_for_bits, encode_upper_bits, encode_lower_bits, pack_value
//   elias_fano_decoder.rs:95-169    combine_high_low_values, unpack_value, to_after_current_high_bit, to_next_high_long,
//                                   to_next_high_value, next_value
// PINNED by the reference's own tests (elias_fano_encoder.rs:397-447: num_longs_for_bits, pack_value's two vectors,
// encode_upper's 1, 3, 11, 27, 91) — tests/test_oracle_kat.py ports them. advance_to_value / back_to_value
// (elias_fano_decoder.rs:171-433) are NOT restated: nothing writes EF blocks (posting_writer.rs:46 use_ef = false, never set),
// and the GPU path converts such a block to plain deltas once, at prepare time.
#pragma once
#include <cstdint>
#include <vector>

#include "store.hpp"

namespace orc {

constexpr int64_t EF_DEFAULT_INDEX_INTERVAL = 256;  // elias_fano_encoder.rs:21
constexpr int64_t EF_NO_MORE_VALUES = -1;           // elias_fano_decoder.rs:22

struct EliasFanoEncoder {
  int64_t num_values = 0, upper_bound = 0;
  int32_t num_low_bits = 0;
  int64_t lower_bits_mask = 0;
  std::vector<int64_t> upper_longs, lower_longs, upper_zero_bit_position_index;
  int64_t num_encoded = 0, last_encoded = 0, num_index_entries = 0, index_interval = EF_DEFAULT_INDEX_INTERVAL;
  int32_t n_index_entry_bits = 0;
  int64_t current_entry_index = 0;

  static int64_t num_longs_for_bits(int64_t n) { return (int64_t)((uint64_t)(n + 63) >> 6); }  // :309-312
  static int lz64(int64_t v) { return v == 0 ? 64 : __builtin_clzll((uint64_t)v); }

  // elias_fano_encoder.rs:47-146
  EliasFanoEncoder(int64_t num_values_, int64_t upper_bound_, int64_t index_interval_ = EF_DEFAULT_INDEX_INTERVAL) {
    if (num_values_ < 0) throw OracleError(E_ILLEGAL_ARGUMENT, "num_values should not be negative");
    if (num_values_ > 0 && upper_bound_ < 0) throw OracleError(E_ILLEGAL_ARGUMENT, "upper_bound should not be negative");
    num_values = num_values_;
    upper_bound = num_values_ > 0 ? upper_bound_ : -1;
    num_low_bits = 0;
    if (num_values > 0) {
      const int64_t low_bits_fac = upper_bound / num_values;
      if (low_bits_fac > 0) num_low_bits = 64 - 1 - lz64(low_bits_fac);
    }
    lower_bits_mask = (int64_t)((uint64_t)INT64_MAX >> (64 - 1 - num_low_bits));
    lower_longs.assign((size_t)num_longs_for_bits(num_values * num_low_bits), 0);
    int64_t num_high_bits_clear = upper_bound > 0 ? upper_bound : 0;
    num_high_bits_clear = (int64_t)((uint64_t)num_high_bits_clear >> num_low_bits);
    if (!(num_high_bits_clear <= 2 * num_values)) throw OracleError(E_ILLEGAL_STATE, "num_high_bits_clear > 2 * num_values");
    upper_longs.assign((size_t)num_longs_for_bits(num_high_bits_clear + num_values), 0);
    if (index_interval_ < 2) throw OracleError(E_ILLEGAL_ARGUMENT, "index_interval should at least 2");
    index_interval = index_interval_;
    const int64_t max_high_value = (int64_t)((uint64_t)upper_bound >> num_low_bits);
    const int64_t n_index_entries = max_high_value / index_interval;
    num_index_entries = n_index_entries >= 0 ? n_index_entries : 0;
    const int64_t max_index_entry = max_high_value + num_values - 1;
    n_index_entry_bits = max_index_entry <= 0 ? 0 : 64 - lz64(max_index_entry);
    upper_zero_bit_position_index.assign((size_t)num_longs_for_bits(num_index_entries * n_index_entry_bits), 0);
  }
  // :209-214
  int32_t encode_size() const { return (int32_t)((upper_longs.size() + lower_longs.size() + upper_zero_bit_position_index.size()) << 3); }
  // :335-346 (the spill into the next long is an assignment in the reference, not an OR: that long is still empty)
  static void pack_value(int64_t value, std::vector<int64_t>& long_array, int32_t num_bits, int64_t pack_index) {
    if (num_bits != 0) {
      const int64_t bit_pos = (int64_t)num_bits * pack_index;
      const size_t index = (size_t)((uint64_t)bit_pos >> 6);
      const int32_t bit_pos_at_index = (int32_t)(bit_pos & 63);
      long_array[index] |= (int64_t)((uint64_t)value << bit_pos_at_index);
      if (bit_pos_at_index + num_bits > 64) long_array[index + 1] = (int64_t)((uint64_t)value >> (64 - bit_pos_at_index));
    }
  }
  // :314-318
  void encode_upper_bits(int64_t high_value) {
    const int64_t next_high_bit_num = num_encoded + high_value;
    upper_longs[(size_t)((uint64_t)next_high_bit_num >> 6)] |= (int64_t)(1ull << (next_high_bit_num & 63));
  }
  // :320-327
  void encode_lower_bits(int64_t low_value) { pack_value(low_value, lower_longs, num_low_bits, num_encoded); }
  // :216-253
  void encode_next(int64_t x) {
    if (num_encoded >= num_values) throw OracleError(E_ILLEGAL_STATE, "encode_next called more than num_values times");
    if (last_encoded > x) throw OracleError(E_ILLEGAL_ARGUMENT, "smaller than previous");
    if (x > upper_bound) throw OracleError(E_ILLEGAL_ARGUMENT, "larger than upperBound");
    const int64_t high_value = (int64_t)((uint64_t)x >> num_low_bits);
    encode_upper_bits(high_value);
    encode_lower_bits(x & lower_bits_mask);
    last_encoded = x;
    int64_t index_value = (current_entry_index + 1) * index_interval;
    while (index_value <= high_value) {
      pack_value(index_value + num_encoded, upper_zero_bit_position_index, n_index_entry_bits, current_entry_index);
      current_entry_index += 1;
      index_value += index_interval;
    }
    num_encoded += 1;
  }
  // :348-357 write_data: the longs' native (little-endian) bytes
  static void write_data(const std::vector<int64_t>& data, ByteOut& out) {
    for (int64_t v : data) for (int i = 0; i < 8; i++) out.write_byte((uint8_t)((uint64_t)v >> (8 * i)));
  }
  static void read_data2(std::vector<int64_t>& buf, ByteIn& in) {  // :371-378
    for (int64_t& v : buf) { uint64_t x = 0; for (int i = 0; i < 8; i++) x |= (uint64_t)in.read_byte() << (8 * i); v = (int64_t)x; }
  }
  // :255-262 (EncodeType::EF << 6 = 0x40)
  void serialize(ByteOut& out) const {
    out.write_byte(0x40);
    out.write_vlong(upper_bound);
    write_data(upper_longs, out);
    write_data(lower_longs, out);
    write_data(upper_zero_bit_position_index, out);
  }
  // :279-287
  void deserialize2(ByteIn& in) {
    num_encoded = num_values;
    last_encoded = upper_bound;
    read_data2(upper_longs, in);
    read_data2(lower_longs, in);
    read_data2(upper_zero_bit_position_index, in);
  }
};

// elias_fano_decoder.rs:41-169, the forward-only part
struct EliasFanoDecoder {
  const EliasFanoEncoder* enc;
  int64_t num_encoded, ef_index = -1, set_bit_for_index = -1, cur_high_long = 0;
  explicit EliasFanoDecoder(const EliasFanoEncoder* e) : enc(e), num_encoded(e->num_encoded) {}
  int64_t current_index() const { return ef_index; }
  int64_t current_high_value() const { return set_bit_for_index - ef_index; }                      // :81-83
  static int64_t unpack_value(const std::vector<int64_t>& a, int32_t num_bits, int64_t pack_index, int64_t mask) {  // :99-112
    if (num_bits == 0) return 0;
    const int64_t bit_pos = pack_index * num_bits;
    const size_t index = (size_t)((uint64_t)bit_pos >> 6);
    const int64_t at = bit_pos & 63;
    uint64_t value = (uint64_t)a[index] >> at;
    if (at + num_bits > 64) value |= (uint64_t)a[index + 1] << (64 - at);
    return (int64_t)value & mask;
  }
  int64_t current_low_value() const { return unpack_value(enc->lower_longs, enc->num_low_bits, ef_index, enc->lower_bits_mask); }  // :85-93
  bool to_after_current_high_bit() {  // :125-138
    ef_index += 1;
    if (ef_index >= num_encoded) return false;
    set_bit_for_index += 1;
    cur_high_long = (int64_t)((uint64_t)enc->upper_longs[(size_t)((uint64_t)set_bit_for_index >> 6)] >> (set_bit_for_index & 63));
    return true;
  }
  void to_next_high_long() {  // :140-147
    set_bit_for_index += 64 - (set_bit_for_index & 63);
    cur_high_long = enc->upper_longs[(size_t)((uint64_t)set_bit_for_index >> 6)];
  }
  void to_next_high_value() {  // :149-154
    while (cur_high_long == 0) to_next_high_long();
    set_bit_for_index += __builtin_ctzll((uint64_t)cur_high_long);
  }
  int64_t next_value() {  // :163-169
    if (!to_after_current_high_bit()) return EF_NO_MORE_VALUES;
    to_next_high_value();
    return (current_high_value() << enc->num_low_bits) | current_low_value();
  }
};

}  // namespace orc
