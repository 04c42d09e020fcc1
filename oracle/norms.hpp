// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/search.hpp). Restatement of Rucene's Lucene53 norms files
// (".nvm" metadata + ".nvd" data), paths relative to /root/reference/src/core:
//   codec/norms/norms.rs:23-28                 codec names, extensions, VERSION_START = VERSION_CURRENT = 0
//   codec/norms/norms_consumer.rs:33-159       Lucene53NormsConsumer (writer): add_norms_field / add_constant / add_byte, Drop
//   codec/norms/norms_producer.rs:40-189       Lucene53NormsProducer (reader): new / read_fields / norms()
// PARITY UNPINNED: the reference has no test for these files; the source text is the only authority.
#pragma once
#include <map>
#include <vector>

#include "store.hpp"

namespace orc {

static const char* const NORMS_DATA_CODEC = "Lucene53NormsData";
static const char* const NORMS_META_CODEC = "Lucene53NormsMetadata";
constexpr int32_t NORMS_VERSION_START = 0, NORMS_VERSION_CURRENT = 0;

// Writer: new() writes both index headers; add_norms_field per field; finish() == Drop (EOF marker + footers).
struct NormsConsumer {
  ByteOut data, meta;
  int32_t max_doc;
  NormsConsumer(int32_t max_doc_, const uint8_t id[ID_LENGTH], const std::string& suffix) : max_doc(max_doc_) {
    write_index_header(data, NORMS_DATA_CODEC, NORMS_VERSION_CURRENT, id, suffix);   // norms_consumer.rs:46-66
    write_index_header(meta, NORMS_META_CODEC, NORMS_VERSION_CURRENT, id, suffix);
  }
  void add_norms_field(int32_t field_number, const std::vector<int64_t>& values) {  // :117-147
    meta.write_vint(field_number);
    int64_t min_value = INT64_MAX, max_value = INT64_MIN;
    for (int64_t v : values) { min_value = std::min(min_value, v); max_value = std::max(max_value, v); }
    if ((int64_t)values.size() != (int64_t)max_doc) throw OracleError(E_ILLEGAL_ARGUMENT, "illegal norms data: count != max_doc");
    if (min_value == max_value) {  // add_constant :79-82
      meta.write_byte(0);
      meta.write_long(min_value);
      return;
    }
    int len;  // add_byte :84-113
    if (min_value >= INT8_MIN && max_value <= INT8_MAX) len = 1;
    else if (min_value >= INT16_MIN && max_value <= INT16_MAX) len = 2;
    else if (min_value >= INT32_MIN && max_value <= INT32_MAX) len = 4;
    else len = 8;
    meta.write_byte((uint8_t)len);
    meta.write_long((int64_t)data.buf.size());  // data.file_pointer()
    for (int64_t v : values) {
      switch (len) {
        case 1: data.write_byte((uint8_t)(int8_t)v); break;
        case 2: data.write_short((int16_t)v); break;
        case 4: data.write_int((int32_t)v); break;
        default: data.write_long(v); break;
      }
    }
  }
  void finish() {  // Drop :150-158
    meta.write_vint(-1);
    write_footer(meta);
    write_footer(data);
  }
};

// Reader. check_footer verifies the metadata file's CRC; the data file's footer is only located (retrieve_checksum).
struct NormsProducer {
  struct Entry { uint8_t bytes_per_value; uint64_t offset; };
  std::map<int32_t, Entry> entries;
  std::vector<uint8_t> data;
  int32_t max_doc;
  NormsProducer(const uint8_t* nvm, size_t nvm_len, const uint8_t* nvd, size_t nvd_len, int32_t max_doc_) : max_doc(max_doc_) {
    ByteIn m(nvm, (int64_t)nvm_len);
    const int32_t meta_version = check_index_header(m, NORMS_META_CODEC, NORMS_VERSION_START, NORMS_VERSION_CURRENT);
    while (true) {  // read_fields :108-140
      const int32_t field_num = m.read_vint();
      if (field_num == -1) break;
      const uint8_t bpv = m.read_byte();
      if (!(bpv == 0 || bpv == 1 || bpv == 2 || bpv == 4 || bpv == 8)) throw OracleError(E_CORRUPT_INDEX, "Invalid field number");
      const uint64_t off = (uint64_t)m.read_long();
      entries[field_num] = Entry{bpv, off};
    }
    // check_footer (codec_util.rs:310-321)
    if (nvm_len < 16 || m.pos != (int64_t)nvm_len - 16) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer");
    if (m.read_int() != FOOTER_MAGIC || m.read_int() != 0) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch");
    const int64_t expected = m.read_long();
    if ((uint64_t)expected & 0xFFFFFFFF00000000ull) throw OracleError(E_CORRUPT_INDEX, "Illegal CRC-32 checksum");
    if ((int64_t)crc32_ieee(nvm, nvm_len - 8) != expected) throw OracleError(E_CORRUPT_INDEX, "checksum failed");
    ByteIn d(nvd, (int64_t)nvd_len);
    const int32_t data_version = check_index_header(d, NORMS_DATA_CODEC, NORMS_VERSION_START, NORMS_VERSION_CURRENT);
    if (data_version != meta_version) throw OracleError(E_CORRUPT_INDEX, "Format versions mismatch");
    if (nvd_len < 16) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer (file truncated?)");  // retrieve_checksum
    ByteIn f(nvd + nvd_len - 16, 16);
    if (f.read_int() != FOOTER_MAGIC || f.read_int() != 0) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch");
    data.assign(nvd, nvd + nvd_len);
  }
  // norms(field).get(doc) :143-189
  int64_t get(int32_t field_number, int32_t doc) const {
    auto it = entries.find(field_number);
    if (it == entries.end()) throw OracleError(E_ILLEGAL_ARGUMENT, "no norms for field");
    const Entry& e = it->second;
    if (e.bytes_per_value == 0) return (int64_t)e.offset;
    const size_t at = (size_t)e.offset + (size_t)doc * e.bytes_per_value;
    if (at + e.bytes_per_value > data.size()) throw OracleError(E_UNEXPECTED_EOF, "norms slice out of range");
    ByteIn in(data.data() + at, e.bytes_per_value);
    switch (e.bytes_per_value) {
      case 1: return (int64_t)(int8_t)in.read_byte();
      case 2: return (int64_t)in.read_short();
      case 4: return (int64_t)in.read_int();
      default: return in.read_long();
    }
  }
};

// ---- Lucene50LiveDocsFormat (codec/live_docs.rs:63-161) ----------------------------------------------------------
// PARITY UNPINNED as well (no reference test). write_live_docs :123-150, read_live_docs :81-121; to_base36
// util/numeric.rs:148-161; bits2words util/bit_set.rs:480-484.
inline std::string to_base36(uint64_t val) {
  static const char digits[] = "0123456789abcdefghijklmnopqrstuvwxyz";
  std::string r;
  while (true) {
    r.push_back(digits[val % 36]);
    val /= 36;
    if (val == 0) break;
  }
  return std::string(r.rbegin(), r.rend());
}
inline std::vector<uint8_t> write_live_docs(const std::vector<int64_t>& words, int32_t max_doc, int32_t del_count_total,
                                            const uint8_t id[ID_LENGTH], uint64_t gen) {
  int64_t live = 0;
  for (int64_t w : words) live += __builtin_popcountll((uint64_t)w);
  if ((int64_t)max_doc - live != del_count_total) throw OracleError(E_CORRUPT_INDEX, "bits.deleted != del_count + new_del_count");
  ByteOut out;
  write_index_header(out, "Lucene50LiveDocs", 0, id, to_base36(gen));
  for (int64_t w : words) out.write_long(w);
  write_footer(out);
  return out.buf;
}
inline std::vector<int64_t> read_live_docs(const uint8_t* liv, size_t len, int32_t max_doc, int32_t del_count) {
  ByteIn in(liv, (int64_t)len);
  check_index_header(in, "Lucene50LiveDocs", 0, 0);
  const size_t num_words = (size_t)(((max_doc - 1) >> 6) + 1);
  std::vector<int64_t> bits;
  for (size_t i = 0; i < num_words; i++) bits.push_back(in.read_long());
  if (len < 16 || in.pos != (int64_t)len - 16) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer");
  if (in.read_int() != FOOTER_MAGIC || in.read_int() != 0) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch");
  const int64_t expected = in.read_long();
  if ((int64_t)crc32_ieee(liv, len - 8) != expected) throw OracleError(E_CORRUPT_INDEX, "checksum failed");
  int64_t live = 0;
  for (int64_t w : bits) live += __builtin_popcountll((uint64_t)w);
  if ((max_doc & 63) != 0 && ((uint64_t)bits.back() >> (max_doc & 63)) != 0) throw OracleError(E_ILLEGAL_STATE, "ghost bits");
  if ((int64_t)max_doc - live != del_count) throw OracleError(E_CORRUPT_INDEX, "bits.deleted != info.delcount");
  return bits;
}

}  // namespace orc
