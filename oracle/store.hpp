// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's byte-level I/O grammar (DataInput / DataOutput / codec header+footer).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// Follows (paths relative to /root/reference/src/core):
//   store/io/data_output.rs:39-95   write_short/int/vint/long/vlong/string (big-endian fixed ints,
//                                   7-bit little-endian groups for VInt/VLong)
//   store/io/data_input.rs:78-199   read_vint (5 bytes max, high nibble of 5th must be 0),
//                                   read_vlong (9 bytes max, negative not allowed)
//   codec/codec_util.rs:46-120      write_header / write_index_header / write_footer
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

constexpr int32_t CODEC_MAGIC = 0x3FD76C17;  // codec_util.rs:30
constexpr int32_t FOOTER_MAGIC = ~CODEC_MAGIC;  // codec_util.rs:32
constexpr int ID_LENGTH = 16;

struct OracleError : std::runtime_error {
  int kind;  // mirrors error.rs ErrorKind ordinal used by the C API (see oracle_capi.cpp)
  OracleError(int k, const std::string& m) : std::runtime_error(m), kind(k) {}
};
enum ErrKind { E_ILLEGAL_STATE = 1, E_ILLEGAL_ARGUMENT = 2, E_UNEXPECTED_EOF = 3, E_CORRUPT_INDEX = 4,
               E_UNSUPPORTED = 5 };

// CRC32 (IEEE, zlib polynomial) — store/io/fs_index_output.rs:77-80 uses crc32fast over all bytes.
inline uint32_t crc32_ieee(const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// DataOutput over a growable byte vector (stands in for IndexOutput / RAMOutputStream).
struct ByteOut {
  std::vector<uint8_t> buf;
  int64_t file_pointer() const { return (int64_t)buf.size(); }
  void reset() { buf.clear(); }
  void write_byte(uint8_t b) { buf.push_back(b); }
  void write_bytes(const uint8_t* p, size_t n) { buf.insert(buf.end(), p, p + n); }
  // data_output.rs:45-49 — big-endian
  void write_int(int32_t i) {
    uint32_t u = (uint32_t)i;
    uint8_t b[4] = {(uint8_t)(u >> 24), (uint8_t)(u >> 16), (uint8_t)(u >> 8), (uint8_t)u};
    write_bytes(b, 4);
  }
  // data_output.rs:39-43
  void write_short(int16_t i) {
    write_byte((uint8_t)((uint16_t)i >> 8));
    write_byte((uint8_t)(uint16_t)i);
  }
  // data_output.rs:64-68
  void write_long(int64_t i) {
    write_int((int32_t)((uint64_t)i >> 32));
    write_int((int32_t)(uint64_t)i);
  }
  // data_output.rs:51-58
  void write_vint(int32_t v) {
    uint32_t i = (uint32_t)v;
    while ((i & ~0x7Fu) != 0) {
      write_byte((uint8_t)((i & 0x7F) | 0x80));
      i >>= 7;
    }
    write_byte((uint8_t)i);
  }
  // data_output.rs:70-84
  void write_vlong(int64_t v) {
    if (v < 0) throw OracleError(E_ILLEGAL_ARGUMENT, "Can't write negative vLong");
    uint64_t i = (uint64_t)v;
    while ((i & ~0x7FULL) != 0) {
      write_byte((uint8_t)((i & 0x7F) | 0x80));
      i >>= 7;
    }
    write_byte((uint8_t)i);
  }
  // data_output.rs:90-95
  void write_string(const std::string& s) {
    write_vint((int32_t)s.size());
    write_bytes((const uint8_t*)s.data(), s.size());
  }
  void write_to(ByteOut& out) const { out.write_bytes(buf.data(), buf.size()); }
};

// DataInput + IndexInput over an in-memory slice (stands in for MmapIndexInput).
struct ByteIn {
  const uint8_t* data = nullptr;
  int64_t len = 0;
  int64_t pos = 0;
  ByteIn() {}
  ByteIn(const uint8_t* d, int64_t l, int64_t p = 0) : data(d), len(l), pos(p) {}
  int64_t file_pointer() const { return pos; }
  void seek(int64_t p) { pos = p; }
  uint8_t read_byte() {
    if (pos >= len) throw OracleError(E_UNEXPECTED_EOF, "read past EOF");
    return data[pos++];
  }
  const uint8_t* get_and_advance(size_t n) {  // mmap_index_input.rs:246-251
    if (pos + (int64_t)n > len) throw OracleError(E_UNEXPECTED_EOF, "read past EOF");
    const uint8_t* p = data + pos;
    pos += (int64_t)n;
    return p;
  }
  void read_exact(uint8_t* dst, size_t n) { std::memcpy(dst, get_and_advance(n), n); }
  int32_t read_int() {  // data_input.rs:66-76 (big-endian)
    const uint8_t* p = get_and_advance(4);
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]);
  }
  int16_t read_short() {  // data_input.rs:59-64 (big-endian)
    const uint8_t* p = get_and_advance(2);
    return (int16_t)(uint16_t)(((uint16_t)p[0] << 8) | p[1]);
  }
  int64_t read_long() {
    int64_t hi = (uint32_t)read_int();
    int64_t lo = (uint32_t)read_int();
    return (hi << 32) | lo;
  }
  // data_input.rs:78-111
  int32_t read_vint() {
    int8_t b = (int8_t)read_byte();
    if (b >= 0) return b;
    int32_t i = b & 0x7F;
    b = (int8_t)read_byte();
    i |= (b & 0x7F) << 7;
    if (b >= 0) return i;
    b = (int8_t)read_byte();
    i |= (b & 0x7F) << 14;
    if (b >= 0) return i;
    b = (int8_t)read_byte();
    i |= (b & 0x7F) << 21;
    if (b >= 0) return i;
    b = (int8_t)read_byte();
    i |= (int32_t)((uint32_t)(b & 0x0F) << 28);
    if (((uint8_t)b & 0xF0) != 0) throw OracleError(E_ILLEGAL_STATE, "Invalid vInt detected");
    return i;
  }
  // data_input.rs:127-199 (negative_allowed = false)
  int64_t read_vlong() {
    int8_t b = (int8_t)read_byte();
    if (b >= 0) return b;
    int64_t i = b & 0x7F;
    for (int shift = 7; shift <= 56; shift += 7) {
      b = (int8_t)read_byte();
      i |= (int64_t)(b & 0x7F) << shift;
      if (b >= 0) return i;
    }
    throw OracleError(E_ILLEGAL_STATE, "Invalid vLong detected");
  }
  std::string read_string() {
    int32_t n = read_vint();
    const uint8_t* p = get_and_advance((size_t)n);
    return std::string((const char*)p, (size_t)n);
  }
};

// codec_util.rs:46-103
inline void write_index_header(ByteOut& out, const std::string& codec, int32_t version,
                               const uint8_t id[ID_LENGTH], const std::string& suffix) {
  if (codec.size() >= 128) throw OracleError(E_ILLEGAL_ARGUMENT, "codec name too long");
  out.write_int(CODEC_MAGIC);
  out.write_string(codec);
  out.write_int(version);
  out.write_bytes(id, ID_LENGTH);
  if (suffix.size() >= 256) throw OracleError(E_ILLEGAL_ARGUMENT, "suffix too long");
  out.write_byte((uint8_t)suffix.size());
  out.write_bytes((const uint8_t*)suffix.data(), suffix.size());
}

// codec_util.rs:110-120 + write_crc: i64 CRC32 of all preceding bytes (magic + algorithm id included)
inline void write_footer(ByteOut& out) {
  out.write_int(FOOTER_MAGIC);
  out.write_int(0);
  uint32_t crc = crc32_ieee(out.buf.data(), out.buf.size());
  out.write_long((int64_t)crc);
}

// codec_util.rs check_index_header: returns version; validates magic, codec, version range.
inline int32_t check_index_header(ByteIn& in, const std::string& codec, int32_t min_version, int32_t max_version) {
  int32_t magic = in.read_int();
  if (magic != CODEC_MAGIC) throw OracleError(E_CORRUPT_INDEX, "codec header mismatch");
  std::string actual = in.read_string();
  if (actual != codec) throw OracleError(E_CORRUPT_INDEX, "codec mismatch: " + actual);
  int32_t version = in.read_int();
  if (version < min_version || version > max_version) throw OracleError(E_CORRUPT_INDEX, "version out of range");
  in.get_and_advance(ID_LENGTH);
  uint8_t slen = in.read_byte();
  in.get_and_advance(slen);
  return version;
}

// codec_util.rs footer_length(): magic + algorithm id + i64 CRC
constexpr int FOOTER_LENGTH = 16;

// codec_util.rs:340-353 retrieve_checksum (+ validate_footer :275-305, read_crc): locates and sanity-checks the
// footer without hashing the file; returns the stored CRC.
inline int64_t retrieve_checksum(const uint8_t* data, size_t len) {
  if (len < (size_t)FOOTER_LENGTH) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer (file truncated?)");
  ByteIn f(data + len - FOOTER_LENGTH, FOOTER_LENGTH);
  if (f.read_int() != FOOTER_MAGIC) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch");
  if (f.read_int() != 0) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch: unknown algorithm_id");
  const int64_t crc = f.read_long();
  if ((uint64_t)crc & 0xFFFFFFFF00000000ull) throw OracleError(E_CORRUPT_INDEX, "Illegal CRC-32 checksum");
  return crc;
}

}  // namespace orc
