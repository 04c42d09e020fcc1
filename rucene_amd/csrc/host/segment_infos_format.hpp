// Host-side readers of the two files that say what a Rucene index directory holds: the commit point "segments_N"
// (segment names, ids, codec, deletion generation and count) and each segment's ".si" (max_doc, compound flag, version).
// With ".fnm" (field_infos_format.hpp) they let a caller outside Rucene open a directory without being told anything.
// Mirrors (paths relative to /root/reference/src/core):
//   codec/segment_infos/segment_infos.rs:39-47, 443-569     segments_N: magic, "segments", format 4..6, id, suffix =
//                                                           base36(generation), [version triple], i64 version, i32 counter,
//                                                           i32 count, [min version], per segment: string name, u8 1, id,
//                                                           string codec, i64 del_gen, i32 del_count, i64 field_infos_gen,
//                                                           i64 dv_gen, set of strings, i32 n + (i32, set)*, then user-data
//                                                           map; validate_footer + check_checksum
//   codec/segment_infos/segment_infos_format.rs:37-247      .si: "Lucene62SegmentInfo" 0..1, id, empty suffix, version
//                                                           triple (i32 each), i32 doc count, u8 compound (1 yes / 0xff no),
//                                                           diagnostics map, files set, attributes map, index sort;
//                                                           validate_footer + check_checksum
//   codec/mod.rs:208-217                                    the only codec name accepted is "Lucene62"
// Error codes are rgpu_status values (include/rucene_gpu.h). No GPU involved.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "norms_format.hpp"

namespace rucene {

struct SegmentInfoEntry {
  int32_t max_doc = 0;
  bool is_compound_file = false;
  int32_t version[3] = {0, 0, 0};
  int32_t n_files = 0, n_sort_fields = 0;
  uint8_t id[16] = {0};
};

struct CommitSegmentEntry {
  std::string name, codec;
  uint8_t id[16] = {0};
  int64_t del_gen = -1, field_infos_gen = -1, dv_gen = -1;
  int32_t del_count = 0;
};

namespace detail {
struct FileCursor : Cursor {
  FileCursor(const uint8_t* p_, size_t len_) : Cursor{p_, len_} {}
  int64_t be64() {
    if (pos + 8 > len) { ok = false; return 0; }
    const int64_t v = (int64_t)be64_at(p + pos);
    pos += 8;
    return v;
  }
  bool string(std::string* s) {
    const uint32_t n = vint();
    if (!ok || pos + n > len) { ok = false; return false; }
    if (s) s->assign(reinterpret_cast<const char*>(p + pos), n);
    pos += n;
    return true;
  }
  bool string_set(int32_t* count) {
    const uint32_t n = vint();
    if (!ok || (int32_t)n < 0 || n > len) { ok = false; return false; }
    for (uint32_t i = 0; i < n && ok; ++i) string(nullptr);
    if (count) *count = (int32_t)n;
    return ok;
  }
  bool string_map() {
    const uint32_t n = vint();
    if (!ok || (int32_t)n < 0 || n > len) { ok = false; return false; }
    for (uint32_t i = 0; i < n && ok; ++i) { string(nullptr); string(nullptr); }
    return ok;
  }
};
inline int finish_checksummed_file(const uint8_t* data, size_t len, size_t body_end, const char* what, std::string* why) {
  uint64_t stored = 0;
  const int rc = read_footer(data, len, body_end, &stored, why);
  if (rc) return rc;
  if ((uint64_t)crc32_ieee(data, len - 8) != stored) { *why = std::string("checksum failed (hardware problems?) in ") + what; return -4; }
  return 0;
}
inline std::string base36(uint64_t v) {  // util/numeric.rs:148-160
  std::string r;
  do { r.insert(r.begin(), "0123456789abcdefghijklmnopqrstuvwxyz"[v % 36]); v /= 36; } while (v);
  return r;
}
}  // namespace detail

// expected_id: the id the commit point recorded for this segment (nullptr: not compared)
inline int read_lucene62_segment_info(const uint8_t* si, size_t len, const uint8_t* expected_id, SegmentInfoEntry* out, std::string* why) {
  const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!si || !out) { *why = "bad arguments"; return ERR_ARG; }
  detail::FileCursor c(si, len);
  int32_t version = 0;
  const uint8_t* id = nullptr;
  std::string suffix;
  int rc = detail::read_index_header(c, "Lucene62SegmentInfo", 0, 1, &version, &id, &suffix, why);
  if (rc) return rc;
  if (!suffix.empty()) { *why = "file mismatch, expected suffix=, got=" + suffix; return ERR_CORRUPT; }
  if (expected_id && std::memcmp(id, expected_id, 16) != 0) { *why = "file mismatch: the .si id differs from the id in the commit point"; return ERR_CORRUPT; }
  std::memcpy(out->id, id, 16);
  for (int i = 0; i < 3; ++i) out->version[i] = (int32_t)c.be32();
  out->max_doc = (int32_t)c.be32();
  const uint8_t compound = c.u8();
  if (!c.ok) { *why = "truncated segment info"; return ERR_EOF; }
  for (int i = 0; i < 3; ++i) if (out->version[i] < 0 || out->version[i] > 255) { *why = "Illegal version in segment info"; return ERR_ARG; }
  if (out->max_doc < 0) { *why = "invalid docCount: " + std::to_string(out->max_doc); return ERR_CORRUPT; }
  out->is_compound_file = compound == 0x01;
  c.string_map();
  c.string_set(&out->n_files);
  c.string_map();
  const uint32_t n_sort = c.vint();
  if (!c.ok) { *why = "truncated segment info"; return ERR_EOF; }
  if ((int32_t)n_sort < 0) { *why = "invalid index sort field count"; return ERR_CORRUPT; }
  out->n_sort_fields = (int32_t)n_sort;
  for (uint32_t i = 0; i < n_sort; ++i) {  // parsed only to reach the footer (segment_infos_format.rs:65-200)
    c.string(nullptr);
    const uint32_t type_id = c.vint();
    int sort_type = (int)type_id;  // 0 String 1 Long 2 Int 3 Double 4 Float
    if (type_id == 5) {
      if (c.u8() > 3) { *why = "invalid index SortedSetSelector ID"; return ERR_CORRUPT; }
      sort_type = 0;
    } else if (type_id == 6) {
      const uint8_t t = c.u8();
      if (t > 3) { *why = "invalid index SortedNumericSortField type ID"; return ERR_CORRUPT; }
      sort_type = 1 + t;
      if (c.u8() > 1) { *why = "invalid index SortedNumericSelector ID"; return ERR_CORRUPT; }
    } else if (type_id > 6) {
      *why = "invalid index sort field type ID";
      return ERR_CORRUPT;
    }
    if (c.u8() > 1) { *why = "invalid index sort reverse"; return ERR_CORRUPT; }
    const uint8_t bv = c.u8();
    if (bv != 0) {
      if (sort_type == 0 || bv != 1) { *why = "invalid missing value flag"; return ERR_CORRUPT; }
      if (sort_type == 1 || sort_type == 3) c.be64(); else c.be32();
    }
    if (!c.ok) { *why = "truncated segment info"; return ERR_EOF; }
  }
  return detail::finish_checksummed_file(si, len, c.pos, "segment info", why);
}

// generation: the N of "segments_N" (base 36 in the file name); < 0 skips the header-suffix comparison
inline int read_segments_file(const uint8_t* data, size_t len, int64_t generation, std::vector<CommitSegmentEntry>* out, std::string* why) {
  const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!data || !out) { *why = "bad arguments"; return ERR_ARG; }
  detail::FileCursor c(data, len);
  if (c.be32() != 0x3FD76C17u) { *why = "invalid magic number"; return ERR_CORRUPT; }
  std::string codec;
  c.string(&codec);
  if (!c.ok || codec != "segments") { *why = "codec mismatch: expected segments"; return ERR_CORRUPT; }
  const int32_t format = (int32_t)c.be32();
  if (format < 4 || format > 6) { *why = "index format either too new or too old: 4 <= " + std::to_string(format) + " <= 6 doesn't hold"; return ERR_CORRUPT; }
  if (c.pos + 16 + 1 > c.len) { *why = "truncated segments file"; return ERR_EOF; }
  c.pos += 16;  // commit id
  const uint8_t slen = c.u8();
  if (c.pos + slen > c.len) { *why = "truncated segments file"; return ERR_EOF; }
  const std::string suffix(reinterpret_cast<const char*>(c.p + c.pos), slen);
  c.pos += slen;
  if (generation >= 0 && suffix != detail::base36((uint64_t)generation)) { *why = "file mismatch, expected suffix=" + detail::base36((uint64_t)generation) + ", got=" + suffix; return ERR_CORRUPT; }
  auto version_triple = [&]() {
    for (int i = 0; i < 3; ++i) { const uint32_t v = c.vint(); if (v > 255) c.ok = false; }
  };
  if (format >= 6) version_triple();
  c.be64();  // version
  c.be32();  // counter
  const int32_t num_segs = (int32_t)c.be32();
  if (!c.ok) { *why = "truncated or invalid segments file header"; return ERR_EOF; }
  if (num_segs < 0 || (size_t)num_segs > len) { *why = "invalid segment count: " + std::to_string(num_segs); return ERR_CORRUPT; }
  if (format >= 6 && num_segs > 0) version_triple();
  out->clear();
  for (int32_t i = 0; i < num_segs; ++i) {
    CommitSegmentEntry s;
    c.string(&s.name);
    const uint8_t has_id = c.u8();
    if (!c.ok || c.pos + 16 > c.len) { *why = "truncated segments file"; return ERR_EOF; }
    if (has_id != 1) { *why = "invalid hasID byte, got: " + std::to_string(has_id); return ERR_CORRUPT; }
    std::memcpy(s.id, c.p + c.pos, 16);
    c.pos += 16;
    c.string(&s.codec);
    if (c.ok && s.codec != "Lucene62") { *why = "Invalid codec name: " + s.codec; return ERR_ARG; }
    s.del_gen = c.be64();
    s.del_count = (int32_t)c.be32();
    s.field_infos_gen = c.be64();
    s.dv_gen = c.be64();
    c.string_set(nullptr);
    const int32_t num_dv = (int32_t)c.be32();
    if (!c.ok || num_dv < 0) { *why = "truncated segments file"; return ERR_EOF; }
    for (int32_t k = 0; k < num_dv && c.ok; ++k) { c.be32(); c.string_set(nullptr); }
    if (!c.ok) { *why = "truncated segments file"; return ERR_EOF; }
    if (s.del_count < 0) { *why = "invalid deletion count: " + std::to_string(s.del_count); return ERR_CORRUPT; }
    out->push_back(std::move(s));
  }
  c.string_map();  // user data
  if (!c.ok) { *why = "truncated segments file"; return ERR_EOF; }
  return detail::finish_checksummed_file(data, len, c.pos, "segments file", why);
}

}  // namespace rucene
