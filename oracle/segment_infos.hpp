// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's per-segment ".si" file (Lucene62SegmentInfoFormat) and of the commit point "segments_N"
// (SegmentInfos::write_output / read_commit_generation): what names a directory's segments, their sizes, ids, codec
// and deletion generations.
//
// PARITY UNPINNED: the reference holds no test and no golden file for these formats; the source text is
// the only authority and each function cites the lines it restates.
//
// Follows (paths relative to /root/reference/src/core):
//   codec/segment_infos/segment_infos_format.rs:37-41     extension, codec name, versions
//   codec/segment_infos/segment_infos_format.rs:43-222    read_segment_info_from_index (incl. the index-sort grammar)
//   codec/segment_infos/segment_infos_format.rs:228-247   read (validate_footer + check_checksum)
//   codec/segment_infos/segment_infos_format.rs:249-380   write
//   codec/segment_infos/mod.rs:154-155                    SEGMENT_USE_COMPOUND_YES / _NO
//   codec/segment_infos/segment_infos.rs:39-47            segments_N format versions
//   codec/segment_infos/segment_infos.rs:243-300          write_output
//   codec/segment_infos/segment_infos.rs:443-569          read_commit / read_commit_generation
//   codec/codec_util.rs:148-221                           check_header_no_magic / check_index_header_suffix
//   util/numeric.rs:148-160                               to_base36 (oracle/norms.hpp)
//   util/version.rs:45, 125-160                           VERSION_LATEST = 6.4.18, Version::new range checks
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "norms.hpp"
#include "store.hpp"

namespace orc {

static const char* const SEGMENT_INFO_CODEC = "Lucene62SegmentInfo";
constexpr int32_t SI_VERSION_START = 0, SI_VERSION_CURRENT = 1;
constexpr uint8_t SEGMENT_USE_COMPOUND_YES = 0x01, SEGMENT_USE_COMPOUND_NO = 0xff;
constexpr int32_t SEGMENTS_VERSION_50 = 4, SEGMENTS_VERSION_53 = 6, SEGMENTS_VERSION_CURRENT = 6;
constexpr int32_t LATEST_MAJOR = 6, LATEST_MINOR = 4, LATEST_BUGFIX = 18;

struct VersionRec {
  int32_t major = LATEST_MAJOR, minor = LATEST_MINOR, bugfix = LATEST_BUGFIX;
  void check() const {  // Version::new
    if (major > 255 || major < 0 || minor > 255 || minor < 0 || bugfix > 255 || bugfix < 0)
      throw OracleError(E_ILLEGAL_ARGUMENT, "Illegal version");
  }
  bool operator<(const VersionRec& o) const {
    return std::tie(major, minor, bugfix) < std::tie(o.major, o.minor, o.bugfix);
  }
};

inline void write_string_map(ByteOut& out, const std::map<std::string, std::string>& m) {
  out.write_vint((int32_t)m.size());
  for (const auto& kv : m) { out.write_string(kv.first); out.write_string(kv.second); }
}
inline std::map<std::string, std::string> read_string_map(ByteIn& in) {
  const int32_t n = in.read_vint();
  if (n < 0) throw OracleError(E_ILLEGAL_STATE, "Invalid StringMap detected");
  std::map<std::string, std::string> m;
  for (int32_t i = 0; i < n; i++) { std::string k = in.read_string(); m[k] = in.read_string(); }
  return m;
}
inline void write_string_set(ByteOut& out, const std::set<std::string>& s) {
  out.write_vint((int32_t)s.size());
  for (const auto& v : s) out.write_string(v);
}
inline std::set<std::string> read_string_set(ByteIn& in) {
  const int32_t n = in.read_vint();
  if (n < 0) throw OracleError(E_ILLEGAL_STATE, "Invalid StringSet detected");
  std::set<std::string> s;
  for (int32_t i = 0; i < n; i++) s.insert(in.read_string());
  return s;
}
inline void check_whole_file_checksum(const uint8_t* data, size_t len, int64_t body_end) {  // validate_footer + check_checksum
  if ((int64_t)len - body_end != FOOTER_LENGTH) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer");
  const int64_t stored = retrieve_checksum(data, len);
  if ((int64_t)crc32_ieee(data, len - 8) != stored) throw OracleError(E_CORRUPT_INDEX, "checksum failed (hardware problems?)");
}

struct SegmentInfoRec {
  std::string name;  // "_0", "_1", ...
  uint8_t id[ID_LENGTH] = {0};
  VersionRec version;
  int32_t max_doc = 0;
  bool is_compound_file = false;
  std::map<std::string, std::string> diagnostics, attributes;
  std::set<std::string> files;
  int32_t num_sort_fields = 0;  // an index sort is parsed (to reach the footer) but not kept
};

// segment_infos_format.rs:249-380 (no index sort: vint 0)
inline std::vector<uint8_t> write_segment_info(const SegmentInfoRec& si) {
  ByteOut out;
  write_index_header(out, SEGMENT_INFO_CODEC, SI_VERSION_CURRENT, si.id, "");
  if (si.version.major < 5) throw OracleError(E_ILLEGAL_ARGUMENT, "invalid major version: should be >= 5");
  out.write_int(si.version.major);
  out.write_int(si.version.minor);
  out.write_int(si.version.bugfix);
  out.write_int(si.max_doc);
  out.write_byte(si.is_compound_file ? SEGMENT_USE_COMPOUND_YES : SEGMENT_USE_COMPOUND_NO);
  write_string_map(out, si.diagnostics);
  for (const std::string& f : si.files) {  // parse_segment_name(file) == name
    const std::string stem = f.substr(0, f.find_first_of("._", 1));
    if (stem != si.name) throw OracleError(E_ILLEGAL_ARGUMENT, "invalid files: expected segment=" + si.name + ", got=" + f);
  }
  write_string_set(out, si.files);
  write_string_map(out, si.attributes);
  out.write_vint(0);
  write_footer(out);
  return out.buf;
}

// segment_infos_format.rs:43-247. expected_id == nullptr skips the id comparison (check_index_header_id).
inline SegmentInfoRec read_segment_info(const uint8_t* data, size_t len, const std::string& segment, const uint8_t* expected_id) {
  ByteIn in(data, (int64_t)len);
  SegmentInfoRec si;
  si.name = segment;
  {
    ByteIn h(data, (int64_t)len);
    check_index_header(h, SEGMENT_INFO_CODEC, SI_VERSION_START, SI_VERSION_CURRENT);
    // id sits 16 + 1 bytes before the end of the header (empty suffix)
    std::memcpy(si.id, data + h.pos - 1 - ID_LENGTH, ID_LENGTH);
    if (data[h.pos - 1] != 0) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected suffix=, got another");
    if (expected_id && std::memcmp(si.id, expected_id, ID_LENGTH) != 0) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected id differs");
    in.pos = h.pos;
  }
  si.version.major = in.read_int();
  si.version.minor = in.read_int();
  si.version.bugfix = in.read_int();
  si.version.check();
  si.max_doc = in.read_int();
  if (si.max_doc < 0) throw OracleError(E_CORRUPT_INDEX, "invalid docCount");
  si.is_compound_file = in.read_byte() == SEGMENT_USE_COMPOUND_YES;
  si.diagnostics = read_string_map(in);
  si.files = read_string_set(in);
  si.attributes = read_string_map(in);
  si.num_sort_fields = in.read_vint();
  if (si.num_sort_fields < 0) throw OracleError(E_CORRUPT_INDEX, "invalid index sort field count");
  for (int32_t i = 0; i < si.num_sort_fields; i++) {
    in.read_string();
    const int32_t type_id = in.read_vint();
    int sort_type = type_id;  // 0 String 1 Long 2 Int 3 Double 4 Float
    if (type_id == 5) {
      if (in.read_byte() > 3) throw OracleError(E_CORRUPT_INDEX, "invalid index SortedSetSelector ID");
      sort_type = 0;
    } else if (type_id == 6) {
      const uint8_t t = in.read_byte();
      if (t > 3) throw OracleError(E_CORRUPT_INDEX, "invalid index SortedNumericSortField type ID");
      sort_type = 1 + t;
      if (in.read_byte() > 1) throw OracleError(E_CORRUPT_INDEX, "invalid index SortedNumericSelector ID");
    } else if (type_id < 0 || type_id > 6) {
      throw OracleError(E_CORRUPT_INDEX, "invalid index sort field type ID");
    }
    if (in.read_byte() > 1) throw OracleError(E_CORRUPT_INDEX, "invalid index sort reverse");
    const uint8_t bv = in.read_byte();
    if (bv != 0) {
      if (sort_type == 0) throw OracleError(E_CORRUPT_INDEX, "missing value on a string sort");  // unreachable!() in the reference
      if (bv != 1) throw OracleError(E_CORRUPT_INDEX, "invalid missing value flag");
      if (sort_type == 1 || sort_type == 3) in.read_long(); else in.read_int();
    }
  }
  check_whole_file_checksum(data, len, in.pos);
  return si;
}

struct CommitSegmentRec {
  std::string name;
  uint8_t id[ID_LENGTH] = {0};
  std::string codec = "Lucene62";
  VersionRec version;  // the segment's own version (feeds min_seg_ver)
  int32_t max_doc = 0;  // for the del_count check (comes from the .si file)
  int64_t del_gen = -1;
  int32_t del_count = 0;
  int64_t field_infos_gen = -1, dv_gen = -1;
  std::set<std::string> field_infos_files;
  std::map<int32_t, std::set<std::string>> dv_update_files;
};
struct CommitRec {
  int64_t generation = 1;
  uint8_t id[ID_LENGTH] = {0};
  VersionRec lucene_version;
  int64_t version = 0;
  int32_t counter = 0;
  std::vector<CommitSegmentRec> segments;
};

// segment_infos.rs:243-300 (the header id is random in the reference; here the caller's)
inline std::vector<uint8_t> write_segments_file(const CommitRec& c) {
  ByteOut out;
  write_index_header(out, "segments", SEGMENTS_VERSION_CURRENT, c.id, to_base36((uint64_t)c.generation));
  out.write_vint(LATEST_MAJOR);
  out.write_vint(LATEST_MINOR);
  out.write_vint(LATEST_BUGFIX);
  out.write_long(c.version);
  out.write_int(c.counter);
  out.write_int((int32_t)c.segments.size());
  if (!c.segments.empty()) {
    VersionRec min_version;
    for (const CommitSegmentRec& s : c.segments) if (s.version < min_version) min_version = s.version;
    out.write_vint(min_version.major);
    out.write_vint(min_version.minor);
    out.write_vint(min_version.bugfix);
  }
  for (const CommitSegmentRec& s : c.segments) {
    out.write_string(s.name);
    out.write_byte(1);
    out.write_bytes(s.id, ID_LENGTH);
    out.write_string(s.codec);
    out.write_long(s.del_gen);
    if (s.del_count < 0 || s.del_count > s.max_doc) throw OracleError(E_ILLEGAL_STATE, "cannot write segment: invalid del_count");
    out.write_int(s.del_count);
    out.write_long(s.field_infos_gen);
    out.write_long(s.dv_gen);
    write_string_set(out, s.field_infos_files);
    out.write_int((int32_t)s.dv_update_files.size());
    for (const auto& kv : s.dv_update_files) { out.write_int(kv.first); write_string_set(out, kv.second); }
  }
  write_string_map(out, {});
  write_footer(out);
  return out.buf;
}

// segment_infos.rs:443-569. `max_docs` (one per segment, from the .si files the reference reads in the same loop) bound
// del_count; pass nullptr to skip that check.
inline CommitRec read_segments_file(const uint8_t* data, size_t len, int64_t generation, const int32_t* max_docs, size_t n_max_docs) {
  ByteIn in(data, (int64_t)len);
  CommitRec c;
  c.generation = generation;
  if (in.read_int() != CODEC_MAGIC) throw OracleError(E_CORRUPT_INDEX, "invalid magic number");
  if (in.read_string() != "segments") throw OracleError(E_CORRUPT_INDEX, "codec mismatch");
  const int32_t format = in.read_int();
  if (format < SEGMENTS_VERSION_50 || format > SEGMENTS_VERSION_CURRENT) throw OracleError(E_CORRUPT_INDEX, "index format either too new or too old");
  in.read_exact(c.id, ID_LENGTH);
  {
    const uint8_t slen = in.read_byte();
    std::string suffix((const char*)in.get_and_advance(slen), slen);
    if (suffix != to_base36((uint64_t)generation)) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected suffix=" + to_base36((uint64_t)generation) + ", got=" + suffix);
  }
  if (format >= SEGMENTS_VERSION_53) {
    c.lucene_version.major = in.read_vint(); c.lucene_version.minor = in.read_vint(); c.lucene_version.bugfix = in.read_vint();
    c.lucene_version.check();
  }
  c.version = in.read_long();
  c.counter = in.read_int();
  const int32_t num_segs = in.read_int();
  if (num_segs < 0) throw OracleError(E_CORRUPT_INDEX, "invalid segment count");
  if (format >= SEGMENTS_VERSION_53 && num_segs > 0) {
    VersionRec v;
    v.major = in.read_vint(); v.minor = in.read_vint(); v.bugfix = in.read_vint();
    v.check();
  }
  for (int32_t i = 0; i < num_segs; i++) {
    CommitSegmentRec s;
    s.name = in.read_string();
    if (in.read_byte() != 1) throw OracleError(E_CORRUPT_INDEX, "invalid hasID byte");
    in.read_exact(s.id, ID_LENGTH);
    s.codec = in.read_string();
    if (s.codec != "Lucene62") throw OracleError(E_ILLEGAL_ARGUMENT, "Invalid codec name: " + s.codec);
    s.del_gen = in.read_long();
    s.del_count = in.read_int();
    if (max_docs && (size_t)i < n_max_docs) s.max_doc = max_docs[i];
    if (s.del_count < 0 || (max_docs && (size_t)i < n_max_docs && s.del_count > s.max_doc))
      throw OracleError(E_CORRUPT_INDEX, "invalid deletion count");
    s.field_infos_gen = in.read_long();
    s.dv_gen = in.read_long();
    s.field_infos_files = read_string_set(in);
    const int32_t num_dv = in.read_int();
    for (int32_t k = 0; k < num_dv; k++) { const int32_t field = in.read_int(); s.dv_update_files[field] = read_string_set(in); }
    c.segments.push_back(std::move(s));
  }
  read_string_map(in);  // user data
  check_whole_file_checksum(data, len, in.pos);
  return c;
}

}  // namespace orc
