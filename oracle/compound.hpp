// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's Lucene50CompoundFormat: ".cfs" (the segment's files copied back to back) + ".cfe" (entry table).
//
// PARITY UNPINNED: no reference test or golden file exists for this format; each function cites the lines it
// restates (paths relative to /root/reference/src/core):
//   codec/compound.rs:32-37          extensions, codec names, version
//   codec/compound.rs:52-101         write (verify_and_copy_index_header, body copy, footer re-emitted from the verified checksum)
//   codec/compound.rs:116-195        Lucene50CompoundReader::new / read_entries
//   codec/segment_infos/mod.rs:64-79 strip_segment_name
#pragma once
#include <map>
#include <string>
#include <vector>

#include "segment_infos.hpp"
#include "store.hpp"

namespace orc {

static const char* const COMPOUND_DATA_CODEC = "Lucene50CompoundData";
static const char* const COMPOUND_ENTRY_CODEC = "Lucene50CompoundEntries";

inline std::string strip_segment_name(const std::string& name) {  // segment_infos/mod.rs:64-79
  const size_t u = name.find('_', 1);
  if (u != std::string::npos) return name.substr(u);
  const size_t d = name.find('.', 1);
  return d != std::string::npos ? name.substr(d) : name;
}

// files: name -> whole file bytes, in si.files() order (a sorted set). Returns {cfs, cfe}.
inline std::pair<std::vector<uint8_t>, std::vector<uint8_t>> write_compound(const std::map<std::string, std::vector<uint8_t>>& files,
                                                                             const uint8_t id[ID_LENGTH]) {
  ByteOut data, entries;
  write_index_header(data, COMPOUND_DATA_CODEC, 0, id, "");
  write_index_header(entries, COMPOUND_ENTRY_CODEC, 0, id, "");
  entries.write_vint((int32_t)files.size());
  for (const auto& kv : files) {
    const std::vector<uint8_t>& f = kv.second;
    const int64_t start_offset = data.file_pointer();
    // verify_and_copy_index_header: the sub-file must carry an index header with this segment's id
    ByteIn in(f.data(), (int64_t)f.size());
    if (in.read_int() != CODEC_MAGIC) throw OracleError(E_CORRUPT_INDEX, "codec header mismatch in " + kv.first);
    in.read_string();
    in.read_int();
    if (std::memcmp(in.get_and_advance(ID_LENGTH), id, ID_LENGTH) != 0) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected id differs: " + kv.first);
    if (f.size() < (size_t)FOOTER_LENGTH) throw OracleError(E_CORRUPT_INDEX, "file too short: " + kv.first);
    // check_footer: the copied file's checksum is verified, then the footer is written anew from it
    const int64_t stored = retrieve_checksum(f.data(), f.size());
    if ((int64_t)crc32_ieee(f.data(), f.size() - 8) != stored) throw OracleError(E_CORRUPT_INDEX, "checksum failed: " + kv.first);
    data.write_bytes(f.data(), f.size() - FOOTER_LENGTH);
    data.write_int(FOOTER_MAGIC);
    data.write_int(0);
    data.write_long(stored);
    entries.write_string(strip_segment_name(kv.first));
    entries.write_long(start_offset);
    entries.write_long(data.file_pointer() - start_offset);
  }
  write_footer(data);
  write_footer(entries);
  return {data.buf, entries.buf};
}

struct CompoundEntryRec { int64_t offset, length; };

inline std::map<std::string, CompoundEntryRec> read_compound(const uint8_t* cfe, size_t cfe_len, const uint8_t* cfs, size_t cfs_len,
                                                             const uint8_t* expected_id) {
  ByteIn e(cfe, (int64_t)cfe_len);
  const int32_t version = check_index_header(e, COMPOUND_ENTRY_CODEC, 0, 0);
  const uint8_t* cfe_id = cfe + e.pos - 1 - ID_LENGTH;  // the id sits right before the (empty) suffix
  if (cfe[e.pos - 1] != 0) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected an empty suffix");
  if (expected_id && std::memcmp(cfe_id, expected_id, ID_LENGTH) != 0) throw OracleError(E_CORRUPT_INDEX, "file mismatch, expected id differs");
  const int32_t n = e.read_vint();
  std::map<std::string, CompoundEntryRec> m;
  for (int32_t i = 0; i < n; i++) {
    std::string id = e.read_string();
    CompoundEntryRec r;
    r.offset = e.read_long();
    r.length = e.read_long();
    if (!m.emplace(id, r).second) throw OracleError(E_CORRUPT_INDEX, "Duplicate cfs entry id=" + id);
  }
  check_whole_file_checksum(cfe, cfe_len, e.pos);
  ByteIn d(cfs, (int64_t)cfs_len);
  check_index_header(d, COMPOUND_DATA_CODEC, version, version);
  if (cfs[d.pos - 1] != 0 || std::memcmp(cfs + d.pos - 1 - ID_LENGTH, cfe_id, ID_LENGTH) != 0)
    throw OracleError(E_CORRUPT_INDEX, "file mismatch, .cfs and .cfe ids differ");
  retrieve_checksum(cfs, cfs_len);
  uint64_t expected = (uint64_t)d.pos + FOOTER_LENGTH;
  for (const auto& kv : m) expected += (uint64_t)kv.second.length;
  if (expected != cfs_len) throw OracleError(E_CORRUPT_INDEX, "length should be " + std::to_string(expected) + " bytes, but is " + std::to_string(cfs_len) + " instead");
  return m;
}

}  // namespace orc
