// Host-side reader of Rucene's Lucene50 compound files: ".cfe" (entry table) + ".cfs" (the segment's files, each copied whole —
// index header, body, footer — back to back). Lets a caller outside Rucene reach the .fnm / .doc / .tim / .tip / .nvm / .nvd
// bytes of a segment whose SegmentInfo says is_compound_file. Mirrors (paths relative to /root/reference/src/core):
//   codec/compound.rs:32-37          extensions "cfs" / "cfe", codecs "Lucene50CompoundData" / "Lucene50CompoundEntries", version 0
//   codec/compound.rs:52-101         write: per file of the segment -> copy into .cfs; entry = string strip_segment_name(file),
//                                    i64 start offset, i64 length
//   codec/compound.rs:116-158        Lucene50CompoundReader::new: entries first, then the .cfs header (same version / id), footer,
//                                    and total length == header + sum of entry lengths + footer
//   codec/compound.rs:160-195        read_entries: index header (segment id), vint count, entries, check_footer (CRC verified)
//   codec/segment_infos/mod.rs:64-79 strip_segment_name: "_0_Lucene50_0.doc" -> "_Lucene50_0.doc", "_0.fnm" -> ".fnm"
// Error codes are rgpu_status values (include/rucene_gpu.h). No GPU involved.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "segment_infos_format.hpp"

namespace rucene {

struct CompoundEntry {
  std::string id;  // file name without the segment name: ".fnm", "_Lucene50_0.doc", ...
  int64_t offset = 0, length = 0;
};

// cfs may be null (entries only); when given, its header, footer and total length are checked like the reference's reader does
inline int read_lucene50_compound_entries(const uint8_t* cfe, size_t cfe_len, const uint8_t* cfs, size_t cfs_len, const uint8_t* expected_id,
                                          std::vector<CompoundEntry>* out, std::string* why) {
  const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!cfe || !out) { *why = "bad arguments"; return ERR_ARG; }
  detail::FileCursor c(cfe, cfe_len);
  int32_t version = 0;
  const uint8_t* id = nullptr;
  std::string suffix;
  int rc = detail::read_index_header(c, "Lucene50CompoundEntries", 0, 0, &version, &id, &suffix, why);
  if (rc) return rc;
  if (!suffix.empty()) { *why = "file mismatch, expected suffix=, got=" + suffix; return ERR_CORRUPT; }
  if (expected_id && std::memcmp(id, expected_id, 16) != 0) { *why = "file mismatch: the .cfe id differs from the segment's id"; return ERR_CORRUPT; }
  const uint32_t n = c.vint();
  if (!c.ok || (int32_t)n < 0 || n > cfe_len) { *why = "invalid entry count in compound entries"; return ERR_CORRUPT; }
  out->clear();
  for (uint32_t i = 0; i < n; ++i) {
    CompoundEntry e;
    c.string(&e.id);
    e.offset = c.be64();
    e.length = c.be64();
    if (!c.ok) { *why = "truncated compound entries"; return ERR_EOF; }
    for (const CompoundEntry& o : *out) if (o.id == e.id) { *why = "Duplicate cfs entry id=" + e.id; return ERR_CORRUPT; }
    out->push_back(std::move(e));
  }
  rc = detail::finish_checksummed_file(cfe, cfe_len, c.pos, "compound entries", why);
  if (rc) return rc;
  if (cfs) {
    detail::FileCursor d(cfs, cfs_len);
    int32_t dversion = 0;
    const uint8_t* did = nullptr;
    std::string dsuffix;
    rc = detail::read_index_header(d, "Lucene50CompoundData", version, version, &dversion, &did, &dsuffix, why);
    if (rc) return rc;
    if (!dsuffix.empty() || std::memcmp(did, id, 16) != 0) { *why = ".cfs and .cfe belong to different segments"; return ERR_CORRUPT; }
    uint64_t stored = 0;
    if (cfs_len < 16) { *why = "misplaced codec footer (file truncated?)"; return ERR_CORRUPT; }
    rc = detail::read_footer(cfs, cfs_len, cfs_len - 16, &stored, why);  // retrieve_checksum
    if (rc) return rc;
    uint64_t expected = d.pos + 16;
    for (const CompoundEntry& e : *out) {
      if (e.offset < (int64_t)d.pos || e.length < 0 || (uint64_t)e.offset > cfs_len - 16 || (uint64_t)e.length > cfs_len - 16 - (uint64_t)e.offset) {
        *why = "compound entry outside the data file: " + e.id;
        return ERR_CORRUPT;
      }
      expected += (uint64_t)e.length;
    }
    if (expected != cfs_len) { *why = "length should be " + std::to_string(expected) + " bytes, but is " + std::to_string(cfs_len) + " instead"; return ERR_CORRUPT; }
  }
  return 0;
}

}  // namespace rucene
