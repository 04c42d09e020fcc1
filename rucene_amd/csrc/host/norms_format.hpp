// Host-side reader of Rucene's Lucene53 norms files: ".nvm" (metadata) + ".nvd" (data) -> the 1-byte-per-doc norms
// array the scoring kernels take (BM25 reads `norms.get(doc) & 0xFF`, similarity/bm25_similarity.rs:205). Mirrors
// (paths relative to /root/reference/src/core):
//   codec/norms/norms.rs:23-28            codec names "Lucene53NormsData" / "Lucene53NormsMetadata", version 0
//   codec/norms/norms_producer.rs:40-105  Lucene53NormsProducer::new: index headers, entries, check_footer (meta: CRC
//                                         verified), retrieve_checksum (data: footer located only), version match
//   codec/norms/norms_producer.rs:108-140 read_fields: vint field number (-1 ends), u8 bytes_per_value in {0,1,2,4,8},
//                                         i64 offset (bytes_per_value 0: the constant itself)
//   codec/norms/norms_producer.rs:143-189 norms(): big-endian values at offset + doc * bytes_per_value
//   codec/codec_util.rs:46-120, 310-353   IndexHeader / Footer layout, check_footer, retrieve_checksum
// Error codes are rgpu_status values (include/rucene_gpu.h). No GPU involved.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

#include "doc_format.hpp"

namespace rucene {

namespace detail {
inline uint32_t crc32_ieee(const uint8_t* p, size_t n) {  // zlib polynomial, as store/io/fs_index_output.rs (crc32fast)
  struct Table {  // built by a function-local static's initialiser: thread-safe since C++11 (the parsers run from any thread)
    uint32_t v[256];
    Table() {
      for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        v[i] = c;
      }
    }
  };
  static const Table tbl;
  const uint32_t* table = tbl.v;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// check_index_header without a SegmentInfo: magic, codec, version range; returns the 16 id bytes + suffix for the
// caller to compare between the two files. 0 or a negative status.
inline int read_index_header(Cursor& c, const char* codec, int32_t min_version, int32_t max_version, int32_t* version,
                             const uint8_t** id, std::string* suffix, std::string* why) {
  const int ERR_CORRUPT = -4, ERR_EOF = -3;
  if (c.len < 4 + 1 + std::strlen(codec) + 4 + 16 + 1) { *why = "file too short for an index header"; return ERR_EOF; }
  if (c.be32() != 0x3FD76C17u) { *why = "codec header mismatch (bad magic)"; return ERR_CORRUPT; }
  const uint32_t n = c.vint();
  if (!c.ok || n != std::strlen(codec) || c.pos + n > c.len || std::memcmp(c.p + c.pos, codec, n) != 0) {
    *why = std::string("codec mismatch: expected ") + codec;
    return ERR_CORRUPT;
  }
  c.pos += n;
  *version = (int32_t)c.be32();
  if (*version < min_version || *version > max_version) { *why = "version out of range"; return ERR_CORRUPT; }
  if (c.pos + 16 + 1 > c.len) { *why = "truncated index header"; return ERR_EOF; }
  *id = c.p + c.pos;
  c.pos += 16;
  const uint8_t slen = c.u8();
  if (c.pos + slen > c.len) { *why = "truncated index header"; return ERR_EOF; }
  suffix->assign(reinterpret_cast<const char*>(c.p + c.pos), slen);
  c.pos += slen;
  return 0;
}
inline uint64_t be64_at(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
  return v;
}
// validate_footer at `at`; returns the stored CRC through *crc
inline int read_footer(const uint8_t* p, size_t len, size_t at, uint64_t* crc, std::string* why) {
  const int ERR_CORRUPT = -4;
  if (len < 16 || at != len - 16) { *why = "misplaced codec footer (file truncated?)"; return ERR_CORRUPT; }
  Cursor f{p, len};
  f.pos = at;
  if (f.be32() != ~0x3FD76C17u) { *why = "codec footer mismatch"; return ERR_CORRUPT; }
  if (f.be32() != 0) { *why = "codec footer mismatch: unknown algorithm id"; return ERR_CORRUPT; }
  *crc = be64_at(p + at + 8);
  if (*crc & 0xFFFFFFFF00000000ull) { *why = "Illegal CRC-32 checksum"; return ERR_CORRUPT; }
  return 0;
}
}  // namespace detail

// norms_out[doc] = norms(field).get(doc) & 0xFF for doc in [0, max_doc). Returns 0 or a negative rgpu_status.
inline int read_lucene53_norms(const uint8_t* nvm, size_t nvm_len, const uint8_t* nvd, size_t nvd_len, int32_t field_number,
                               int32_t max_doc, uint8_t* norms_out, std::string* why) {
  const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!nvm || !nvd || !norms_out || max_doc < 0 || field_number < 0) { *why = "bad arguments"; return ERR_ARG; }
  detail::Cursor m{nvm, nvm_len};
  int32_t meta_version = 0, data_version = 0;
  const uint8_t *meta_id = nullptr, *data_id = nullptr;
  std::string meta_suffix, data_suffix;
  int rc = detail::read_index_header(m, "Lucene53NormsMetadata", 0, 0, &meta_version, &meta_id, &meta_suffix, why);
  if (rc) return rc;
  bool have = false;
  uint8_t bpv = 0;
  uint64_t offset = 0;
  while (true) {  // read_fields
    const uint32_t field = m.vint();
    if (!m.ok) { *why = "truncated norms metadata"; return ERR_EOF; }
    if (field == 0xFFFFFFFFu) break;  // write_vint(-1)
    const uint8_t b = m.u8();
    if (!(b == 0 || b == 1 || b == 2 || b == 4 || b == 8)) { *why = "Invalid bytes_per_value in norms metadata"; return ERR_CORRUPT; }
    if (m.pos + 8 > m.len) { *why = "truncated norms metadata"; return ERR_EOF; }
    const uint64_t off = detail::be64_at(m.p + m.pos);
    m.pos += 8;
    if ((int32_t)field == field_number) { have = true; bpv = b; offset = off; }  // HashMap::insert: the last entry wins
  }
  uint64_t stored = 0;
  rc = detail::read_footer(nvm, nvm_len, m.pos, &stored, why);
  if (rc) return rc;
  if ((uint64_t)detail::crc32_ieee(nvm, nvm_len - 8) != stored) { *why = "checksum failed (hardware problems?) in norms metadata"; return ERR_CORRUPT; }
  detail::Cursor d{nvd, nvd_len};
  rc = detail::read_index_header(d, "Lucene53NormsData", 0, 0, &data_version, &data_id, &data_suffix, why);
  if (rc) return rc;
  if (data_version != meta_version) { *why = "Format versions mismatch between .nvm and .nvd"; return ERR_CORRUPT; }
  if (std::memcmp(meta_id, data_id, 16) != 0 || meta_suffix != data_suffix) { *why = ".nvm and .nvd belong to different segments"; return ERR_CORRUPT; }
  if (nvd_len < 16) { *why = "misplaced codec footer (file truncated?)"; return ERR_CORRUPT; }
  rc = detail::read_footer(nvd, nvd_len, nvd_len - 16, &stored, why);  // retrieve_checksum: located, not verified
  if (rc) return rc;
  if (!have) { *why = "the field has no norms in this segment"; return ERR_ARG; }
  if (bpv == 0) {  // ScalarNumericDocValue
    std::memset(norms_out, (int)(offset & 0xFF), (size_t)max_doc);
    return 0;
  }
  const uint64_t need = (uint64_t)max_doc * bpv;
  if (offset > nvd_len - 16 || need > nvd_len - 16 - offset) { *why = "norms slice outside the data file"; return ERR_EOF; }
  const uint8_t* base = nvd + offset + (bpv - 1);  // big-endian: the low byte comes last
  for (int32_t doc = 0; doc < max_doc; ++doc) norms_out[doc] = base[(size_t)doc * bpv];
  return 0;
}

}  // namespace rucene

namespace rucene {

// Lucene50LiveDocsFormat::read_live_docs (codec/live_docs.rs:81-121): ".liv" -> FixedBitSet words (bit doc&63 of word
// doc>>6 set = live, util/bit_set.rs:453-460) as rgpu_segment_upload takes them. Checks the index header (codec
// "Lucene50LiveDocs", version 0; the suffix is base36(del_gen), which only the caller's SegmentCommitInfo could
// confirm), the footer checksum, clear ghost bits (FixedBitSet::copy_from, bit_set.rs:155-171) and — when
// del_count >= 0 — `max_doc - cardinality == del_count`.
inline int read_lucene50_live_docs(const uint8_t* liv, size_t liv_len, int32_t max_doc, int32_t del_count, uint64_t* words_out,
                                   std::string* why) {
  const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!liv || !words_out || max_doc <= 0) { *why = "bad arguments"; return ERR_ARG; }
  detail::Cursor c{liv, liv_len};
  int32_t version = 0;
  const uint8_t* id = nullptr;
  std::string suffix;
  int rc = detail::read_index_header(c, "Lucene50LiveDocs", 0, 0, &version, &id, &suffix, why);
  if (rc) return rc;
  const size_t num_words = (size_t)(((max_doc - 1) >> 6) + 1);  // bits2words, bit_set.rs:480-484
  if (c.pos + 8 * num_words + 16 > liv_len) { *why = "live docs file shorter than max_doc needs"; return ERR_EOF; }
  int64_t live = 0;
  for (size_t w = 0; w < num_words; ++w) {
    const uint64_t v = detail::be64_at(liv + c.pos + 8 * w);
    words_out[w] = v;
    live += __builtin_popcountll(v);
  }
  c.pos += 8 * num_words;
  uint64_t stored = 0;
  rc = detail::read_footer(liv, liv_len, c.pos, &stored, why);
  if (rc) return rc;
  if ((uint64_t)detail::crc32_ieee(liv, liv_len - 8) != stored) { *why = "checksum failed (hardware problems?) in live docs"; return ERR_CORRUPT; }
  if ((max_doc & 63) != 0 && (words_out[num_words - 1] >> (max_doc & 63)) != 0) { *why = "ghost bits set past max_doc"; return ERR_CORRUPT; }
  if (del_count >= 0 && (int64_t)max_doc - live != (int64_t)del_count) {
    *why = "bits.deleted= " + std::to_string((int64_t)max_doc - live) + " info.delcount= " + std::to_string(del_count);
    return ERR_CORRUPT;
  }
  return 0;
}

}  // namespace rucene
