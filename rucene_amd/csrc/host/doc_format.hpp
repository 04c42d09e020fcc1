// Host-side parse of a ".doc" file's framing: the open() half of Lucene50PostingsReader, without any
// posting decode (that happens on the GPU). Mirrors (paths relative to /root/reference/src/core):
//   codec/postings/posting_reader.rs:85-110   open: check_index_header(DOC_CODEC, VERSION_START..=VERSION_CURRENT),
//                                             use_simd = version > VERSION_START, ForUtil::with_input, retrieve_checksum
//   codec/codec_util.rs:46-120                IndexHeader / Footer layout
//   codec/postings/for_util.rs:120-148        ForUtilInstance::with_input (PackedInts version + 32 format codes)
// Error codes are rgpu_status values (include/rucene_gpu.h).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace rucene {

struct DocFileInfo {
  int32_t version = 0;           // 0 = legacy PackedInts blocks, 1 = SIMD-BP128 blocks
  int64_t postings_start = 0;    // first byte after the ForUtil table
  int64_t postings_end = 0;      // first byte of the footer
  int32_t encoded_sizes[32] = {0};
};

namespace detail {
struct Cursor {
  const uint8_t* p;
  size_t len, pos = 0;
  bool ok = true;
  uint8_t u8() { if (pos >= len) { ok = false; return 0; } return p[pos++]; }
  uint32_t be32() { uint32_t v = 0; for (int i = 0; i < 4; ++i) v = (v << 8) | u8(); return v; }
  uint32_t vint() {
    uint32_t v = 0;
    for (int i = 0; i < 5; ++i) { uint8_t b = u8(); v |= (uint32_t)(b & 0x7f) << (7 * i); if (!(b & 0x80)) return v; }
    ok = false;
    return v;
  }
};
}  // namespace detail

// returns 0 or a negative rgpu_status; *why explains a failure
inline int parse_doc_file(const uint8_t* data, size_t len, DocFileInfo* out, std::string* why) {
  const int ERR_CORRUPT = -4, ERR_EOF = -3, ERR_UNSUPPORTED = -5;
  detail::Cursor c{data, len};
  if (len < 16 + 4) { *why = "file too short"; return ERR_EOF; }
  if (c.be32() != 0x3FD76C17u) { *why = "codec header mismatch (bad magic)"; return ERR_CORRUPT; }
  uint32_t n = c.vint();
  static const char kCodec[] = "Lucene50PostingsWriterDoc";
  if (!c.ok || n != sizeof(kCodec) - 1 || c.pos + n > len || std::memcmp(data + c.pos, kCodec, n) != 0) {
    *why = "codec mismatch: expected Lucene50PostingsWriterDoc";
    return ERR_CORRUPT;
  }
  c.pos += n;
  int32_t version = (int32_t)c.be32();
  if (version < 0 || version > 1) { *why = "unsupported .doc version " + std::to_string(version); return ERR_CORRUPT; }
  c.pos += 16;  // segment id
  uint8_t slen = c.u8();
  c.pos += slen;
  if (!c.ok || c.pos >= len) { *why = "truncated index header"; return ERR_EOF; }
  uint32_t packed_version = c.vint();
  if (!c.ok || packed_version > 2) { *why = "bad PackedInts version"; return ERR_CORRUPT; }
  for (int bpv = 0; bpv < 32; ++bpv) {
    uint32_t code = c.vint();
    if (!c.ok) { *why = "truncated ForUtil table"; return ERR_EOF; }
    uint32_t format_id = code >> 5, bits = (code & 31) + 1;
    if (format_id > 1) { *why = "Invalid format id in ForUtil table"; return ERR_CORRUPT; }
    int64_t sz = format_id == 0 ? ((int64_t)128 * bits + 7) / 8 : (int64_t)((128 + (64 / bits) - 1) / (64 / bits)) * 8;
    out->encoded_sizes[bpv] = (int32_t)sz;
    // The kernels assume what Rucene's writer always produces (for_util.rs:161-177 with COMPACT): slot
    // bpv-1 stores exactly bpv bits, Packed except PackedSingleBlock for 1/2/4, i.e. 16*bpv bytes.
    const uint32_t want_fmt = (bpv + 1 == 1 || bpv + 1 == 2 || bpv + 1 == 4) ? 1u : 0u;
    if ((int)bits != bpv + 1 || sz != 16 * (bpv + 1) || (version == 0 && format_id != want_fmt)) {
      *why = "ForUtil table differs from the layout Rucene writes (acceptable_overhead_ratio != COMPACT?)";
      return ERR_UNSUPPORTED;
    }
  }
  out->version = version;
  out->postings_start = (int64_t)c.pos;
  // footer: i32 ~magic | i32 algorithm id (0) | i64 crc
  detail::Cursor f{data, len};
  f.pos = len - 16;
  if (f.be32() != ~0x3FD76C17u) { *why = "codec footer mismatch"; return ERR_CORRUPT; }
  if (f.be32() != 0) { *why = "codec footer mismatch: unknown algorithm id"; return ERR_CORRUPT; }
  out->postings_end = (int64_t)len - 16;
  if (out->postings_start > out->postings_end) { *why = "header overlaps footer"; return ERR_CORRUPT; }
  return 0;
}

// The ".pos" file of a positions field, or the ".pay" file of one that stores payloads / offsets (posting_reader.rs:112-158:
// opened next to .doc with the same version; header "Lucene50PostingsWriterPos" / "...Pay", same segment id and suffix as the
// .doc file, footer magic). No ForUtil table: the .doc file's applies. `doc_file` is the segment's already validated .doc image.
inline int parse_side_file(const uint8_t* data, size_t len, const char* codec, const char* ext, const uint8_t* doc_file, int32_t doc_version,
                           int64_t* postings_start, std::string* why) {
  const int ERR_CORRUPT = -4, ERR_EOF = -3;
  static const char kDoc[] = "Lucene50PostingsWriterDoc";
  const size_t codec_len = std::strlen(codec);
  const std::string x(ext);
  if (len < 16 + 4 + 1 + codec_len + 4 + 16 + 1) { *why = x + " file too short"; return ERR_EOF; }
  detail::Cursor c{data, len};
  if (c.be32() != 0x3FD76C17u) { *why = x + ": codec header mismatch (bad magic)"; return ERR_CORRUPT; }
  const uint32_t n = c.vint();
  if (!c.ok || n != codec_len || std::memcmp(data + c.pos, codec, n) != 0) { *why = std::string("codec mismatch: expected ") + codec; return ERR_CORRUPT; }
  c.pos += n;
  if ((int32_t)c.be32() != doc_version) { *why = x + " version differs from the .doc file's"; return ERR_CORRUPT; }
  // segment id + suffix must be the .doc file's (check_index_header with the segment's id / suffix)
  const size_t doc_id_at = 4 + 1 + (sizeof(kDoc) - 1) + 4;
  const size_t id_len = 16 + 1 + (size_t)doc_file[doc_id_at + 16];
  if (c.pos + id_len > len - 16 || std::memcmp(data + c.pos, doc_file + doc_id_at, id_len) != 0) { *why = x + " belongs to another segment (id / suffix mismatch)"; return ERR_CORRUPT; }
  c.pos += id_len;
  detail::Cursor f{data, len};
  f.pos = len - 16;
  if (f.be32() != ~0x3FD76C17u || f.be32() != 0) { *why = x + ": codec footer mismatch"; return ERR_CORRUPT; }
  *postings_start = (int64_t)c.pos;
  return 0;
}
inline int parse_pos_file(const uint8_t* data, size_t len, const uint8_t* doc_file, int32_t doc_version, int64_t* postings_start, std::string* why) {
  return parse_side_file(data, len, "Lucene50PostingsWriterPos", ".pos", doc_file, doc_version, postings_start, why);
}
inline int parse_pay_file(const uint8_t* data, size_t len, const uint8_t* doc_file, int32_t doc_version, int64_t* postings_start, std::string* why) {
  return parse_side_file(data, len, "Lucene50PostingsWriterPay", ".pay", doc_file, doc_version, postings_start, why);
}

}  // namespace rucene
