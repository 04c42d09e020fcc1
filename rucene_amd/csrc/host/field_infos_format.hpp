// Host-side reader of Rucene's Lucene60FieldInfosFormat (".fnm"): field name -> number, index options and flags, i.e.
// what SegmentReadState::field_infos tells BlockTreeTermsReader::new and Lucene53NormsProducer about each field.
// Mirrors (paths relative to /root/reference/src/core):
//   codec/field_infos/field_infos_format.rs:44-53    codec "Lucene60FieldInfos", version 0, STORE_TERM_VECTOR 1 / OMIT_NORMS 2 /
//                                                    STORE_PAYLOADS 4
//   codec/field_infos/field_infos_format.rs:55-128   read: index header, vint count, then per field
//                                                    string name, vint number, u8 bits, u8 index options, u8 doc values type,
//                                                    i64 dv_gen, map of strings, vint point dims [, vint point bytes];
//                                                    validate_footer + check_checksum (CRC verified)
//   codec/field_infos/mod.rs:111-175, 424-500        FieldInfo::check_consistency, FieldInfos::new (no duplicate numbers / names)
// Error codes are rgpu_status values (include/rucene_gpu.h). No GPU involved.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "norms_format.hpp"

namespace rucene {

struct FieldInfoEntry {
  std::string name;
  int32_t number = 0;
  int32_t index_options = 0;   // 0 Null, 1 Docs, 2 DocsAndFreqs, 3 +Positions, 4 +Offsets
  bool store_term_vector = false, omit_norms = false, store_payloads = false;
  int32_t doc_values_type = 0;
};

inline int read_lucene60_field_infos(const uint8_t* fnm, size_t len, std::vector<FieldInfoEntry>* out, std::string* why) {
  const int ERR_STATE = -1, ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4;
  if (!fnm || !out) { *why = "bad arguments"; return ERR_ARG; }
  detail::Cursor c{fnm, len};
  int32_t version = 0;
  const uint8_t* id = nullptr;
  std::string suffix;
  int rc = detail::read_index_header(c, "Lucene60FieldInfos", 0, 0, &version, &id, &suffix, why);
  if (rc) return rc;
  // like the reference, parse to wherever the records end and only then demand the footer exactly there
  auto read_string = [&](std::string* s) {
    const uint32_t n = c.vint();
    if (!c.ok || c.pos + n > c.len) { c.ok = false; return; }
    s->assign(reinterpret_cast<const char*>(c.p + c.pos), n);
    c.pos += n;
  };
  const uint32_t size = c.vint();
  if (!c.ok || size > c.len) { *why = "truncated field infos"; return ERR_EOF; }
  out->clear();
  for (uint32_t i = 0; i < size; ++i) {
    FieldInfoEntry fi;
    read_string(&fi.name);
    const uint32_t number = c.vint();
    const uint8_t bits = c.u8(), opts = c.u8(), dvt = c.u8();
    if (!c.ok || c.pos + 8 > c.len) { *why = "truncated field infos"; return ERR_EOF; }
    if ((int32_t)number < 0) { *why = "invalid field number for field: " + fi.name; return ERR_CORRUPT; }
    if (opts > 4) { *why = "invalid IndexOptions byte: " + std::to_string(opts); return ERR_CORRUPT; }
    if (dvt > 5) { *why = "invalid DocValuesType byte: " + std::to_string(dvt); return ERR_CORRUPT; }
    const int64_t dv_gen = (int64_t)detail::be64_at(c.p + c.pos);
    c.pos += 8;
    const uint32_t n_attr = c.vint();
    if (!c.ok || (int32_t)n_attr < 0) { *why = "Invalid StringMap detected"; return ERR_STATE; }
    for (uint32_t k = 0; k < n_attr && c.ok; ++k) { std::string key, val; read_string(&key); read_string(&val); }
    const uint32_t point_dims = c.vint();
    const uint32_t point_bytes = point_dims != 0 ? c.vint() : 0;
    if (!c.ok) { *why = "truncated field infos"; return ERR_EOF; }
    fi.number = (int32_t)number;
    fi.index_options = opts;
    fi.store_term_vector = bits & 1;
    fi.omit_norms = bits & 2;
    fi.store_payloads = bits & 4;
    fi.doc_values_type = dvt;
    // FieldInfo::check_consistency
    if (opts == 0 && (fi.store_term_vector || fi.store_payloads)) { *why = "non-indexed field '" + fi.name + "' cannot store term vectors / payloads"; return ERR_STATE; }
    if (opts != 0 && opts <= 2 && fi.store_payloads) { *why = "indexed field '" + fi.name + "' cannot have payloads without positions"; return ERR_STATE; }
    if ((point_dims != 0) != (point_bytes != 0)) { *why = "pointDimensionCount and pointNumBytes must both be set or both be 0"; return ERR_STATE; }
    if (dv_gen != -1 && dvt == 0) { *why = "field '" + fi.name + "' cannot have a docvalues update generation without having docvalues"; return ERR_STATE; }
    for (const FieldInfoEntry& o : *out)
      if (o.number == fi.number || o.name == fi.name) { *why = "duplicated field numbers or names: " + fi.name; return ERR_ARG; }
    out->push_back(std::move(fi));
  }
  uint64_t stored = 0;
  rc = detail::read_footer(fnm, len, c.pos, &stored, why);
  if (rc) return rc;
  if ((uint64_t)detail::crc32_ieee(fnm, len - 8) != stored) { *why = "checksum failed (hardware problems?) in field infos"; return ERR_CORRUPT; }
  return 0;
}

}  // namespace rucene
