// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's Lucene60FieldInfosFormat (".fnm"): the per-segment table of field name -> number,
// index options and flags that the term dictionary and the norms reader are keyed by.
//
// PARITY UNPINNED: the reference holds no test and no golden file for this format; the source text is
// the only authority and each function cites the lines it restates.
//
// Follows (paths relative to /root/reference/src/core):
//   codec/field_infos/field_infos_format.rs:44-53    extension, codec name, format version, flag bits
//   codec/field_infos/field_infos_format.rs:55-112   read_field_infos_from_index
//   codec/field_infos/field_infos_format.rs:114-128  read_field_infos (validate_footer + check_checksum)
//   codec/field_infos/field_infos_format.rs:130-184  index options / doc values type bytes
//   codec/field_infos/field_infos_format.rs:214-259  write
//   codec/field_infos/mod.rs:111-175                 FieldInfo::check_consistency
//   store/io/data_output.rs:97-107, data_input.rs:224-240  map of strings (keys written sorted)
#pragma once
#include <map>
#include <string>
#include <vector>

#include "store.hpp"

namespace orc {

static const char* const FIELD_INFOS_CODEC = "Lucene60FieldInfos";
constexpr uint8_t FI_STORE_TERM_VECTOR = 0x1, FI_OMIT_NORMS = 0x2, FI_STORE_PAYLOADS = 0x4;

struct FieldInfoRec {
  std::string name;
  int32_t number = 0;
  bool store_term_vector = false, omit_norms = false, store_payloads = false;
  int32_t index_options = 0;    // 0 Null .. 4 DocsAndFreqsAndPositionsAndOffsets
  int32_t doc_values_type = 0;  // 0 Null .. 5 SortedNumeric
  int64_t dv_gen = -1;
  std::map<std::string, std::string> attributes;
  int32_t point_dimension_count = 0, point_num_bytes = 0;

  // codec/field_infos/mod.rs:111-175
  void check_consistency() const {
    if (index_options == 0) {
      if (store_term_vector) throw OracleError(E_ILLEGAL_STATE, "non-indexed field cannot store term vectors");
      if (store_payloads) throw OracleError(E_ILLEGAL_STATE, "non-indexed field cannot store payloads");
    } else if (index_options <= 2 && store_payloads) {
      throw OracleError(E_ILLEGAL_STATE, "indexed field cannot have payloads without positions");
    }
    if (point_dimension_count != 0 && point_num_bytes == 0) throw OracleError(E_ILLEGAL_STATE, "pointNumBytes must be > 0");
    if (point_num_bytes != 0 && point_dimension_count == 0) throw OracleError(E_ILLEGAL_STATE, "pointDimensionCount must be > 0");
    if (dv_gen != -1 && doc_values_type == 0) throw OracleError(E_ILLEGAL_STATE, "docvalues update generation without docvalues");
  }
};

// field_infos_format.rs:214-259 (infos.by_number: ascending field number)
inline std::vector<uint8_t> write_field_infos(std::vector<FieldInfoRec> infos, const uint8_t id[ID_LENGTH], const std::string& suffix) {
  std::sort(infos.begin(), infos.end(), [](const FieldInfoRec& a, const FieldInfoRec& b) { return a.number < b.number; });
  ByteOut out;
  write_index_header(out, FIELD_INFOS_CODEC, 0, id, suffix);
  out.write_vint((int32_t)infos.size());
  for (const FieldInfoRec& fi : infos) {
    fi.check_consistency();
    out.write_string(fi.name);
    out.write_vint(fi.number);
    uint8_t bits = 0;
    if (fi.store_term_vector) bits |= FI_STORE_TERM_VECTOR;
    if (fi.omit_norms) bits |= FI_OMIT_NORMS;
    if (fi.store_payloads) bits |= FI_STORE_PAYLOADS;
    out.write_byte(bits);
    out.write_byte((uint8_t)fi.index_options);
    out.write_byte((uint8_t)fi.doc_values_type);
    out.write_long(fi.dv_gen);
    out.write_vint((int32_t)fi.attributes.size());
    for (const auto& kv : fi.attributes) { out.write_string(kv.first); out.write_string(kv.second); }  // std::map: sorted keys
    out.write_vint(fi.point_dimension_count);
    if (fi.point_dimension_count > 0) out.write_vint(fi.point_num_bytes);
  }
  write_footer(out);
  return out.buf;
}

// field_infos_format.rs:55-128
inline std::vector<FieldInfoRec> read_field_infos(const uint8_t* fnm, size_t len) {
  ByteIn in(fnm, (int64_t)len);
  check_index_header(in, FIELD_INFOS_CODEC, 0, 0);
  std::vector<FieldInfoRec> infos;
  const int32_t size = in.read_vint();
  for (int32_t i = 0; i < size; i++) {
    FieldInfoRec fi;
    fi.name = in.read_string();
    fi.number = in.read_vint();
    if (fi.number < 0) throw OracleError(E_CORRUPT_INDEX, "invalid field number for field: " + fi.name);
    const uint8_t bits = in.read_byte();
    fi.store_term_vector = bits & FI_STORE_TERM_VECTOR;
    fi.omit_norms = bits & FI_OMIT_NORMS;
    fi.store_payloads = bits & FI_STORE_PAYLOADS;
    fi.index_options = in.read_byte();
    if (fi.index_options > 4) throw OracleError(E_CORRUPT_INDEX, "invalid IndexOptions byte");
    fi.doc_values_type = in.read_byte();
    if (fi.doc_values_type > 5) throw OracleError(E_CORRUPT_INDEX, "invalid DocValuesType byte");
    fi.dv_gen = in.read_long();
    const int32_t count = in.read_vint();
    if (count < 0) throw OracleError(E_ILLEGAL_STATE, "Invalid StringMap detected");
    for (int32_t k = 0; k < count; k++) {
      std::string key = in.read_string();
      fi.attributes[key] = in.read_string();
    }
    fi.point_dimension_count = in.read_vint();
    fi.point_num_bytes = fi.point_dimension_count != 0 ? in.read_vint() : 0;
    fi.check_consistency();  // FieldInfo::new
    infos.push_back(std::move(fi));
  }
  // validate_footer + check_checksum
  if ((int64_t)len - in.pos != FOOTER_LENGTH) throw OracleError(E_CORRUPT_INDEX, "misplaced codec footer");
  const int64_t stored = retrieve_checksum(fnm, len);
  if ((int64_t)crc32_ieee(fnm, len - 8) != stored) throw OracleError(E_CORRUPT_INDEX, "checksum failed (hardware problems?)");
  // FieldInfos::new: duplicate numbers / names are rejected
  for (size_t a = 0; a < infos.size(); a++)
    for (size_t b = a + 1; b < infos.size(); b++)
      if (infos[a].number == infos[b].number || infos[a].name == infos[b].name)
        throw OracleError(E_ILLEGAL_ARGUMENT, "duplicate field numbers or names");
  return infos;
}

}  // namespace orc
