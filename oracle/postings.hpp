// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's Lucene50 postings ".doc" codec: ForUtil block framing, multi-level skip
// list writer/reader, postings writer, and the docs+freqs BlockDocIterator.
//
// PARITY UNPINNED for: .doc block framing (ForUtil::read_block/write_block/skip_block), the skip list
// (skip_writer.rs / skip_reader.rs), read_vint_block, finish_term/decode_term and BlockDocIterator — the
// reference has no test that exercises them; the source text is the only authority and
// each function below cites the lines it restates. Pinned pieces (BP128 layout, legacy bit order,
// MAX_DATA_SIZE) live in packed.hpp.
//
// Follows (paths relative to /root/reference/src/core):
//   codec/postings/for_util.rs:120-185   ForUtilInstance::with_input / with_output
//   codec/postings/for_util.rs:187-243   read_block
//   codec/postings/for_util.rs:263-272   skip_block
//   codec/postings/for_util.rs:374-478   is_all_equal / bits_required / write_block
//   codec/postings/skip_writer.rs:80-289 Lucene50SkipWriter
//   codec/postings/skip_reader.rs:222-584 Lucene50SkipReader
//   codec/postings/posting_writer.rs:289-361,457-591 start_term/start_doc/finish_doc/finish_term
//   codec/postings/posting_reader.rs:85-110   open (header, version, use_simd)
//   codec/postings/posting_reader.rs:308-333  read_vint_block
//   codec/postings/posting_reader.rs:460-561, 612-794 BlockDocIterator reset/refill_docs/next/advance
//   codec/postings/blocktree/mod.rs:33-59     BlockTermState
//   util/math.rs:21-32                        log(x, base)
#pragma once
#include <memory>
#include <vector>

#include "elias_fano.hpp"
#include "packed.hpp"
#include "store.hpp"

namespace orc {

constexpr int32_t NO_MORE_DOCS = INT32_MAX;  // search/mod.rs:59
constexpr int MAX_SKIP_LEVELS = 10;          // posting_reader.rs:49
constexpr int SKIP_MULTIPLIER = 8;           // skip_reader.rs:235
constexpr int32_t VERSION_START = 0, VERSION_CURRENT = 1;  // posting_reader.rs:57-58
static const char* const DOC_CODEC = "Lucene50PostingsWriterDoc";  // posting_reader.rs:52
constexpr int PACKED_VERSION_CURRENT = 2;    // packed_misc.rs:47-65 (VERSION_MONOTONIC_WITHOUT_ZIGZAG)
constexpr uint16_t FLAG_FREQS = 1 << 3;      // posting_iterator.rs:18-49

// util/math.rs:21-32
inline int ilog(int64_t x, int base) {
  int ret = 0;
  while (x >= base) { x /= base; ret++; }
  return ret;
}

// blocktree/mod.rs:33-59 (fields used by the docs+freqs path)
// Field order is the C layout shared with include/rucene_gpu.h's rgpu_term_state (32 bytes) so that one
// numpy record array can feed both the oracle and the C ABI.
struct BlockTermState {
  int64_t doc_start_fp = 0;
  int64_t skip_offset = -1;
  int64_t total_term_freq = 0;
  int32_t doc_freq = 0;
  int32_t singleton_doc_id = -1;
};
static_assert(sizeof(BlockTermState) == 32, "BlockTermState must match rgpu_term_state");

// posting_writer.rs:33-56
struct EfWriterMeta {
  int32_t ef_base_doc = -1, ef_upper_doc = 0;
  bool use_ef = false, with_pf = true;
  void reset() { ef_base_doc = -1; ef_upper_doc = 0; }
};

// ---- ForUtil ---------------------------------------------------------------------------------------------------

struct ForUtil {
  int encoded_sizes[32];
  int formats[32];
  int bpvs[32];
  int iterations[32];

  static int iterations_for(int format, int bpv) {
    if (format == FMT_PACKED) return compute_iterations(BulkOperationPacked(bpv).byte_value_count);
    return compute_iterations(BulkOperationPackedSingleBlock(bpv).byte_value_count());
  }

  // for_util.rs:150-185
  static ForUtil with_output(float acceptable_overhead_ratio, ByteOut& out) {
    ForUtil f;
    out.write_vint(PACKED_VERSION_CURRENT);
    for (int bpv = 1; bpv < 33; bpv++) {
      FormatAndBits fb = format_fastest(BLOCK_SIZE, bpv, acceptable_overhead_ratio);
      f.formats[bpv - 1] = fb.format;
      f.bpvs[bpv - 1] = fb.bits_per_value;
      f.encoded_sizes[bpv - 1] = (int)format_byte_count(fb.format, BLOCK_SIZE, fb.bits_per_value);
      f.iterations[bpv - 1] = iterations_for(fb.format, fb.bits_per_value);
      out.write_vint(fb.format << 5 | (fb.bits_per_value - 1));
    }
    return f;
  }

  // for_util.rs:120-148
  static ForUtil with_input(ByteIn& in) {
    ForUtil f;
    int32_t packed_ints_version = in.read_vint();
    if (packed_ints_version < 0 || packed_ints_version > PACKED_VERSION_CURRENT)
      throw OracleError(E_CORRUPT_INDEX, "bad PackedInts version");  // check_version
    for (int bpv = 0; bpv < 32; bpv++) {
      int32_t code = in.read_vint();
      int format_id = (int)((uint32_t)code >> 5);
      int bits_per_value = (code & 31) + 1;
      if (format_id != 0 && format_id != 1) throw OracleError(E_CORRUPT_INDEX, "Invalid format id");
      f.formats[bpv] = format_id;
      f.bpvs[bpv] = bits_per_value;
      f.encoded_sizes[bpv] = (int)format_byte_count(format_id, BLOCK_SIZE, bits_per_value);
      f.iterations[bpv] = iterations_for(format_id, bits_per_value);
    }
    return f;
  }

  // for_util.rs:187-243. `etype` non-null == the doc-delta stream (encode type honoured);
  // returns the encode type (0 = PF). For non-PF types nothing is decoded (caller must handle).
  int read_block(ByteIn& in, int32_t* decoded, bool is_doc_stream, bool by_simd) const {
    uint8_t code = in.read_byte();
    if (is_doc_stream) {
      int etype = code >> 6;  // for_util.rs:505-513
      if (etype != 0) return etype;
    }
    int num_bits = code & 0x3F;
    if (num_bits == 0) {  // ALL_VALUES_EQUAL
      int32_t value = in.read_vint();
      for (int i = 0; i < BLOCK_SIZE; i++) decoded[i] = value;
      return 0;
    }
    if (num_bits > 32) throw OracleError(E_CORRUPT_INDEX, "num_bits > 32");
    int encoded_size = encoded_sizes[num_bits - 1];
    if (by_simd) {
      const uint8_t* enc = in.get_and_advance((size_t)(num_bits * BLOCK_SIZE / 8));  // SIMD_ENCODE_SIZE
      Simd128Packer::unpack(enc, (uint32_t*)decoded, num_bits);
    } else {
      uint8_t encoded[MAX_ENCODED_SIZE + 64];
      in.read_exact(encoded, (size_t)encoded_size);
      int iters = iterations[num_bits - 1];
      if (formats[num_bits - 1] == FMT_PACKED)
        BulkOperationPacked(bpvs[num_bits - 1]).decode_byte_to_int(encoded, decoded, iters);
      else
        BulkOperationPackedSingleBlock(bpvs[num_bits - 1]).decode_byte_to_int(encoded, decoded, iters);
    }
    return 0;
  }

  // for_util.rs:263-272
  void skip_block(ByteIn& in) const {
    int num_bits = in.read_byte();
    if (num_bits == 0) { in.read_vint(); return; }
    if (num_bits > 32) throw OracleError(E_CORRUPT_INDEX, "num_bits > 32");
    in.seek(in.file_pointer() + encoded_sizes[num_bits - 1]);
  }

  // for_util.rs:374-394
  static bool is_all_equal(const int32_t* data) {
    for (int i = 1; i < BLOCK_SIZE; i++) if (data[i] != data[0]) return false;
    return true;
  }
  static int bits_required(const int32_t* data) {
    int32_t o = 0;
    for (int i = 0; i < BLOCK_SIZE; i++) o |= data[i];
    return o == 0 ? 0 : 32 - __builtin_clz((uint32_t)o);
  }

  // for_util.rs:396-478. `meta` non-null == the doc-delta stream of a postings writer (posting_writer.rs:334-351); its
  // use_ef is false in every Rucene build (posting_writer.rs:46, never set) — the EF / BITSET arms below exist so that the
  // READ side (read_other_encode_block, the iterator's EF / BITSET arms) has files to be checked against.
  void write_block(const int32_t* data, ByteOut& out, bool by_simd, EfWriterMeta* meta = nullptr) const {
    if (is_all_equal(data)) {
      out.write_byte(0);
      out.write_vint(data[0]);
      return;
    }
    int num_bits = bits_required(data);
    if (!(num_bits > 0 && num_bits <= 32)) throw OracleError(E_ILLEGAL_STATE, "bad num_bits");
    const int encoded_size = encoded_sizes[num_bits - 1];
    if (meta != nullptr && meta->use_ef) {  // for_util.rs:417-468
      EliasFanoEncoder ef_encoder(BLOCK_SIZE, (int64_t)(meta->ef_upper_doc - meta->ef_base_doc - 1));
      if (ef_encoder.encode_size() <= MAX_ENCODED_SIZE) {
        int32_t doc_id = meta->ef_base_doc;
        if (doc_id < 0) doc_id = 0;
        int32_t min_doc = INT32_MAX, max_doc = 0;
        for (int i = 0; i < BLOCK_SIZE; i++) {
          doc_id += data[i];
          if (doc_id > max_doc) max_doc = doc_id;
          if (doc_id < min_doc) min_doc = doc_id;
          ef_encoder.encode_next((int64_t)(doc_id - meta->ef_base_doc - 1));
        }
        // FixedBitSet::resize(max_doc - min_doc + 1): num_words = bits2words (util/bit_set.rs:193-199); encode_size = words * 8
        const int num_words = (max_doc - min_doc + 1 + 63) >> 6;
        if (num_words * 8 <= encoded_size) {
          std::vector<int64_t> bits((size_t)num_words, 0);
          doc_id = meta->ef_base_doc;
          if (doc_id < 0) doc_id = 0;
          for (int i = 0; i < BLOCK_SIZE; i++) {
            doc_id += data[i];
            const int b = doc_id - min_doc;
            bits[(size_t)(b >> 6)] |= (int64_t)(1ull << (b & 63));
          }
          out.write_byte(0x80);  // EncodeType::BITSET << 6
          out.write_vint(min_doc);
          out.write_byte((uint8_t)num_words);
          EliasFanoEncoder::write_data(bits, out);
          return;
        }
        if (!meta->with_pf || ef_encoder.encode_size() <= encoded_size) { ef_encoder.serialize(out); return; }
      }
    }
    uint8_t encoded[MAX_ENCODED_SIZE + 64] = {0};
    out.write_byte((uint8_t)num_bits);
    if (by_simd) {
      Simd128Packer::pack((const uint32_t*)data, encoded, num_bits);
      out.write_bytes(encoded, (size_t)(num_bits * BLOCK_SIZE / 8));
    } else {
      int iters = iterations[num_bits - 1];
      int32_t padded[MAX_DATA_SIZE + 8] = {0};  // writer buffers are MAX_DATA_SIZE long (posting_writer.rs:216-217)
      std::memcpy(padded, data, sizeof(int32_t) * BLOCK_SIZE);
      if (formats[num_bits - 1] == FMT_PACKED)
        BulkOperationPacked(bpvs[num_bits - 1]).encode_int_to_byte(padded, encoded, iters);
      else
        BulkOperationPackedSingleBlock(bpvs[num_bits - 1]).encode_int_to_byte(padded, encoded, iters);
      out.write_bytes(encoded, (size_t)encoded_sizes[num_bits - 1]);
    }
  }
};

// ---- Lucene50SkipWriter (docs+freqs fields only: no positions/payloads/offsets) -------------------------------

struct SkipWriter {
  std::vector<int32_t> last_skip_doc;
  std::vector<int64_t> last_skip_doc_pointer;
  int32_t cur_doc = 0;
  int64_t cur_doc_pointer = 0;
  int number_of_skip_levels;
  uint32_t skip_interval, skip_multiplier;
  std::vector<ByteOut> skip_buffer;
  bool initialized = false;
  int64_t last_doc_fp = 0;

  // skip_writer.rs:81-127 (doc_count = segment max_doc)
  SkipWriter(int max_skip_levels, uint32_t block_size, uint32_t doc_count)
      : last_skip_doc(max_skip_levels, 0), last_skip_doc_pointer(max_skip_levels, 0),
        skip_interval(block_size), skip_multiplier(SKIP_MULTIPLIER) {
    int n = (doc_count <= block_size) ? 1 : 1 + ilog((int64_t)doc_count / (int64_t)block_size, SKIP_MULTIPLIER);
    number_of_skip_levels = std::min(max_skip_levels, n);
  }
  // skip_writer.rs:140-149
  void reset_skip(int64_t doc_fp) { last_doc_fp = doc_fp; initialized = false; }
  virtual ~SkipWriter() {}
  // skip_writer.rs:151-183 (virtual: positions.hpp adds the position pointers)
  virtual void init_skip() {
    if (!initialized) {
      if (skip_buffer.empty()) skip_buffer.resize((size_t)number_of_skip_levels);
      else for (auto& b : skip_buffer) b.reset();
      std::fill(last_skip_doc.begin(), last_skip_doc.end(), 0);
      std::fill(last_skip_doc_pointer.begin(), last_skip_doc_pointer.end(), last_doc_fp);
      initialized = true;
    }
  }
  // skip_writer.rs:187-205
  void buffer_skip(int32_t doc, uint32_t num_docs, int64_t doc_out_pointer) {
    init_skip();
    cur_doc = doc;
    cur_doc_pointer = doc_out_pointer;
    buffer_skip_levels(num_docs);
  }
  // skip_writer.rs:209-238
  void buffer_skip_levels(uint32_t df) {
    int num_levels = 1;
    df /= skip_interval;
    while (true) {
      if (df % skip_multiplier != 0 || num_levels >= number_of_skip_levels) break;
      num_levels++;
      df /= skip_multiplier;
    }
    int64_t child_pointer = 0;
    for (int i = 0; i < num_levels; i++) {
      write_skip_data_local(i);
      int64_t new_child_pointer = skip_buffer[(size_t)i].file_pointer();
      if (i != 0) skip_buffer[(size_t)i].write_vlong(child_pointer);
      child_pointer = new_child_pointer;
    }
  }
  // skip_writer.rs:261-289 (the positions arm lives in positions.hpp's subclass)
  virtual void write_skip_data_local(int level) {
    int32_t delta = cur_doc - last_skip_doc[(size_t)level];
    skip_buffer[(size_t)level].write_vint(delta);
    last_skip_doc[(size_t)level] = cur_doc;
    skip_buffer[(size_t)level].write_vlong(cur_doc_pointer - last_skip_doc_pointer[(size_t)level]);
    last_skip_doc_pointer[(size_t)level] = cur_doc_pointer;
  }
  // skip_writer.rs:241-258
  int64_t write_skip(ByteOut& output) const {
    int64_t skip_pointer = output.file_pointer();
    if (skip_buffer.empty()) return skip_pointer;
    for (int i = 1; i < number_of_skip_levels; i++) {
      int level = number_of_skip_levels - i;
      int64_t length = skip_buffer[(size_t)level].file_pointer();
      if (length > 0) {
        output.write_vlong(length);
        skip_buffer[(size_t)level].write_to(output);
      }
    }
    skip_buffer[0].write_to(output);
    return skip_pointer;
  }
};

// ---- Lucene50PostingsWriter (docs, or docs+freqs) --------------------------------------------------------------

struct PostingsWriter {
  ByteOut doc_out;
  int64_t doc_start_fp = 0;
  std::vector<int32_t> doc_delta_buffer, freq_buffer;
  int doc_buffer_upto = 0;
  int32_t last_block_doc_id = 0;
  int32_t last_doc_id = 0;
  int32_t doc_count = 0;
  ForUtil for_util;
  SkipWriter skip_writer;
  bool write_freqs;
  bool use_simd;
  int32_t version;
  EfWriterMeta ef_writer_meta;  // use_ef stays false unless a test switches it on (no Rucene build does)

  // posting_writer.rs:116-251. `version` 1 + simd == the live Zhihu layout; 0 == legacy Lucene PackedInts.
  PostingsWriter(int32_t max_doc, int32_t version_, bool write_freqs_, const uint8_t segment_id[ID_LENGTH],
                 const std::string& suffix)
      : doc_delta_buffer(MAX_DATA_SIZE, 0), freq_buffer(MAX_DATA_SIZE, 0),
        skip_writer(MAX_SKIP_LEVELS, BLOCK_SIZE, (uint32_t)max_doc), write_freqs(write_freqs_), version(version_) {
    write_index_header(doc_out, DOC_CODEC, version, segment_id, suffix);
    for_util = ForUtil::with_output(0.0f /*COMPACT*/, doc_out);
    use_simd = version > VERSION_START;  // posting_writer.rs:201-205 (SSE3 assumed present)
  }

  // posting_writer.rs:289-302
  void start_term() {
    doc_start_fp = doc_out.file_pointer();
    last_doc_id = 0;
    last_block_doc_id = -1;
    skip_writer.reset_skip(doc_start_fp);
    ef_writer_meta.reset();  // posting_writer.rs:300
  }
  // posting_writer.rs:304-361
  void start_doc(int32_t doc_id, int32_t term_doc_freq) {
    if (last_block_doc_id != -1 && doc_buffer_upto == 0)
      skip_writer.buffer_skip(last_block_doc_id, (uint32_t)doc_count, doc_out.file_pointer());
    int32_t doc_delta = doc_id - last_doc_id;
    if (doc_id < 0 || (doc_count > 0 && doc_delta <= 0)) throw OracleError(E_CORRUPT_INDEX, "docs out of order");
    doc_delta_buffer[(size_t)doc_buffer_upto] = doc_delta;
    if (write_freqs) freq_buffer[(size_t)doc_buffer_upto] = term_doc_freq;
    doc_buffer_upto++;
    doc_count++;
    if (doc_buffer_upto == BLOCK_SIZE) {
      ef_writer_meta.ef_upper_doc = doc_id;  // posting_writer.rs:335
      for_util.write_block(doc_delta_buffer.data(), doc_out, use_simd, &ef_writer_meta);
      if (write_freqs) for_util.write_block(freq_buffer.data(), doc_out, use_simd);
    }
    last_doc_id = doc_id;
  }
  // posting_writer.rs:457-474
  void finish_doc() {
    if (doc_buffer_upto == BLOCK_SIZE) {
      last_block_doc_id = last_doc_id;
      doc_buffer_upto = 0;
      ef_writer_meta.ef_base_doc = last_block_doc_id;  // posting_writer.rs:472
    }
  }
  // posting_writer.rs:477-591
  void finish_term(BlockTermState& state) {
    if (!(state.doc_freq > 0) || state.doc_freq != doc_count) throw OracleError(E_ILLEGAL_STATE, "doc_freq mismatch");
    int32_t singleton_doc_id;
    if (state.doc_freq == 1) {
      singleton_doc_id = doc_delta_buffer[0];
    } else {
      for (int i = 0; i < doc_buffer_upto; i++) {
        int32_t doc_delta = doc_delta_buffer[(size_t)i];
        int32_t freq = freq_buffer[(size_t)i];
        if (!write_freqs) doc_out.write_vint(doc_delta);
        else if (freq == 1) doc_out.write_vint(doc_delta << 1 | 1);
        else { doc_out.write_vint(doc_delta << 1); doc_out.write_vint(freq); }
      }
      singleton_doc_id = -1;
    }
    int64_t skip_offset = (doc_count > BLOCK_SIZE) ? skip_writer.write_skip(doc_out) - doc_start_fp : -1;
    state.doc_start_fp = doc_start_fp;
    state.singleton_doc_id = singleton_doc_id;
    state.skip_offset = skip_offset;
    doc_buffer_upto = 0;
    last_doc_id = 0;
    doc_count = 0;
  }
  // posting_writer.rs:610-619
  void close() { write_footer(doc_out); }
};

// ---- Lucene50SkipReader (docs+freqs fields only) ---------------------------------------------------------------

struct SkipReader {
  int max_number_of_skip_levels;
  int number_of_skip_levels = 0;
  int number_of_levels_to_buffer = 1;
  int32_t doc_count = 0;
  std::vector<ByteIn> skip_stream;
  std::vector<bool> stream_present;
  std::vector<int64_t> skip_pointer, skip_interval, num_skipped, child_pointer, doc_pointer;
  std::vector<int32_t> skip_doc;
  int32_t last_doc = 0;
  int64_t last_child_pointer = 0;
  int64_t last_doc_pointer = 0;

  // skip_reader.rs:222-299
  SkipReader(const ByteIn& stream, int max_skip_levels)
      : max_number_of_skip_levels(max_skip_levels), skip_stream((size_t)max_skip_levels),
        stream_present((size_t)max_skip_levels, false), skip_pointer((size_t)max_skip_levels, 0),
        skip_interval(), num_skipped((size_t)max_skip_levels, 0), child_pointer((size_t)max_skip_levels, 0),
        doc_pointer((size_t)max_skip_levels, 0), skip_doc((size_t)max_skip_levels, 0) {
    skip_stream[0] = stream;
    stream_present[0] = true;
    skip_interval.push_back(BLOCK_SIZE);
    for (int i = 1; i < max_skip_levels; i++) skip_interval.push_back(skip_interval[(size_t)i - 1] * SKIP_MULTIPLIER);
  }
  // skip_reader.rs:307-313
  static int32_t trim(int32_t df) { return (df % BLOCK_SIZE == 0) ? df - 1 : df; }
  // skip_reader.rs:315-356
  void init(int64_t skip_ptr, int64_t doc_base_pointer, int32_t df) {
    df = trim(df);
    skip_pointer[0] = skip_ptr;
    doc_count = df;
    std::fill(skip_doc.begin(), skip_doc.end(), 0);
    std::fill(num_skipped.begin(), num_skipped.end(), 0);
    std::fill(child_pointer.begin(), child_pointer.end(), 0);
    for (int i = 1; i < number_of_skip_levels; i++) stream_present[(size_t)i] = false;
    load_skip_levels();
    last_doc_pointer = doc_base_pointer;
    std::fill(doc_pointer.begin(), doc_pointer.end(), doc_base_pointer);
  }
  int64_t get_doc_pointer() const { return last_doc_pointer; }  // skip_reader.rs:360-362
  int32_t next_skip_doc() const { return skip_doc[0]; }         // skip_reader.rs:380-382
  int32_t doc() const { return last_doc; }                      // skip_reader.rs:548-550
  virtual ~SkipReader() {}
  // skip_reader.rs:385-408 (virtual: positions.hpp adds the position pointers)
  virtual void seek_child(int level) {
    size_t ul = (size_t)level;
    skip_stream[ul].seek(last_child_pointer);
    num_skipped[ul] = num_skipped[ul + 1] - skip_interval[ul + 1];
    skip_doc[ul] = last_doc;
    if (level > 0) child_pointer[ul] = skip_stream[ul].read_vlong() + skip_pointer[ul - 1];
    doc_pointer[ul] = last_doc_pointer;
  }
  // skip_reader.rs:410-429
  virtual void set_last_skip_data(int level) {
    last_doc = skip_doc[(size_t)level];
    last_child_pointer = child_pointer[(size_t)level];
    last_doc_pointer = doc_pointer[(size_t)level];
  }
  // skip_reader.rs:431-453
  virtual int32_t read_skip_data(int level) {
    int32_t delta = skip_stream[(size_t)level].read_vint();
    int64_t pointer = skip_stream[(size_t)level].read_vlong();
    doc_pointer[(size_t)level] += pointer;
    return delta;
  }
  // skip_reader.rs:460-511
  void load_skip_levels() {
    if ((int64_t)doc_count <= skip_interval[0]) number_of_skip_levels = 1;
    else number_of_skip_levels = 1 + ilog((int64_t)doc_count / skip_interval[0], SKIP_MULTIPLIER);
    if (number_of_skip_levels > max_number_of_skip_levels) number_of_skip_levels = max_number_of_skip_levels;
    skip_stream[0].seek(skip_pointer[0]);
    int to_buffer = number_of_levels_to_buffer;
    for (int i = number_of_skip_levels - 1; i >= 1; i--) {
      int64_t length = skip_stream[0].read_vlong();
      skip_pointer[(size_t)i] = skip_stream[0].file_pointer();
      // Both arms (SkipBuffer copy vs clone) leave level i positioned at its start and the base stream
      // just past it; an in-memory ByteIn clone models either.
      skip_stream[(size_t)i] = skip_stream[0];
      stream_present[(size_t)i] = true;
      if (to_buffer > 0) to_buffer--;
      skip_stream[0].seek(skip_stream[0].file_pointer() + length);
    }
    skip_pointer[0] = skip_stream[0].file_pointer();
  }
  // skip_reader.rs:513-539
  bool load_next_skip(int level) {
    size_t ul = (size_t)level;
    set_last_skip_data(level);
    num_skipped[ul] += skip_interval[ul];
    if (num_skipped[ul] > (int64_t)doc_count) {
      skip_doc[ul] = INT32_MAX;
      if (number_of_skip_levels > level) number_of_skip_levels = level;
      return false;
    }
    skip_doc[ul] += read_skip_data(level);
    if (level != 0) child_pointer[ul] = skip_stream[ul].read_vlong() + skip_pointer[ul - 1];
    return true;
  }
  // skip_reader.rs:554-584
  int32_t skip_to(int32_t target) {
    int level = 0;
    while (level < number_of_skip_levels - 1 && target > skip_doc[(size_t)(level + 1)]) level++;
    while (level >= 0) {
      if (target > skip_doc[(size_t)level]) {
        if (!load_next_skip(level)) continue;
      } else {
        if (level > 0 && last_child_pointer > skip_stream[(size_t)(level - 1)].file_pointer()) seek_child(level - 1);
        level--;
      }
    }
    return (int32_t)(num_skipped[0] - skip_interval[0] - 1);
  }
};

// ---- Lucene50PostingsReader::open + BlockDocIterator ----------------------------------------------------------

// One opened ".doc" file (posting_reader.rs:85-158): header check, version, ForUtil table.
struct PostingsReader {
  const uint8_t* data;
  int64_t len;
  int32_t version;
  bool use_simd;
  ForUtil for_util;
  PostingsReader(const uint8_t* d, int64_t l) : data(d), len(l) {
    ByteIn in(d, l);
    version = check_index_header(in, DOC_CODEC, VERSION_START, VERSION_CURRENT);
    use_simd = version > VERSION_START;  // posting_reader.rs:103-107
    for_util = ForUtil::with_input(in);
    // retrieve_checksum (codec_util.rs): footer magic + algorithm id must be sane
    if (l < 16) throw OracleError(E_CORRUPT_INDEX, "file too short for footer");
    ByteIn f(d, l, l - 16);
    if (f.read_int() != FOOTER_MAGIC) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch");
    if (f.read_int() != 0) throw OracleError(E_CORRUPT_INDEX, "codec footer mismatch: unknown algorithm");
  }
};

// posting_reader.rs:308-333
inline void read_vint_block(ByteIn& in, int32_t* doc_buffer, int32_t* freq_buffer, int num, bool index_has_freq) {
  if (index_has_freq) {
    for (int i = 0; i < num; i++) {
      uint32_t code = (uint32_t)in.read_vint();
      doc_buffer[i] = (int32_t)(code >> 1);
      if ((code & 1) != 0) freq_buffer[i] = 1;
      else freq_buffer[i] = in.read_vint();
    }
  } else {
    for (int i = 0; i < num; i++) doc_buffer[i] = in.read_vint();
  }
}

struct BlockDocIterator {
  int32_t doc_delta_buffer[MAX_DATA_SIZE + 8];
  int32_t freq_buffer[MAX_DATA_SIZE + 8];
  int32_t doc_buffer_upto = 0;
  std::unique_ptr<SkipReader> skipper;
  bool skipped = false;
  ByteIn doc_in;
  bool index_has_freq;
  int32_t doc_freq = 0;
  int64_t total_term_freq = 0;
  int32_t doc_upto = 0;
  int32_t doc = 0;
  int32_t accum = 0;
  int32_t freq_ = 0;
  int64_t doc_term_start_fp = 0;
  int64_t skip_offset = 0;
  int32_t next_skip_doc = 0;
  bool needs_freq = false;
  int32_t singleton_doc_id = 0;
  const PostingsReader* reader;
  uint64_t blocks_decoded = 0;  // instrumentation for the CPU baseline report (not in the reference)
  // EF / BITSET blocks (posting_reader.rs:392-400): the block's encode type and what its arm of next() walks
  int encode_type = 0;  // 0 PF, 1 EF, 2 BITSET
  int32_t ef_base_doc = -1, ef_base_total = 0, bits_min_doc = 0, bits_index = 0;
  std::unique_ptr<EliasFanoEncoder> ef_encoder;
  std::unique_ptr<EliasFanoDecoder> ef_decoder;
  std::vector<int64_t> doc_bits;

  // posting_reader.rs:410-458
  BlockDocIterator(const PostingsReader* r, bool index_has_freq_, const BlockTermState& st, uint16_t flags)
      : doc_in(r->data, r->len), index_has_freq(index_has_freq_), reader(r) {
    reset(st, flags);
  }
  // posting_reader.rs:460-499
  void reset(const BlockTermState& st, uint16_t flags) {
    doc_freq = st.doc_freq;
    total_term_freq = index_has_freq ? st.total_term_freq : (int64_t)doc_freq;
    doc_term_start_fp = st.doc_start_fp;
    skip_offset = st.skip_offset;
    singleton_doc_id = st.singleton_doc_id;
    if (doc_freq > 1) doc_in.seek(doc_term_start_fp);
    doc = -1;
    needs_freq = (flags & FLAG_FREQS) != 0;
    if (!index_has_freq || !needs_freq) for (int i = 0; i < MAX_DATA_SIZE; i++) freq_buffer[i] = 1;
    accum = 0;
    doc_upto = 0;
    next_skip_doc = BLOCK_SIZE - 1;
    doc_buffer_upto = BLOCK_SIZE;
    skipped = false;
    ef_base_doc = -1;  // posting_reader.rs:493-494
    ef_base_total = 0;
    encode_type = 0;
  }
  // posting_reader.rs:501-561
  void refill_docs() {
    if (accum > 0) ef_base_doc = accum;  // :503-505 "EF & PF compatible"
    ef_base_total = doc_upto;
    encode_type = 0;
    bits_index = 0;
    int32_t left = doc_freq - doc_upto;
    if (left >= BLOCK_SIZE) {
      int etype = reader->for_util.read_block(doc_in, doc_delta_buffer, true, reader->use_simd);
      if (etype == 3) throw OracleError(E_UNSUPPORTED, "EncodeType::FULL is unimplemented in the reference (posting_reader.rs:639-641)");
      encode_type = etype;
      if (etype == 1) {  // ForUtil::read_other_encode_block, EF arm (for_util.rs:345-362)
        const int64_t upper_bound = doc_in.read_vlong();
        ef_encoder.reset(new EliasFanoEncoder(BLOCK_SIZE, upper_bound));
        ef_encoder->deserialize2(doc_in);
        ef_decoder.reset(new EliasFanoDecoder(ef_encoder.get()));
      } else if (etype == 2) {  // BITSET arm (for_util.rs:363-368)
        bits_min_doc = doc_in.read_vint();
        const int num_longs = doc_in.read_byte();
        doc_bits.assign((size_t)num_longs, 0);
        EliasFanoEncoder::read_data2(doc_bits, doc_in);
      }
      if (index_has_freq) {
        if (needs_freq) reader->for_util.read_block(doc_in, freq_buffer, false, reader->use_simd);
        else reader->for_util.skip_block(doc_in);
      }
      blocks_decoded++;
    } else if (doc_freq == 1) {
      doc_delta_buffer[0] = singleton_doc_id;
      freq_buffer[0] = (int32_t)total_term_freq;
    } else {
      read_vint_block(doc_in, doc_delta_buffer, freq_buffer, left, index_has_freq);
    }
    doc_buffer_upto = 0;
  }
  int32_t doc_id() const { return doc; }
  int32_t freq() const { return freq_; }
  size_t cost() const { return (size_t)doc_freq; }  // posting_reader.rs:791-793
  // posting_reader.rs:612-647
  int32_t next() {
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    if (encode_type == 0) {
      doc = accum + doc_delta_buffer[doc_buffer_upto];
    } else if (encode_type == 1) {  // posting_reader.rs:624-626
      doc = (int32_t)ef_decoder->next_value() + 1 + ef_base_doc;
    } else {                        // :628-633 FixedBitSet::next_set_bit (util/bit_set.rs:351-376)
      int32_t i = bits_index >> 6;
      uint64_t word = (uint64_t)doc_bits[(size_t)i] >> (bits_index & 63);
      int32_t found;
      if (word != 0) found = bits_index + __builtin_ctzll(word);
      else {
        found = NO_MORE_DOCS;
        while (++i < (int32_t)doc_bits.size()) if (doc_bits[(size_t)i] != 0) { found = (i << 6) + __builtin_ctzll((uint64_t)doc_bits[(size_t)i]); break; }
      }
      bits_index = found;
      doc = bits_min_doc + bits_index;
      bits_index += 1;
    }
    accum = doc;
    doc_upto++;
    freq_ = freq_buffer[doc_buffer_upto];
    doc_buffer_upto++;
    return doc;
  }
  // posting_reader.rs:649-789 (PF arm)
  int32_t advance(int32_t target) {
    if (target == NO_MORE_DOCS) { doc = NO_MORE_DOCS; return doc; }
    if (doc_freq > BLOCK_SIZE && target > next_skip_doc) {
      if (!skipper) skipper.reset(new SkipReader(doc_in, MAX_SKIP_LEVELS));
      if (!skipped) {
        skipper->init(doc_term_start_fp + skip_offset, doc_term_start_fp, doc_freq);
        skipped = true;
      }
      int32_t new_doc_upto = skipper->skip_to(target) + 1;
      if (new_doc_upto > doc_upto) {
        doc_upto = new_doc_upto;
        doc_buffer_upto = BLOCK_SIZE;
        accum = skipper->doc();
        doc_in.seek(skipper->get_doc_pointer());
      }
      next_skip_doc = skipper->next_skip_doc();
    }
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    if (encode_type != 0)  // posting_reader.rs:733-777 (EliasFanoDecoder::advance_to_value, count_ones_before_index2): not restated
      throw OracleError(E_UNSUPPORTED, "advance() inside an EF / BITSET block is not restated (no Rucene build writes such blocks)");
    while (true) {
      accum += doc_delta_buffer[doc_buffer_upto];
      doc_upto++;
      if (accum >= target) break;
      doc_buffer_upto++;
      if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    }
    doc = accum;
    freq_ = freq_buffer[doc_buffer_upto];
    doc_buffer_upto++;
    return doc;
  }
};

}  // namespace orc
