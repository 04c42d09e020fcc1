// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of the POSITIONS side of Rucene's Lucene50 postings: the ".pos" file, the position pointers inside the
// ".doc" skip entries, BlockPostingIterator (docs + freqs + positions) and — for fields that store payloads or offsets —
// the ".pay" file, the payload / offset words of the skip entries and of the trailing VInt position block, and
// EverythingIterator (positions + payloads + offsets). SURVEY.md §8(f)3.
//
// PARITY UNPINNED: the reference holds no test for any of this; the source text is the only authority and
// each function cites the lines it restates. The tests (tests/test_positions.py) check the writer/reader pair against
// brute force over the input postings, through every skip level and across block boundaries.
//
// Follows (paths relative to /root/reference/src/core/codec/postings):
//   posting_writer.rs:116-251, 595-619   new (.pos header "Lucene50PostingsWriterPos"), init, close
//   posting_writer.rs:289-302            start_term (pos_start_fp; reset_skip with the position pointer)
//   posting_writer.rs:304-361            start_doc (buffer_skip with last_block_pos_fp / last_block_pos_buffer_upto)
//   posting_writer.rs:363-455            add_position (128-delta blocks through ForUtil::write_block)
//   posting_writer.rs:457-474            finish_doc
//   posting_writer.rs:477-591            finish_term (last_pos_block_offset, vint tail of positions)
//   skip_writer.rs:138-205, 261-289      reset_skip / buffer_skip / write_skip_data_local with positions
//   skip_reader.rs:315-356, 385-453      init / seek_child / set_last_skip_data / read_skip_data with positions
//   posting_reader.rs:112-158            open (.pos header check)
//   posting_reader.rs:1180-1230          BlockPostingIterator::reset (last_pos_block_fp)
//   posting_reader.rs:1232-1283          refill_docs
//   posting_reader.rs:1285-1324          refill_positions
//   posting_reader.rs:1326-1350          skip_positions
//   posting_reader.rs:1357-1380          next_position
//   posting_reader.rs:1400-1437          next
//   posting_reader.rs:1439-1587          advance (PF arm)
//   posting_writer.rs:363-455, 477-591   the payload / offset arms of add_position and finish_term (".pay" blocks, the VInt tail)
//   posting_reader.rs:1595-2337          EverythingIterator
#pragma once
#include <memory>
#include <vector>

#include "postings.hpp"

namespace orc {

static const char* const POS_CODEC = "Lucene50PostingsWriterPos";  // posting_reader.rs:53
static const char* const PAY_CODEC = "Lucene50PostingsWriterPay";  // posting_reader.rs:54
constexpr int32_t INDEX_MAX_POSITION = INT32_MAX - 128;            // index/mod.rs INDEX_MAX_POSITION

// ---- skip list with position pointers ------------------------------------------------------------------------------

struct PosSkipWriter : SkipWriter {
  std::vector<int64_t> last_skip_pos_pointer, last_skip_pay_pointer;
  int64_t cur_pos_pointer = 0, last_pos_fp = 0, cur_pay_pointer = 0, last_pay_fp = 0;
  int32_t cur_pos_buffer_upto = 0, cur_payload_byte_upto = 0;
  bool field_has_offsets = false, field_has_payloads = false;  // (field_has_positions: always, in this subclass)
  PosSkipWriter(int max_skip_levels, uint32_t block_size, uint32_t doc_count)
      : SkipWriter(max_skip_levels, block_size, doc_count), last_skip_pos_pointer((size_t)max_skip_levels, 0),
        last_skip_pay_pointer((size_t)max_skip_levels, 0) {}
  // skip_writer.rs:128-137
  void set_field(bool has_offsets, bool has_payloads) { field_has_offsets = has_offsets; field_has_payloads = has_payloads; }
  // skip_writer.rs:138-149
  void reset_skip_pos(int64_t doc_fp, int64_t pos_fp, int64_t pay_fp = 0) {
    reset_skip(doc_fp);
    last_pos_fp = pos_fp;
    if (field_has_offsets || field_has_payloads) last_pay_fp = pay_fp;
  }
  // skip_writer.rs:151-183
  void init_skip() override {
    const bool was = initialized;
    SkipWriter::init_skip();
    if (!was) {
      std::fill(last_skip_pos_pointer.begin(), last_skip_pos_pointer.end(), last_pos_fp);
      if (field_has_offsets || field_has_payloads) std::fill(last_skip_pay_pointer.begin(), last_skip_pay_pointer.end(), last_pay_fp);
    }
  }
  // skip_writer.rs:187-205
  void buffer_skip_pos(int32_t doc, uint32_t num_docs, int64_t pos_fp, int32_t pos_buffer_upto, int64_t doc_out_pointer, int64_t pay_fp = 0,
                       int32_t payload_byte_upto = 0) {
    init_skip();
    cur_doc = doc;
    cur_doc_pointer = doc_out_pointer;
    cur_pos_pointer = pos_fp;
    cur_pay_pointer = pay_fp;
    cur_pos_buffer_upto = pos_buffer_upto;
    cur_payload_byte_upto = payload_byte_upto;
    buffer_skip_levels(num_docs);
  }
  // skip_writer.rs:261-289
  void write_skip_data_local(int level) override {
    SkipWriter::write_skip_data_local(level);
    skip_buffer[(size_t)level].write_vlong(cur_pos_pointer - last_skip_pos_pointer[(size_t)level]);
    last_skip_pos_pointer[(size_t)level] = cur_pos_pointer;
    skip_buffer[(size_t)level].write_vint(cur_pos_buffer_upto);
    if (field_has_payloads) skip_buffer[(size_t)level].write_vint(cur_payload_byte_upto);
    if (field_has_offsets || field_has_payloads) {
      skip_buffer[(size_t)level].write_vlong(cur_pay_pointer - last_skip_pay_pointer[(size_t)level]);
      last_skip_pay_pointer[(size_t)level] = cur_pay_pointer;
    }
  }
};

struct PosSkipReader : SkipReader {
  std::vector<int64_t> pos_pointer, pay_pointer;
  std::vector<int32_t> pos_buffer_upto, payload_byte_upto;
  int64_t last_pos_pointer = 0, last_pay_pointer = 0;
  int32_t last_pos_buffer_upto = 0, last_payload_byte_upto = 0;
  bool has_payload_upto, has_pay_pointer;  // skip_reader.rs:259-268: payload_byte_upto iff payloads; pay_pointer iff offsets or payloads
  // skip_reader.rs:222-299 (has_pos: always, in this subclass)
  PosSkipReader(const ByteIn& stream, int max_skip_levels, bool has_offsets = false, bool has_payloads = false)
      : SkipReader(stream, max_skip_levels), pos_pointer((size_t)max_skip_levels, 0), pay_pointer((size_t)max_skip_levels, 0),
        pos_buffer_upto((size_t)max_skip_levels, 0), payload_byte_upto((size_t)max_skip_levels, 0), has_payload_upto(has_payloads),
        has_pay_pointer(has_offsets || has_payloads) {}
  // skip_reader.rs:315-356
  void init_pos(int64_t skip_ptr, int64_t doc_base_pointer, int64_t pos_base_pointer, int32_t df, int64_t pay_base_pointer = 0) {
    init(skip_ptr, doc_base_pointer, df);
    last_pos_pointer = pos_base_pointer;
    last_pay_pointer = pay_base_pointer;
    std::fill(pos_pointer.begin(), pos_pointer.end(), pos_base_pointer);
    if (has_pay_pointer) std::fill(pay_pointer.begin(), pay_pointer.end(), pay_base_pointer);
    // (pos_buffer_upto / payload_byte_upto keep their values across init in the reference; every path sets them before they are read)
  }
  int64_t get_pos_pointer() const { return last_pos_pointer; }
  int32_t get_pos_buffer_upto() const { return last_pos_buffer_upto; }
  int64_t get_pay_pointer() const { return last_pay_pointer; }
  int32_t get_payload_byte_upto() const { return last_payload_byte_upto; }
  // skip_reader.rs:385-408
  void seek_child(int level) override {
    SkipReader::seek_child(level);
    pos_pointer[(size_t)level] = last_pos_pointer;
    pos_buffer_upto[(size_t)level] = last_pos_buffer_upto;
    if (has_payload_upto) payload_byte_upto[(size_t)level] = last_payload_byte_upto;
    if (has_pay_pointer) pay_pointer[(size_t)level] = last_pay_pointer;
  }
  // skip_reader.rs:410-429
  void set_last_skip_data(int level) override {
    SkipReader::set_last_skip_data(level);
    last_pos_pointer = pos_pointer[(size_t)level];
    last_pos_buffer_upto = pos_buffer_upto[(size_t)level];
    if (has_pay_pointer) last_pay_pointer = pay_pointer[(size_t)level];
    if (has_payload_upto) last_payload_byte_upto = payload_byte_upto[(size_t)level];
  }
  // skip_reader.rs:431-453
  int32_t read_skip_data(int level) override {
    const int32_t delta = SkipReader::read_skip_data(level);
    pos_pointer[(size_t)level] += skip_stream[(size_t)level].read_vlong();
    pos_buffer_upto[(size_t)level] = skip_stream[(size_t)level].read_vint();
    if (has_payload_upto) payload_byte_upto[(size_t)level] = skip_stream[(size_t)level].read_vint();
    if (has_pay_pointer) pay_pointer[(size_t)level] += skip_stream[(size_t)level].read_vlong();
    return delta;
  }
};

// ---- writer -----------------------------------------------------------------------------------------------------------

struct PosTermState {  // blocktree/mod.rs:33-59, the fields a positions field uses
  BlockTermState base;
  int64_t pos_start_fp = 0;
  int64_t pay_start_fp = 0;  // fields with payloads or offsets
  int64_t last_pos_block_offset = -1;
};

struct PosPostingsWriter {
  ByteOut doc_out, pos_out, pay_out;
  int64_t doc_start_fp = 0, pos_start_fp = 0, pay_start_fp = 0;
  std::vector<int32_t> doc_delta_buffer, freq_buffer, pos_delta_buffer, payload_length_buffer, offset_start_delta_buffer, offset_length_buffer;
  std::vector<uint8_t> payload_bytes;
  int doc_buffer_upto = 0, pos_buffer_upto = 0, payload_byte_upto = 0;
  int32_t last_block_doc_id = 0, last_doc_id = 0, last_position = 0, last_start_offset = 0, doc_count = 0;
  int64_t last_block_pos_fp = 0, last_block_pay_fp = 0;
  int32_t last_block_pos_buffer_upto = 0, last_block_payload_byte_upto = 0;
  ForUtil for_util;
  PosSkipWriter skip_writer;
  bool use_simd;
  bool write_offsets, write_payloads;  // set_field_base (posting_writer.rs:260-287): one positions field per writer here

  // posting_writer.rs:116-251: every file gets an index header; only .doc carries the ForUtil table; ".pay" exists when the
  // field stores payloads or offsets
  PosPostingsWriter(int32_t max_doc, int32_t version, const uint8_t segment_id[ID_LENGTH], const std::string& suffix, bool offsets = false,
                    bool payloads = false)
      : doc_delta_buffer(MAX_DATA_SIZE, 0), freq_buffer(MAX_DATA_SIZE, 0), pos_delta_buffer(MAX_DATA_SIZE, 0),
        payload_length_buffer(MAX_DATA_SIZE, 0), offset_start_delta_buffer(MAX_DATA_SIZE, 0), offset_length_buffer(MAX_DATA_SIZE, 0),
        payload_bytes(128, 0), skip_writer(MAX_SKIP_LEVELS, BLOCK_SIZE, (uint32_t)max_doc), write_offsets(offsets), write_payloads(payloads) {
    write_index_header(doc_out, DOC_CODEC, version, segment_id, suffix);
    for_util = ForUtil::with_output(0.0f, doc_out);
    write_index_header(pos_out, POS_CODEC, version, segment_id, suffix);
    if (has_pay()) write_index_header(pay_out, PAY_CODEC, version, segment_id, suffix);
    use_simd = version > VERSION_START;
    skip_writer.set_field(write_offsets, write_payloads);  // posting_writer.rs:720-727
  }
  bool has_pay() const { return write_offsets || write_payloads; }
  // posting_writer.rs:289-302
  void start_term() {
    doc_start_fp = doc_out.file_pointer();
    pos_start_fp = pos_out.file_pointer();
    if (has_pay()) pay_start_fp = pay_out.file_pointer();
    last_doc_id = 0;
    last_block_doc_id = -1;
    skip_writer.reset_skip_pos(doc_start_fp, pos_start_fp, pay_start_fp);
  }
  // posting_writer.rs:304-361
  void start_doc(int32_t doc_id, int32_t term_doc_freq) {
    if (last_block_doc_id != -1 && doc_buffer_upto == 0)
      skip_writer.buffer_skip_pos(last_block_doc_id, (uint32_t)doc_count, last_block_pos_fp, last_block_pos_buffer_upto,
                                  doc_out.file_pointer(), last_block_pay_fp, last_block_payload_byte_upto);
    const int32_t doc_delta = doc_id - last_doc_id;
    if (doc_id < 0 || (doc_count > 0 && doc_delta <= 0)) throw OracleError(E_CORRUPT_INDEX, "docs out of order");
    doc_delta_buffer[(size_t)doc_buffer_upto] = doc_delta;
    freq_buffer[(size_t)doc_buffer_upto] = term_doc_freq;
    doc_buffer_upto++;
    doc_count++;
    if (doc_buffer_upto == BLOCK_SIZE) {
      for_util.write_block(doc_delta_buffer.data(), doc_out, use_simd);
      for_util.write_block(freq_buffer.data(), doc_out, use_simd);
    }
    last_doc_id = doc_id;
    last_position = 0;
    last_start_offset = 0;
  }
  // posting_writer.rs:363-455
  void add_position(int32_t position, const uint8_t* payload = nullptr, size_t payload_len = 0, int32_t start_offset = 0, int32_t end_offset = 0) {
    if (position > INDEX_MAX_POSITION) throw OracleError(E_CORRUPT_INDEX, "position is too large (> INDEX_MAX_POSITION)");
    if (position < 0) throw OracleError(E_CORRUPT_INDEX, "position < 0");
    pos_delta_buffer[(size_t)pos_buffer_upto] = position - last_position;
    if (write_payloads) {
      if (payload_len == 0) {
        payload_length_buffer[(size_t)pos_buffer_upto] = 0;
      } else {
        payload_length_buffer[(size_t)pos_buffer_upto] = (int32_t)payload_len;
        const size_t total = (size_t)payload_byte_upto + payload_len;
        if (total > payload_bytes.size()) payload_bytes.resize(total, 0);
        std::memcpy(payload_bytes.data() + payload_byte_upto, payload, payload_len);
        payload_byte_upto += (int)payload_len;
      }
    }
    if (write_offsets) {
      if (start_offset < last_start_offset || end_offset < start_offset) throw OracleError(E_ILLEGAL_ARGUMENT, "offsets out of order");  // debug_assert
      offset_start_delta_buffer[(size_t)pos_buffer_upto] = start_offset - last_start_offset;
      offset_length_buffer[(size_t)pos_buffer_upto] = end_offset - start_offset;
      last_start_offset = start_offset;
    }
    pos_buffer_upto++;
    last_position = position;
    if (pos_buffer_upto == BLOCK_SIZE) {
      for_util.write_block(pos_delta_buffer.data(), pos_out, use_simd);
      if (write_payloads) {
        for_util.write_block(payload_length_buffer.data(), pay_out, use_simd);
        pay_out.write_vint(payload_byte_upto);
        pay_out.write_bytes(payload_bytes.data(), (size_t)payload_byte_upto);
        payload_byte_upto = 0;
      }
      if (write_offsets) {
        for_util.write_block(offset_start_delta_buffer.data(), pay_out, use_simd);
        for_util.write_block(offset_length_buffer.data(), pay_out, use_simd);
      }
      pos_buffer_upto = 0;
    }
  }
  // posting_writer.rs:457-474
  void finish_doc() {
    if (doc_buffer_upto == BLOCK_SIZE) {
      last_block_doc_id = last_doc_id;
      if (has_pay()) last_block_pay_fp = pay_out.file_pointer();
      last_block_pos_fp = pos_out.file_pointer();
      last_block_pos_buffer_upto = pos_buffer_upto;
      last_block_payload_byte_upto = payload_byte_upto;
      doc_buffer_upto = 0;
    }
  }
  // posting_writer.rs:477-591
  void finish_term(PosTermState& state) {
    if (!(state.base.doc_freq > 0) || state.base.doc_freq != doc_count) throw OracleError(E_ILLEGAL_STATE, "doc_freq mismatch");
    int32_t singleton_doc_id;
    if (state.base.doc_freq == 1) {
      singleton_doc_id = doc_delta_buffer[0];
    } else {
      for (int i = 0; i < doc_buffer_upto; i++) {
        const int32_t doc_delta = doc_delta_buffer[(size_t)i], freq = freq_buffer[(size_t)i];
        if (freq == 1) doc_out.write_vint(doc_delta << 1 | 1);
        else { doc_out.write_vint(doc_delta << 1); doc_out.write_vint(freq); }
      }
      singleton_doc_id = -1;
    }
    int64_t last_pos_block_offset = -1;
    if (state.base.total_term_freq > BLOCK_SIZE) last_pos_block_offset = pos_out.file_pointer() - pos_start_fp;
    if (pos_buffer_upto > 0) {
      int32_t last_payload_length = -1, last_offset_length = -1;  // force the first lengths to be written
      size_t payload_bytes_read_upto = 0;
      for (int i = 0; i < pos_buffer_upto; i++) {
        const int32_t pos_delta = pos_delta_buffer[(size_t)i];
        if (write_payloads) {
          const int32_t payload_length = payload_length_buffer[(size_t)i];
          if (payload_length != last_payload_length) {
            last_payload_length = payload_length;
            pos_out.write_vint(pos_delta << 1 | 1);
            pos_out.write_vint(payload_length);
          } else {
            pos_out.write_vint(pos_delta << 1);
          }
          if (payload_length != 0) {
            pos_out.write_bytes(payload_bytes.data() + payload_bytes_read_upto, (size_t)payload_length);
            payload_bytes_read_upto += (size_t)payload_length;
          }
        } else {
          pos_out.write_vint(pos_delta);
        }
        if (write_offsets) {
          const int32_t delta = offset_start_delta_buffer[(size_t)i], length = offset_length_buffer[(size_t)i];
          if (length == last_offset_length) {
            pos_out.write_vint(delta << 1);
          } else {
            pos_out.write_vint(delta << 1 | 1);
            pos_out.write_vint(length);
            last_offset_length = length;
          }
        }
      }
      if (write_payloads) payload_byte_upto = 0;
    }
    const int64_t skip_offset = (doc_count > BLOCK_SIZE) ? skip_writer.write_skip(doc_out) - doc_start_fp : -1;
    state.base.doc_start_fp = doc_start_fp;
    state.pos_start_fp = pos_start_fp;
    state.pay_start_fp = pay_start_fp;
    state.base.singleton_doc_id = singleton_doc_id;
    state.base.skip_offset = skip_offset;
    state.last_pos_block_offset = last_pos_block_offset;
    doc_buffer_upto = 0;
    pos_buffer_upto = 0;
    last_doc_id = 0;
    doc_count = 0;
  }
  // posting_writer.rs:610-619
  void close() {
    write_footer(doc_out);
    write_footer(pos_out);
    if (has_pay()) write_footer(pay_out);
  }
};

// ---- reader -----------------------------------------------------------------------------------------------------------

// posting_reader.rs:112-158: the .pos file is opened next to .doc with the same version; footer located
struct PosFile {
  const uint8_t* data;
  int64_t len;
  PosFile(const uint8_t* d, int64_t l, int32_t doc_version) : data(d), len(l) {
    ByteIn in(d, l);
    check_index_header(in, POS_CODEC, doc_version, doc_version);
    retrieve_checksum(d, (size_t)l);
  }
};

// posting_reader.rs:131-156: ".pay" is opened when the segment has a field with payloads or offsets
struct PayFile {
  const uint8_t* data;
  int64_t len;
  PayFile(const uint8_t* d, int64_t l, int32_t doc_version) : data(d), len(l) {
    ByteIn in(d, l);
    check_index_header(in, PAY_CODEC, doc_version, doc_version);
    retrieve_checksum(d, (size_t)l);
  }
};

// What FieldInfo tells an iterator about the field (posting_reader.rs:1161-1163, 1707-1717)
struct PosFieldFlags {
  bool has_offsets = false, has_payloads = false;
};

struct BlockPostingIterator {
  int32_t doc_delta_buffer[MAX_DATA_SIZE + 8], freq_buffer[MAX_DATA_SIZE + 8], pos_delta_buffer[MAX_DATA_SIZE + 8];
  int32_t doc_buffer_upto = 0, pos_buffer_upto = 0;
  std::unique_ptr<PosSkipReader> skipper;
  bool skipped = false;
  ByteIn doc_in, pos_in;
  int32_t doc_freq = 0, doc_upto = 0, doc = 0, accum = 0, freq_ = 0, position = 0, pos_pending_count = 0;
  int64_t total_term_freq = 0, pos_pending_fp = 0, doc_term_start_fp = 0, pos_term_start_fp = 0, skip_offset = 0, last_pos_block_fp = 0;
  int32_t next_skip_doc = 0, singleton_doc_id = 0;
  const PostingsReader* reader;

  bool index_has_offsets = false, index_has_payloads = false;
  int64_t pay_term_start_fp = 0;

  BlockPostingIterator(const PostingsReader* r, const PosFile* pf, const PosTermState& st, PosFieldFlags field = PosFieldFlags())
      : doc_in(r->data, r->len), pos_in(pf->data, pf->len), reader(r), index_has_offsets(field.has_offsets),
        index_has_payloads(field.has_payloads) {
    reset(st);
  }
  // posting_reader.rs:1180-1230
  void reset(const PosTermState& st) {
    doc_freq = st.base.doc_freq;
    doc_term_start_fp = st.base.doc_start_fp;
    pos_term_start_fp = st.pos_start_fp;
    pay_term_start_fp = st.pay_start_fp;
    skip_offset = st.base.skip_offset;
    total_term_freq = st.base.total_term_freq;
    singleton_doc_id = st.base.singleton_doc_id;
    if (doc_freq > 1) doc_in.seek(doc_term_start_fp);
    pos_pending_fp = pos_term_start_fp;
    pos_pending_count = 0;
    if (total_term_freq < BLOCK_SIZE) last_pos_block_fp = pos_term_start_fp;
    else if (total_term_freq == BLOCK_SIZE) last_pos_block_fp = -1;
    else last_pos_block_fp = pos_term_start_fp + st.last_pos_block_offset;
    doc = -1;
    accum = 0;
    doc_upto = 0;
    next_skip_doc = doc_freq > BLOCK_SIZE ? BLOCK_SIZE - 1 : NO_MORE_DOCS;
    doc_buffer_upto = BLOCK_SIZE;
    skipped = false;
  }
  // posting_reader.rs:1232-1283
  void refill_docs() {
    const int32_t left = doc_freq - doc_upto;
    if (left >= BLOCK_SIZE) {
      if (reader->for_util.read_block(doc_in, doc_delta_buffer, true, reader->use_simd) != 0)
        throw OracleError(E_UNSUPPORTED, "EF/BITSET/FULL blocks are never written by Rucene");
      reader->for_util.read_block(doc_in, freq_buffer, false, reader->use_simd);
    } else if (doc_freq == 1) {
      doc_delta_buffer[0] = singleton_doc_id;
      freq_buffer[0] = (int32_t)total_term_freq;
    } else {
      read_vint_block(doc_in, doc_delta_buffer, freq_buffer, left, true);
    }
    doc_buffer_upto = 0;
  }
  // posting_reader.rs:1285-1324: this iterator serves a field with payloads / offsets too (whenever the caller asked for
  // positions only, :189-212) — it walks past the payload bytes and the offset words of the trailing VInt block
  void refill_positions() {
    if (pos_in.file_pointer() == last_pos_block_fp) {
      const int count = (int)(total_term_freq % BLOCK_SIZE);
      int32_t payload_length = 0;
      for (int i = 0; i < count; i++) {
        const int32_t code = pos_in.read_vint();
        if (index_has_payloads) {
          if ((code & 1) != 0) payload_length = pos_in.read_vint();
          pos_delta_buffer[i] = (int32_t)((uint32_t)code >> 1);
          if (payload_length != 0) pos_in.seek(pos_in.file_pointer() + payload_length);
        } else {
          pos_delta_buffer[i] = code;
        }
        if (index_has_offsets && (pos_in.read_vint() & 1) != 0) (void)pos_in.read_vint();  // offset length changed
      }
    } else {
      reader->for_util.read_block(pos_in, pos_delta_buffer, false, reader->use_simd);
    }
  }
  // posting_reader.rs:1326-1350
  void skip_positions() {
    int32_t to_skip = pos_pending_count - freq_;
    const int32_t left_in_block = BLOCK_SIZE - pos_buffer_upto;
    if (to_skip < left_in_block) {
      pos_buffer_upto += to_skip;
    } else {
      to_skip -= left_in_block;
      while (to_skip >= BLOCK_SIZE) {
        if (pos_in.file_pointer() == last_pos_block_fp) throw OracleError(E_ILLEGAL_STATE, "skipping past the last position block");
        reader->for_util.skip_block(pos_in);
        to_skip -= BLOCK_SIZE;
      }
      refill_positions();
      pos_buffer_upto = to_skip;
    }
    position = 0;
  }
  int32_t freq() const { return freq_; }
  int32_t doc_id() const { return doc; }
  size_t cost() const { return (size_t)doc_freq; }
  // posting_reader.rs:1357-1380
  int32_t next_position() {
    if (pos_pending_count <= 0) throw OracleError(E_ILLEGAL_STATE, "next_position() called more than freq() times");
    if (pos_pending_fp != -1) {
      pos_in.seek(pos_pending_fp);
      pos_pending_fp = -1;
      pos_buffer_upto = BLOCK_SIZE;
    }
    if (pos_pending_count > freq_) {
      skip_positions();
      pos_pending_count = freq_;
    }
    if (pos_buffer_upto == BLOCK_SIZE) {
      refill_positions();
      pos_buffer_upto = 0;
    }
    position += pos_delta_buffer[pos_buffer_upto];
    pos_buffer_upto++;
    pos_pending_count--;
    return position;
  }
  // posting_reader.rs:1400-1437 (PF arm)
  int32_t next() {
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    doc = accum + doc_delta_buffer[doc_buffer_upto];
    accum = doc;
    freq_ = freq_buffer[doc_buffer_upto];
    pos_pending_count += freq_;
    doc_buffer_upto++;
    doc_upto++;
    position = 0;
    return doc;
  }
  // posting_reader.rs:1439-1587 (PF arm)
  int32_t advance(int32_t target) {
    if (target == NO_MORE_DOCS) { doc = NO_MORE_DOCS; return doc; }
    if (target > next_skip_doc) {
      if (!skipper) skipper.reset(new PosSkipReader(doc_in, MAX_SKIP_LEVELS, index_has_offsets, index_has_payloads));
      if (!skipped) {
        skipper->init_pos(doc_term_start_fp + skip_offset, doc_term_start_fp, pos_term_start_fp, doc_freq, pay_term_start_fp);
        skipped = true;
      }
      const int32_t new_doc_upto = skipper->skip_to(target) + 1;
      if (new_doc_upto > doc_upto) {
        doc_upto = new_doc_upto;
        doc_buffer_upto = BLOCK_SIZE;
        accum = skipper->doc();
        doc_in.seek(skipper->get_doc_pointer());
        pos_pending_fp = skipper->get_pos_pointer();
        pos_pending_count = skipper->get_pos_buffer_upto();
      }
      next_skip_doc = skipper->next_skip_doc();
    }
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    while (true) {
      accum += doc_delta_buffer[doc_buffer_upto];
      freq_ = freq_buffer[doc_buffer_upto];
      pos_pending_count += freq_;
      doc_buffer_upto++;
      doc_upto++;
      if (accum >= target) break;
      if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    }
    position = 0;
    doc = accum;
    return doc;
  }
};

// ---- EverythingIterator (posting_reader.rs:1595-2337): positions + payloads + offsets --------------------------------------
// What Lucene50PostingsReader::postings hands out when the caller asks for PAYLOADS or OFFSETS on a field that has them
// (:213-229). Docs are PF blocks only (EF / BITSET arms: postings.hpp's BlockDocIterator; no Rucene build writes them).
constexpr uint16_t FLAG_POSITIONS = FLAG_FREQS | (1 << 4);   // posting_iterator.rs:18-49
constexpr uint16_t FLAG_OFFSETS = FLAG_POSITIONS | (1 << 5);
constexpr uint16_t FLAG_PAYLOADS = FLAG_POSITIONS | (1 << 6);
constexpr uint16_t FLAG_ALL = FLAG_OFFSETS | FLAG_PAYLOADS;
inline bool feature_requested(uint16_t flags, uint16_t feature) { return (flags & feature) == feature; }

struct EverythingIterator {
  int32_t doc_delta_buffer[MAX_DATA_SIZE + 8], freq_buffer[MAX_DATA_SIZE + 8], pos_delta_buffer[MAX_DATA_SIZE + 8];
  int32_t payload_length_buffer[MAX_DATA_SIZE + 8], offset_start_delta_buffer[MAX_DATA_SIZE + 8], offset_length_buffer[MAX_DATA_SIZE + 8];
  std::vector<uint8_t> payload_bytes;
  int32_t payload_byte_upto = 0, payload_length = 0;
  int32_t last_start_offset = 0, start_offset_ = 0, end_offset_ = 0;
  int32_t doc_buffer_upto = 0, pos_buffer_upto = 0;
  std::unique_ptr<PosSkipReader> skipper;
  bool skipped = false;
  ByteIn doc_in, pos_in, pay_in;
  bool index_has_offsets, index_has_payloads;
  int32_t doc_freq = 0, doc_upto = 0, doc = 0, accum = 0, freq_ = 0, position = 0, pos_pending_count = 0;
  int64_t total_term_freq = 0, pos_pending_fp = 0, pay_pending_fp = 0, doc_term_start_fp = 0, pos_term_start_fp = 0, pay_term_start_fp = 0;
  int64_t last_pos_block_fp = 0, skip_offset = 0;
  int32_t next_skip_doc = 0, singleton_doc_id = 0;
  bool needs_offsets = false, needs_payloads = false;
  const PostingsReader* reader;

  // posting_reader.rs:1699-1782
  EverythingIterator(const PostingsReader* r, const PosFile* pf, const PayFile* yf, PosFieldFlags field, const PosTermState& st, uint16_t flags)
      : doc_in(r->data, r->len), pos_in(pf->data, pf->len), pay_in(yf->data, yf->len), index_has_offsets(field.has_offsets),
        index_has_payloads(field.has_payloads), reader(r) {
    if (index_has_offsets) { start_offset_ = 0; end_offset_ = 0; } else { start_offset_ = -1; end_offset_ = -1; }
    if (index_has_payloads) payload_bytes.assign(128, 0);
    reset(st, flags);
  }
  // posting_reader.rs:1784-1841
  void reset(const PosTermState& st, uint16_t flags) {
    doc_freq = st.base.doc_freq;
    doc_term_start_fp = st.base.doc_start_fp;
    pos_term_start_fp = st.pos_start_fp;
    pay_term_start_fp = st.pay_start_fp;
    skip_offset = st.base.skip_offset;
    total_term_freq = st.base.total_term_freq;
    singleton_doc_id = st.base.singleton_doc_id;
    if (doc_freq > 1) doc_in.seek(doc_term_start_fp);
    pos_pending_fp = pos_term_start_fp;
    pay_pending_fp = pay_term_start_fp;
    pos_pending_count = 0;
    if (total_term_freq < BLOCK_SIZE) last_pos_block_fp = pos_term_start_fp;
    else if (total_term_freq == BLOCK_SIZE) last_pos_block_fp = -1;
    else last_pos_block_fp = pos_term_start_fp + st.last_pos_block_offset;
    needs_offsets = feature_requested(flags, FLAG_OFFSETS);
    needs_payloads = feature_requested(flags, FLAG_PAYLOADS);
    doc = -1;
    accum = 0;
    doc_upto = 0;
    next_skip_doc = doc_freq > BLOCK_SIZE ? BLOCK_SIZE - 1 : NO_MORE_DOCS;
    doc_buffer_upto = BLOCK_SIZE;
    skipped = false;
  }
  // posting_reader.rs:1843-1893 (PF arm)
  void refill_docs() {
    const int32_t left = doc_freq - doc_upto;
    if (left >= BLOCK_SIZE) {
      if (reader->for_util.read_block(doc_in, doc_delta_buffer, true, reader->use_simd) != 0)
        throw OracleError(E_UNSUPPORTED, "EF/BITSET/FULL blocks are never written by Rucene");
      reader->for_util.read_block(doc_in, freq_buffer, false, reader->use_simd);
    } else if (doc_freq == 1) {
      doc_delta_buffer[0] = singleton_doc_id;
      freq_buffer[0] = (int32_t)total_term_freq;
    } else {
      read_vint_block(doc_in, doc_delta_buffer, freq_buffer, left, true);
    }
    doc_buffer_upto = 0;
  }
  // posting_reader.rs:1895-1991
  void refill_positions() {
    if (pos_in.file_pointer() == last_pos_block_fp) {
      const int count = (int)(total_term_freq % BLOCK_SIZE);
      int32_t payload_len = 0, offset_length = 0;
      payload_byte_upto = 0;
      for (int i = 0; i < count; i++) {
        const int32_t code = pos_in.read_vint();
        if (index_has_payloads) {
          if ((code & 1) != 0) payload_len = pos_in.read_vint();
          payload_length_buffer[i] = payload_len;
          pos_delta_buffer[i] = (int32_t)((uint32_t)code >> 1);
          if (payload_len != 0) {
            if ((size_t)(payload_byte_upto + payload_len) > payload_bytes.size()) payload_bytes.resize((size_t)(payload_byte_upto + payload_len), 0);
            pos_in.read_exact(payload_bytes.data() + payload_byte_upto, (size_t)payload_len);
            payload_byte_upto += payload_len;
          }
        } else {
          pos_delta_buffer[i] = code;
        }
        if (index_has_offsets) {
          const int32_t delta_code = pos_in.read_vint();
          if ((delta_code & 1) != 0) offset_length = pos_in.read_vint();
          offset_start_delta_buffer[i] = (int32_t)((uint32_t)delta_code >> 1);
          offset_length_buffer[i] = offset_length;
        }
      }
      payload_byte_upto = 0;
    } else {
      reader->for_util.read_block(pos_in, pos_delta_buffer, false, reader->use_simd);
      if (index_has_payloads) {
        if (needs_payloads) {
          reader->for_util.read_block(pay_in, payload_length_buffer, false, reader->use_simd);
          const size_t num_bytes = (size_t)pay_in.read_vint();
          if (num_bytes > payload_bytes.size()) payload_bytes.resize(num_bytes, 0);
          pay_in.read_exact(payload_bytes.data(), num_bytes);
        } else {
          reader->for_util.skip_block(pay_in);                 // the lengths
          const int32_t num_bytes = pay_in.read_vint();
          pay_in.seek(pay_in.file_pointer() + num_bytes);      // the bytes
        }
        payload_byte_upto = 0;
      }
      if (index_has_offsets) {
        if (needs_offsets) {
          reader->for_util.read_block(pay_in, offset_start_delta_buffer, false, reader->use_simd);
          reader->for_util.read_block(pay_in, offset_length_buffer, false, reader->use_simd);
        } else {
          reader->for_util.skip_block(pay_in);
          reader->for_util.skip_block(pay_in);
        }
      }
    }
  }
  // posting_reader.rs:1997-2049
  void skip_positions() {
    int32_t to_skip = pos_pending_count - freq_;
    const int32_t left_in_block = BLOCK_SIZE - pos_buffer_upto;
    if (to_skip < left_in_block) {
      const int32_t end = pos_buffer_upto + to_skip;
      while (pos_buffer_upto < end) {
        if (index_has_payloads) payload_byte_upto += payload_length_buffer[pos_buffer_upto];
        pos_buffer_upto++;
      }
    } else {
      to_skip -= left_in_block;
      while (to_skip >= BLOCK_SIZE) {
        reader->for_util.skip_block(pos_in);
        if (index_has_payloads) {
          reader->for_util.skip_block(pay_in);
          const int32_t num_bytes = pay_in.read_vint();
          pay_in.seek(pay_in.file_pointer() + num_bytes);
        }
        if (index_has_offsets) {
          reader->for_util.skip_block(pay_in);
          reader->for_util.skip_block(pay_in);
        }
        to_skip -= BLOCK_SIZE;
      }
      refill_positions();
      payload_byte_upto = 0;
      pos_buffer_upto = 0;
      while (pos_buffer_upto < to_skip) {
        if (index_has_payloads) payload_byte_upto += payload_length_buffer[pos_buffer_upto];
        pos_buffer_upto++;
      }
    }
    position = 0;
    last_start_offset = 0;
  }
  int32_t freq() const { return freq_; }
  int32_t doc_id() const { return doc; }
  size_t cost() const { return (size_t)doc_freq; }
  // posting_reader.rs:2069-2119
  int32_t next_position() {
    if (pos_pending_count <= 0) throw OracleError(E_ILLEGAL_STATE, "next_position() called more than freq() times");
    if (pos_pending_fp != -1) {
      pos_in.seek(pos_pending_fp);
      pos_pending_fp = -1;
      if (pay_pending_fp != -1) {
        pay_in.seek(pay_pending_fp);
        pay_pending_fp = -1;
      }
      pos_buffer_upto = BLOCK_SIZE;
    }
    if (pos_pending_count > freq_) {
      skip_positions();
      pos_pending_count = freq_;
    }
    if (pos_buffer_upto == BLOCK_SIZE) {
      refill_positions();
      pos_buffer_upto = 0;
    }
    position += pos_delta_buffer[pos_buffer_upto];
    if (index_has_payloads) {
      payload_length = payload_length_buffer[pos_buffer_upto];
      payload_byte_upto += payload_length;
    }
    if (index_has_offsets) {
      start_offset_ = last_start_offset + offset_start_delta_buffer[pos_buffer_upto];
      end_offset_ = start_offset_ + offset_length_buffer[pos_buffer_upto];
      last_start_offset = start_offset_;
    }
    pos_buffer_upto++;
    pos_pending_count--;
    return position;
  }
  int32_t start_offset() const { return start_offset_; }  // :2121-2127
  int32_t end_offset() const { return end_offset_; }
  // posting_reader.rs:2129-2137
  std::vector<uint8_t> payload() const {
    if (payload_length == 0 || payload_byte_upto < payload_length) return {};
    const size_t end = (size_t)payload_byte_upto, start = end - (size_t)payload_length;
    if (end > payload_bytes.size()) throw OracleError(E_ILLEGAL_STATE, "payload() beyond the loaded bytes (the Rust slice would panic): PAYLOADS not requested?");
    return std::vector<uint8_t>(payload_bytes.begin() + (ptrdiff_t)start, payload_bytes.begin() + (ptrdiff_t)end);
  }
  // posting_reader.rs:2145-2186 (PF arm)
  int32_t next() {
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    doc = accum + doc_delta_buffer[doc_buffer_upto];
    accum = doc;
    freq_ = freq_buffer[doc_buffer_upto];
    pos_pending_count += freq_;
    doc_buffer_upto++;
    doc_upto++;
    position = 0;
    last_start_offset = 0;
    return doc;
  }
  // posting_reader.rs:2188-2331 (PF arm)
  int32_t advance(int32_t target) {
    if (target == NO_MORE_DOCS) { doc = NO_MORE_DOCS; return doc; }
    if (target > next_skip_doc) {
      if (!skipper) skipper.reset(new PosSkipReader(doc_in, MAX_SKIP_LEVELS, index_has_offsets, index_has_payloads));
      if (!skipped) {
        skipper->init_pos(doc_term_start_fp + skip_offset, doc_term_start_fp, pos_term_start_fp, doc_freq, pay_term_start_fp);
        skipped = true;
      }
      const int32_t new_doc_upto = skipper->skip_to(target) + 1;
      if (new_doc_upto > doc_upto) {
        doc_upto = new_doc_upto;
        doc_buffer_upto = BLOCK_SIZE;
        accum = skipper->doc();
        doc_in.seek(skipper->get_doc_pointer());
        pos_pending_fp = skipper->get_pos_pointer();
        pay_pending_fp = skipper->get_pay_pointer();
        pos_pending_count = skipper->get_pos_buffer_upto();
        last_start_offset = 0;
        payload_byte_upto = skipper->get_payload_byte_upto();
      }
      next_skip_doc = skipper->next_skip_doc();
    }
    if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    if (doc_buffer_upto == BLOCK_SIZE) refill_docs();
    while (true) {
      accum += doc_delta_buffer[doc_buffer_upto];
      freq_ = freq_buffer[doc_buffer_upto];
      pos_pending_count += freq_;
      doc_buffer_upto++;
      doc_upto++;
      if (accum >= target) break;
      if (doc_upto == doc_freq) { doc = NO_MORE_DOCS; return doc; }
    }
    position = 0;
    last_start_offset = 0;
    doc = accum;
    return doc;
  }
};

}  // namespace orc
