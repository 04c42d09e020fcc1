// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of the POSITIONS side of Rucene's Lucene50 postings: the ".pos" file, the position pointers inside the
// ".doc" skip entries, and BlockPostingIterator (docs + freqs + positions). Fields indexed with
// IndexOptions::DocsAndFreqsAndPositions without payloads or offsets — what a plain text field is — are covered; the payload
// and offset arms (a third file, ".pay") are NOT restated and are refused.
// Groundwork for SURVEY.md §8(f)3 (positions + PhraseScorer); no product code reads positions yet.
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
#pragma once
#include <memory>
#include <vector>

#include "postings.hpp"

namespace orc {

static const char* const POS_CODEC = "Lucene50PostingsWriterPos";  // posting_reader.rs:53
constexpr int32_t INDEX_MAX_POSITION = INT32_MAX - 128;            // index/mod.rs INDEX_MAX_POSITION

// ---- skip list with position pointers ------------------------------------------------------------------------------

struct PosSkipWriter : SkipWriter {
  std::vector<int64_t> last_skip_pos_pointer;
  int64_t cur_pos_pointer = 0, last_pos_fp = 0;
  int32_t cur_pos_buffer_upto = 0;
  PosSkipWriter(int max_skip_levels, uint32_t block_size, uint32_t doc_count)
      : SkipWriter(max_skip_levels, block_size, doc_count), last_skip_pos_pointer((size_t)max_skip_levels, 0) {}
  // skip_writer.rs:138-149
  void reset_skip_pos(int64_t doc_fp, int64_t pos_fp) { reset_skip(doc_fp); last_pos_fp = pos_fp; }
  // skip_writer.rs:151-183
  void init_skip() override {
    const bool was = initialized;
    SkipWriter::init_skip();
    if (!was) std::fill(last_skip_pos_pointer.begin(), last_skip_pos_pointer.end(), last_pos_fp);
  }
  // skip_writer.rs:187-205
  void buffer_skip_pos(int32_t doc, uint32_t num_docs, int64_t pos_fp, int32_t pos_buffer_upto, int64_t doc_out_pointer) {
    init_skip();
    cur_doc = doc;
    cur_doc_pointer = doc_out_pointer;
    cur_pos_pointer = pos_fp;
    cur_pos_buffer_upto = pos_buffer_upto;
    buffer_skip_levels(num_docs);
  }
  // skip_writer.rs:261-289
  void write_skip_data_local(int level) override {
    SkipWriter::write_skip_data_local(level);
    skip_buffer[(size_t)level].write_vlong(cur_pos_pointer - last_skip_pos_pointer[(size_t)level]);
    last_skip_pos_pointer[(size_t)level] = cur_pos_pointer;
    skip_buffer[(size_t)level].write_vint(cur_pos_buffer_upto);
  }
};

struct PosSkipReader : SkipReader {
  std::vector<int64_t> pos_pointer;
  std::vector<int32_t> pos_buffer_upto;
  int64_t last_pos_pointer = 0;
  int32_t last_pos_buffer_upto = 0;
  PosSkipReader(const ByteIn& stream, int max_skip_levels)
      : SkipReader(stream, max_skip_levels), pos_pointer((size_t)max_skip_levels, 0), pos_buffer_upto((size_t)max_skip_levels, 0) {}
  // skip_reader.rs:315-356
  void init_pos(int64_t skip_ptr, int64_t doc_base_pointer, int64_t pos_base_pointer, int32_t df) {
    init(skip_ptr, doc_base_pointer, df);
    last_pos_pointer = pos_base_pointer;
    std::fill(pos_pointer.begin(), pos_pointer.end(), pos_base_pointer);
    std::fill(pos_buffer_upto.begin(), pos_buffer_upto.end(), 0);
  }
  int64_t get_pos_pointer() const { return last_pos_pointer; }
  int32_t get_pos_buffer_upto() const { return last_pos_buffer_upto; }
  // skip_reader.rs:385-408
  void seek_child(int level) override {
    SkipReader::seek_child(level);
    pos_pointer[(size_t)level] = last_pos_pointer;
    pos_buffer_upto[(size_t)level] = last_pos_buffer_upto;
  }
  // skip_reader.rs:410-429
  void set_last_skip_data(int level) override {
    SkipReader::set_last_skip_data(level);
    last_pos_pointer = pos_pointer[(size_t)level];
    last_pos_buffer_upto = pos_buffer_upto[(size_t)level];
  }
  // skip_reader.rs:431-453
  int32_t read_skip_data(int level) override {
    const int32_t delta = SkipReader::read_skip_data(level);
    pos_pointer[(size_t)level] += skip_stream[(size_t)level].read_vlong();
    pos_buffer_upto[(size_t)level] = skip_stream[(size_t)level].read_vint();
    return delta;
  }
};

// ---- writer -----------------------------------------------------------------------------------------------------------

struct PosTermState {  // blocktree/mod.rs:33-59, the fields a positions field uses
  BlockTermState base;
  int64_t pos_start_fp = 0;
  int64_t last_pos_block_offset = -1;
};

struct PosPostingsWriter {
  ByteOut doc_out, pos_out;
  int64_t doc_start_fp = 0, pos_start_fp = 0;
  std::vector<int32_t> doc_delta_buffer, freq_buffer, pos_delta_buffer;
  int doc_buffer_upto = 0, pos_buffer_upto = 0;
  int32_t last_block_doc_id = 0, last_doc_id = 0, last_position = 0, doc_count = 0;
  int64_t last_block_pos_fp = 0;
  int32_t last_block_pos_buffer_upto = 0;
  ForUtil for_util;
  PosSkipWriter skip_writer;
  bool use_simd;

  // posting_writer.rs:116-251: both files get an index header; only .doc carries the ForUtil table
  PosPostingsWriter(int32_t max_doc, int32_t version, const uint8_t segment_id[ID_LENGTH], const std::string& suffix)
      : doc_delta_buffer(MAX_DATA_SIZE, 0), freq_buffer(MAX_DATA_SIZE, 0), pos_delta_buffer(MAX_DATA_SIZE, 0),
        skip_writer(MAX_SKIP_LEVELS, BLOCK_SIZE, (uint32_t)max_doc) {
    write_index_header(doc_out, DOC_CODEC, version, segment_id, suffix);
    for_util = ForUtil::with_output(0.0f, doc_out);
    write_index_header(pos_out, POS_CODEC, version, segment_id, suffix);
    use_simd = version > VERSION_START;
  }
  // posting_writer.rs:289-302
  void start_term() {
    doc_start_fp = doc_out.file_pointer();
    pos_start_fp = pos_out.file_pointer();
    last_doc_id = 0;
    last_block_doc_id = -1;
    skip_writer.reset_skip_pos(doc_start_fp, pos_start_fp);
  }
  // posting_writer.rs:304-361
  void start_doc(int32_t doc_id, int32_t term_doc_freq) {
    if (last_block_doc_id != -1 && doc_buffer_upto == 0)
      skip_writer.buffer_skip_pos(last_block_doc_id, (uint32_t)doc_count, last_block_pos_fp, last_block_pos_buffer_upto,
                                  doc_out.file_pointer());
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
  }
  // posting_writer.rs:363-455 (no payloads, no offsets)
  void add_position(int32_t position) {
    if (position > INDEX_MAX_POSITION) throw OracleError(E_CORRUPT_INDEX, "position is too large (> INDEX_MAX_POSITION)");
    if (position < 0) throw OracleError(E_CORRUPT_INDEX, "position < 0");
    pos_delta_buffer[(size_t)pos_buffer_upto] = position - last_position;
    pos_buffer_upto++;
    last_position = position;
    if (pos_buffer_upto == BLOCK_SIZE) {
      for_util.write_block(pos_delta_buffer.data(), pos_out, use_simd);
      pos_buffer_upto = 0;
    }
  }
  // posting_writer.rs:457-474
  void finish_doc() {
    if (doc_buffer_upto == BLOCK_SIZE) {
      last_block_doc_id = last_doc_id;
      last_block_pos_fp = pos_out.file_pointer();
      last_block_pos_buffer_upto = pos_buffer_upto;
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
    for (int i = 0; i < pos_buffer_upto; i++) pos_out.write_vint(pos_delta_buffer[(size_t)i]);
    const int64_t skip_offset = (doc_count > BLOCK_SIZE) ? skip_writer.write_skip(doc_out) - doc_start_fp : -1;
    state.base.doc_start_fp = doc_start_fp;
    state.pos_start_fp = pos_start_fp;
    state.base.singleton_doc_id = singleton_doc_id;
    state.base.skip_offset = skip_offset;
    state.last_pos_block_offset = last_pos_block_offset;
    doc_buffer_upto = 0;
    pos_buffer_upto = 0;
    last_doc_id = 0;
    doc_count = 0;
  }
  // posting_writer.rs:610-619
  void close() { write_footer(doc_out); write_footer(pos_out); }
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

  BlockPostingIterator(const PostingsReader* r, const PosFile* pf, const PosTermState& st)
      : doc_in(r->data, r->len), pos_in(pf->data, pf->len), reader(r) {
    reset(st);
  }
  // posting_reader.rs:1180-1230
  void reset(const PosTermState& st) {
    doc_freq = st.base.doc_freq;
    doc_term_start_fp = st.base.doc_start_fp;
    pos_term_start_fp = st.pos_start_fp;
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
  // posting_reader.rs:1285-1324 (no payloads, no offsets)
  void refill_positions() {
    if (pos_in.file_pointer() == last_pos_block_fp) {
      const int count = (int)(total_term_freq % BLOCK_SIZE);
      for (int i = 0; i < count; i++) pos_delta_buffer[i] = pos_in.read_vint();
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
      if (!skipper) skipper.reset(new PosSkipReader(doc_in, MAX_SKIP_LEVELS));
      if (!skipped) {
        skipper->init_pos(doc_term_start_fp + skip_offset, doc_term_start_fp, pos_term_start_fp, doc_freq);
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

}  // namespace orc
