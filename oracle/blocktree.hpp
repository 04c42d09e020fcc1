// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's block-tree term dictionary (.tim) and terms index (.tip): the writer that lays
// terms out in prefix-shared blocks with floor blocks, and the reader's `seek_exact` that walks the index FST to
// the deepest block, picks the floor block, scans it and decodes the term's metadata into a BlockTermState.
//
// PARITY UNPINNED: the reference holds no test and no golden file for the block-tree writer/reader
// (SURVEY.md §4); the source text is the only authority and each function cites the lines it restates.
//
// Follows (paths relative to /root/reference/src/core/codec/postings):
//   blocktree/blocktree_writer.rs:140-192   BlockTreeTermsWriter::new (two index headers + postings header)
//   blocktree/blocktree_writer.rs:228-270   close (field summary, trailers, footers)
//   blocktree/blocktree_writer.rs:383-489   TermsWriter::write_blocks
//   blocktree/blocktree_writer.rs:497-700   write_block
//   blocktree/blocktree_writer.rs:703-774   write / push_term
//   blocktree/blocktree_writer.rs:777-850   finish
//   blocktree/blocktree_writer.rs:909-990   PendingBlock::compile_index / append, encode_output
//   posting_writer.rs:595-607, 688-733      Lucene50PostingsWriter::init / encode_term / set_field
//   blocktree/blocktree_reader.rs:132-304   BlockTreeTermsReader::new (summary parsing and its checks)
//   blocktree/blocktree_reader.rs:410-460   FieldReader::new
//   blocktree/blocktree_reader.rs:1184-1232 push_frame_by_data / push_frame_by_fp
//   blocktree/blocktree_reader.rs:1364-1550 seek_exact (restated for a fresh iterator: no frame reuse)
//   blocktree/term_iter_frame.rs:147-160    set_floor_data
//   blocktree/term_iter_frame.rs:176-232    load_block
//   blocktree/term_iter_frame.rs:334-372    scan_to_floor_frame
//   blocktree/term_iter_frame.rs:374-402    decode_metadata
//   blocktree/term_iter_frame.rs:456-640    scan_to_term_leaf / scan_to_term_non_leaf
//   posting_reader.rs:160-180, 264-306      Lucene50PostingsReader::init / lucene50_decode_term
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fst.hpp"
#include "postings.hpp"
#include "store.hpp"

namespace orc {

static const char* const TERMS_DICT_CODEC = "BlockTreeTermsDict";        // blocktree_reader.rs:52
static const char* const TERMS_INDEX_CODEC = "BlockTreeTermsIndex";      // blocktree_reader.rs:72
static const char* const POSTINGS_TERMS_CODEC = "Lucene50PostingsWriterTerms";  // posting_reader.rs:51
constexpr int32_t BT_VERSION_START = 0, BT_VERSION_AUTO_PREFIX_TERMS = 1, BT_VERSION_AUTO_PREFIX_TERMS_REMOVED = 3,
                  BT_VERSION_CURRENT = 3;  // blocktree_reader.rs:55-68
constexpr int64_t BT_OUTPUT_FLAGS_IS_FLOOR = 1, BT_OUTPUT_FLAGS_HAS_TERMS = 2;  // blocktree_reader.rs:45-48
constexpr int BT_DEFAULT_MIN_BLOCK_SIZE = 25, BT_DEFAULT_MAX_BLOCK_SIZE = 48;   // lucene50 posting_format.rs defaults

// doc/index_options: ordinal order as in core/doc (Null, Docs, DocsAndFreqs, DocsAndFreqsAndPositions,
// DocsAndFreqsAndPositionsAndOffsets)
enum IndexOptions { IO_NULL = 0, IO_DOCS = 1, IO_DOCS_FREQS = 2, IO_DOCS_FREQS_POS = 3, IO_DOCS_FREQS_POS_OFFS = 4 };

struct BtFieldInfo {
  int32_t number = 0;
  int32_t index_options = IO_DOCS_FREQS;
  bool has_payloads = false;
  bool has_positions() const { return index_options >= IO_DOCS_FREQS_POS; }
  bool has_offsets() const { return index_options >= IO_DOCS_FREQS_POS_OFFS; }
  int longs_size() const { return has_positions() ? ((has_payloads || has_offsets()) ? 3 : 2) : 1; }  // posting_writer.rs:720-733
};

// blocktree/mod.rs:33-59, every field
struct FullTermState {
  BlockTermState base;
  int64_t pos_start_fp = 0, pay_start_fp = 0, last_pos_block_offset = -1;
};

// ---- writer --------------------------------------------------------------------------------------------------------

struct BlockTreeTermsWriter {
  struct IndexEntry { Bytes prefix, output; };  // one (input, output) pair of a block's index FST
  struct Pending {
    bool is_term = true;
    Bytes bytes;  // term bytes, or block prefix
    FullTermState state;
    // block:
    int64_t fp = 0;
    bool has_terms = false, is_floor = false;
    int floor_lead_byte = -1;
    std::vector<IndexEntry> index;        // the compiled index, as its enumeration
    std::vector<std::vector<IndexEntry>> sub_indices;
  };
  struct FieldMeta {
    BtFieldInfo info;
    Bytes root_code, min_term, max_term;
    int64_t num_terms = 0, index_start_fp = 0, sum_total_term_freq = 0, sum_doc_freq = 0;
    int32_t doc_count = 0;
    int longs_size = 1;
  };

  ByteOut terms_out, index_out;
  int min_items, max_items;
  std::vector<FieldMeta> fields;
  bool closed = false;

  // per-field TermsWriter state
  BtFieldInfo field;
  std::vector<Pending> pending;
  std::vector<Pending> new_blocks;
  Bytes last_term;
  std::vector<size_t> prefix_starts;
  int64_t num_terms = 0, sum_total_term_freq = 0, sum_doc_freq = 0;
  Bytes first_pending_term, last_pending_term;
  FullTermState last_state;  // Lucene50PostingsWriter::last_state

  // blocktree_writer.rs:140-192 + posting_writer.rs:595-607
  BlockTreeTermsWriter(const uint8_t id[ID_LENGTH], const std::string& suffix, int min_items_in_block = BT_DEFAULT_MIN_BLOCK_SIZE,
                       int max_items_in_block = BT_DEFAULT_MAX_BLOCK_SIZE)
      : min_items(min_items_in_block), max_items(max_items_in_block) {
    if (min_items <= 1) throw OracleError(E_ILLEGAL_ARGUMENT, "min_items_in_block must be >= 2");
    if (min_items > max_items) throw OracleError(E_ILLEGAL_ARGUMENT, "min_items_in_block > max_items_in_block");
    if (2 * (min_items - 1) > max_items) throw OracleError(E_ILLEGAL_ARGUMENT, "2 * (min_items_in_block - 1) > max_items_in_block");
    write_index_header(terms_out, TERMS_DICT_CODEC, BT_VERSION_CURRENT, id, suffix);
    write_index_header(index_out, TERMS_INDEX_CODEC, BT_VERSION_CURRENT, id, suffix);
    write_index_header(terms_out, POSTINGS_TERMS_CODEC, VERSION_CURRENT, id, suffix);
    terms_out.write_vint(BLOCK_SIZE);
  }

  void start_field(const BtFieldInfo& info) {  // TermsWriter::new + postings_writer.set_field
    if (info.index_options == IO_NULL) throw OracleError(E_ILLEGAL_ARGUMENT, "field is not indexed");
    field = info;
    pending.clear();
    new_blocks.clear();
    last_term.clear();
    prefix_starts.assign(8, 0);
    num_terms = sum_total_term_freq = sum_doc_freq = 0;
    first_pending_term.clear();
    last_pending_term.clear();
    last_state = FullTermState();
  }

  // posting_writer.rs:688-718
  void encode_term(int64_t longs[3], ByteOut& out, const FullTermState& st, bool absolute) {
    if (absolute) last_state = FullTermState();
    longs[0] = st.base.doc_start_fp - last_state.base.doc_start_fp;
    if (field.has_positions()) {
      longs[1] = st.pos_start_fp - last_state.pos_start_fp;
      if (field.has_payloads || field.has_offsets()) longs[2] = st.pay_start_fp - last_state.pay_start_fp;
    }
    if (st.base.singleton_doc_id != -1) out.write_vint(st.base.singleton_doc_id);
    if (field.has_positions() && st.last_pos_block_offset != -1) out.write_vlong(st.last_pos_block_offset);
    if (st.base.skip_offset != -1) out.write_vlong(st.base.skip_offset);
    last_state = st;
  }

  static int64_t encode_output(int64_t fp, bool has_terms, bool is_floor) {  // blocktree_writer.rs:1001-1006
    return (fp << 2) | (has_terms ? BT_OUTPUT_FLAGS_HAS_TERMS : 0) | (is_floor ? BT_OUTPUT_FLAGS_IS_FLOOR : 0);
  }

  // blocktree_writer.rs:909-975. `blocks` = the other floor blocks of this group (self excluded).
  static void compile_index(Pending& self, std::vector<Pending>& blocks) {
    if (self.is_floor != !blocks.empty()) throw OracleError(E_ILLEGAL_STATE, "floor block bookkeeping");
    ByteOut scratch;
    scratch.write_vlong(encode_output(self.fp, self.has_terms, self.is_floor));
    if (self.is_floor) {
      scratch.write_vint((int32_t)blocks.size());
      for (const Pending& b : blocks) {
        if (b.floor_lead_byte == -1 || b.fp <= self.fp) throw OracleError(E_ILLEGAL_STATE, "floor block order");
        scratch.write_byte((uint8_t)b.floor_lead_byte);
        scratch.write_vlong(((b.fp - self.fp) << 1) | (b.has_terms ? 1 : 0));
      }
    }
    self.index.clear();
    self.index.push_back({self.bytes, scratch.buf});
    for (auto& sub : self.sub_indices) self.index.insert(self.index.end(), sub.begin(), sub.end());
    self.sub_indices.clear();
    for (Pending& b : blocks) {
      for (auto& sub : b.sub_indices) self.index.insert(self.index.end(), sub.begin(), sub.end());
      b.sub_indices.clear();
    }
  }

  // blocktree_writer.rs:497-700
  Pending write_block(size_t prefix_length, bool is_floor, int floor_lead_label, size_t start, size_t end, bool has_terms,
                      bool has_sub_blocks) {
    const int64_t start_fp = terms_out.file_pointer();
    const bool has_floor_lead_label = is_floor && floor_lead_label != -1;
    Bytes prefix(last_term.begin(), last_term.begin() + prefix_length);
    const size_t num_entries = end - start;
    int32_t code = (int32_t)(num_entries << 1);
    if (end == pending.size()) code |= 1;  // last block of its floor group
    terms_out.write_vint(code);

    const bool is_leaf_block = !has_sub_blocks;
    ByteOut suffix_writer, stats_writer, meta_writer, bytes_writer;
    std::vector<std::vector<IndexEntry>> sub_indices;
    bool absolute = true;
    int64_t longs[3] = {0, 0, 0};
    const int longs_size = field.longs_size();
    for (size_t i = start; i < end; i++) {
      Pending& ent = pending[i];
      if (ent.is_term) {
        const size_t suffix = ent.bytes.size() - prefix_length;
        suffix_writer.write_vint(is_leaf_block ? (int32_t)suffix : (int32_t)(suffix << 1));
        suffix_writer.write_bytes(ent.bytes.data() + prefix_length, suffix);
        stats_writer.write_vint(ent.state.base.doc_freq);
        if (field.index_options != IO_DOCS) {
          if (ent.state.base.total_term_freq < ent.state.base.doc_freq) throw OracleError(E_ILLEGAL_STATE, "ttf < df");
          stats_writer.write_vlong(ent.state.base.total_term_freq - ent.state.base.doc_freq);
        }
        encode_term(longs, bytes_writer, ent.state, absolute);
        for (int p = 0; p < longs_size; p++) {
          if (longs[p] < 0) throw OracleError(E_ILLEGAL_STATE, "negative metadata long");
          meta_writer.write_vlong(longs[p]);
        }
        bytes_writer.write_to(meta_writer);
        bytes_writer.reset();
        absolute = false;
      } else {
        if (is_leaf_block) throw OracleError(E_ILLEGAL_STATE, "sub-block in a leaf block");
        const size_t suffix = ent.bytes.size() - prefix_length;
        suffix_writer.write_vint((int32_t)((suffix << 1) | 1));
        suffix_writer.write_bytes(ent.bytes.data() + prefix_length, suffix);
        suffix_writer.write_vlong(start_fp - ent.fp);
        sub_indices.push_back(std::move(ent.index));
      }
    }
    terms_out.write_vint((int32_t)(suffix_writer.file_pointer() << 1) + (is_leaf_block ? 1 : 0));
    suffix_writer.write_to(terms_out);
    terms_out.write_vint((int32_t)stats_writer.file_pointer());
    stats_writer.write_to(terms_out);
    terms_out.write_vint((int32_t)meta_writer.file_pointer());
    meta_writer.write_to(terms_out);

    if (has_floor_lead_label) prefix.push_back((uint8_t)floor_lead_label);
    Pending block;
    block.is_term = false;
    block.bytes = prefix;
    block.fp = start_fp;
    block.has_terms = has_terms;
    block.is_floor = is_floor;
    block.floor_lead_byte = floor_lead_label;
    block.sub_indices = std::move(sub_indices);
    return block;
  }

  // blocktree_writer.rs:383-489
  void write_blocks(size_t prefix_length, size_t count) {
    if (!(prefix_length > 0 || count == pending.size())) throw OracleError(E_ILLEGAL_STATE, "write_blocks arguments");
    int last_suffix_lead_label = -1;
    bool has_terms = false, has_sub_blocks = false;
    const size_t start = pending.size() - count, end = pending.size();
    size_t next_block_start = start;
    int next_floor_lead_label = -1;
    for (size_t i = start; i < end; i++) {
      const Pending& ent = pending[i];
      int suffix_lead_label;
      if (ent.is_term && ent.bytes.size() == prefix_length) suffix_lead_label = -1;
      else suffix_lead_label = ent.bytes[prefix_length];
      if (suffix_lead_label != last_suffix_lead_label) {
        const size_t items_in_block = i - next_block_start;
        if (items_in_block >= (size_t)min_items && end - next_block_start > (size_t)max_items) {
          const bool is_floor = items_in_block < count;
          new_blocks.push_back(write_block(prefix_length, is_floor, next_floor_lead_label, next_block_start, i, has_terms,
                                           has_sub_blocks));
          has_terms = has_sub_blocks = false;
          next_floor_lead_label = suffix_lead_label;
          next_block_start = i;
        }
        last_suffix_lead_label = suffix_lead_label;
      }
      if (ent.is_term) has_terms = true; else has_sub_blocks = true;
    }
    if (next_block_start < end) {
      const size_t items_in_block = end - next_block_start;
      const bool is_floor = items_in_block < count;
      new_blocks.push_back(write_block(prefix_length, is_floor, next_floor_lead_label, next_block_start, end, has_terms,
                                       has_sub_blocks));
    }
    Pending first = std::move(new_blocks.front());
    new_blocks.erase(new_blocks.begin());
    compile_index(first, new_blocks);
    pending.resize(pending.size() - count);
    pending.push_back(std::move(first));
    new_blocks.clear();
  }

  // blocktree_writer.rs:741-774
  void push_term(const Bytes& text) {
    const size_t limit = std::min(last_term.size(), text.size());
    size_t pos = 0;
    while (pos < limit && last_term[pos] == text[pos]) pos++;
    const size_t last_term_len = last_term.size();
    for (size_t i = 0; i < last_term_len - pos; i++) {
      const size_t idx = last_term_len - 1 - i;
      const size_t prefix_top_size = pending.size() - prefix_starts[idx];
      if (prefix_top_size >= (size_t)min_items) {
        write_blocks(idx + 1, prefix_top_size);
        prefix_starts[idx] -= prefix_top_size - 1;
      }
    }
    if (prefix_starts.size() < text.size()) prefix_starts.resize(text.size(), 0);
    for (size_t i = pos; i < text.size(); i++) prefix_starts[i] = pending.size();
    last_term = text;
  }

  // blocktree_writer.rs:703-738 (the postings themselves were written by orc::PostingsWriter, which produced `state`)
  void write_term(const Bytes& text, const FullTermState& state) {
    if (state.base.doc_freq == 0) throw OracleError(E_ILLEGAL_STATE, "doc_freq == 0");
    push_term(text);
    sum_doc_freq += state.base.doc_freq;
    sum_total_term_freq += state.base.total_term_freq;
    num_terms++;
    Pending p;
    p.is_term = true;
    p.bytes = text;
    p.state = state;
    pending.push_back(std::move(p));
    if (num_terms == 1) first_pending_term = text;
    last_pending_term = text;
  }

  // blocktree_writer.rs:777-850. doc_count = cardinality of docs_seen, supplied by the caller.
  void finish_field(int32_t doc_count) {
    if (num_terms == 0) return;
    push_term(Bytes());
    push_term(Bytes());
    write_blocks(0, pending.size());
    if (pending.size() != 1 || pending[0].is_term) throw OracleError(E_ILLEGAL_STATE, "no single root block");
    Pending root = std::move(pending[0]);
    pending.clear();
    if (!root.bytes.empty() || root.index.empty() || !root.index[0].prefix.empty())
      throw OracleError(E_ILLEGAL_STATE, "root block without an empty-prefix output");
    FstBuilder builder(true, false);  // blocktree_writer.rs:947-957
    for (const IndexEntry& e : root.index) builder.add(e.prefix, e.output);
    if (!builder.finish()) throw OracleError(E_ILLEGAL_STATE, "empty terms index");
    FieldMeta meta;
    meta.info = field;
    meta.root_code = builder.fst.empty_output;
    meta.num_terms = num_terms;
    meta.index_start_fp = index_out.file_pointer();
    builder.fst.save(index_out);
    meta.sum_total_term_freq = sum_total_term_freq;
    meta.sum_doc_freq = sum_doc_freq;
    meta.doc_count = doc_count;
    meta.longs_size = field.longs_size();
    meta.min_term = first_pending_term;
    meta.max_term = last_pending_term;
    fields.push_back(std::move(meta));
  }

  // blocktree_writer.rs:228-270
  void close() {
    if (closed) return;
    closed = true;
    const int64_t dir_start = terms_out.file_pointer(), index_dir_start = index_out.file_pointer();
    terms_out.write_vint((int32_t)fields.size());
    for (const FieldMeta& f : fields) {
      terms_out.write_vint(f.info.number);
      terms_out.write_vlong(f.num_terms);
      terms_out.write_vint((int32_t)f.root_code.size());
      terms_out.write_bytes(f.root_code.data(), f.root_code.size());
      if (f.info.index_options != IO_DOCS) terms_out.write_vlong(f.sum_total_term_freq);
      terms_out.write_vlong(f.sum_doc_freq);
      terms_out.write_vint(f.doc_count);
      terms_out.write_vint(f.longs_size);
      index_out.write_vlong(f.index_start_fp);
      terms_out.write_vint((int32_t)f.min_term.size());
      terms_out.write_bytes(f.min_term.data(), f.min_term.size());
      terms_out.write_vint((int32_t)f.max_term.size());
      terms_out.write_bytes(f.max_term.data(), f.max_term.size());
    }
    terms_out.write_long(dir_start);
    write_footer(terms_out);
    index_out.write_long(index_dir_start);
    write_footer(index_out);
  }
};

// ---- reader --------------------------------------------------------------------------------------------------------

struct BlockTreeTermsReader {
  struct FieldReader {
    BtFieldInfo info;
    int64_t num_terms = 0, sum_total_term_freq = -1, sum_doc_freq = 0, index_start_fp = 0, root_block_fp = 0;
    int32_t doc_count = 0, longs_size = 1;
    Bytes root_code, min_term, max_term;
    Fst index;
  };

  const uint8_t* tim;
  size_t tim_len;
  int32_t version = 0;
  std::map<int32_t, FieldReader> fields;  // by field number

  static int64_t seek_dir(ByteIn& in, size_t len) {  // blocktree_reader.rs:327-332
    if (len < (size_t)FOOTER_LENGTH + 8) throw OracleError(E_CORRUPT_INDEX, "file too short for a directory pointer");
    in.seek((int64_t)len - FOOTER_LENGTH - 8);
    int64_t dir = in.read_long();
    if (dir < 0 || dir > (int64_t)len) throw OracleError(E_CORRUPT_INDEX, "directory pointer out of range");
    in.seek(dir);
    return dir;
  }

  // blocktree_reader.rs:132-304. `infos` plays the part of SegmentReadState::field_infos.
  BlockTreeTermsReader(const uint8_t* tim_, size_t tim_len_, const uint8_t* tip, size_t tip_len,
                       const std::vector<BtFieldInfo>& infos, int32_t max_doc)
      : tim(tim_), tim_len(tim_len_) {
    ByteIn terms_in(tim, tim_len), index_in(tip, tip_len);
    version = check_index_header(terms_in, TERMS_DICT_CODEC, BT_VERSION_START, BT_VERSION_CURRENT);
    if (version >= BT_VERSION_AUTO_PREFIX_TERMS && version < BT_VERSION_AUTO_PREFIX_TERMS_REMOVED)
      throw OracleError(E_UNSUPPORTED, "auto-prefix term dictionaries (versions 1-2) are not restated");
    check_index_header(index_in, TERMS_INDEX_CODEC, version, version);
    check_index_header(terms_in, POSTINGS_TERMS_CODEC, VERSION_START, VERSION_CURRENT);  // postings_reader.init
    int32_t index_block_size = terms_in.read_vint();
    if (index_block_size != BLOCK_SIZE) throw OracleError(E_ILLEGAL_STATE, "index-time BLOCK_SIZE != read-time BLOCK_SIZE");
    retrieve_checksum(tim, tim_len);
    seek_dir(terms_in, tim_len);
    seek_dir(index_in, tip_len);
    int32_t num_fields = terms_in.read_vint();
    if (num_fields < 0) throw OracleError(E_CORRUPT_INDEX, "invalid num_fields");
    for (int32_t i = 0; i < num_fields; i++) {
      FieldReader fr;
      int32_t field = terms_in.read_vint();
      fr.num_terms = terms_in.read_vlong();
      if (fr.num_terms <= 0) throw OracleError(E_CORRUPT_INDEX, "Illegal num_terms");
      int32_t num_bytes = terms_in.read_vint();
      if (num_bytes < 0) throw OracleError(E_CORRUPT_INDEX, "invalid root_code");
      fr.root_code.resize((size_t)num_bytes);
      terms_in.read_exact(fr.root_code.data(), fr.root_code.size());
      const BtFieldInfo* info = nullptr;
      for (const BtFieldInfo& fi : infos) if (fi.number == field) info = &fi;
      if (!info) throw OracleError(E_CORRUPT_INDEX, "invalid field number");
      fr.info = *info;
      fr.sum_total_term_freq = info->index_options == IO_DOCS ? -1 : terms_in.read_vlong();
      fr.sum_doc_freq = terms_in.read_vlong();
      fr.doc_count = terms_in.read_vint();
      fr.longs_size = terms_in.read_vint();
      if (fr.longs_size < 0 || fr.longs_size > 3) throw OracleError(E_CORRUPT_INDEX, "invalid longs_size");
      auto read_bytes = [&](Bytes& b) {
        int32_t n = terms_in.read_vint();
        if (n < 0) throw OracleError(E_CORRUPT_INDEX, "invalid term length");
        b.resize((size_t)n);
        terms_in.read_exact(b.data(), b.size());
      };
      read_bytes(fr.min_term);
      read_bytes(fr.max_term);
      if (fr.doc_count < 0 || fr.doc_count > max_doc) throw OracleError(E_CORRUPT_INDEX, "invalid doc_count");
      if (fr.sum_doc_freq < fr.doc_count) throw OracleError(E_CORRUPT_INDEX, "invalid sum_doc_freq");
      if (fr.sum_total_term_freq != -1 && fr.sum_total_term_freq < fr.sum_doc_freq)
        throw OracleError(E_CORRUPT_INDEX, "invalid sum_total_term_freq");
      fr.index_start_fp = index_in.read_vlong();
      if (fields.count(field)) throw OracleError(E_CORRUPT_INDEX, "duplicated field");
      {  // FieldReader::new, blocktree_reader.rs:428-440
        ByteIn rc(fr.root_code.data(), fr.root_code.size());
        fr.root_block_fp = (int64_t)((uint64_t)rc.read_vlong() >> 2);
        ByteIn clone(tip, tip_len);
        clone.seek(fr.index_start_fp);
        fr.index = Fst::from_input(clone);
      }
      fields.emplace(field, std::move(fr));
    }
  }

  // One SegmentTermsIterFrame, reduced to what a fresh seek_exact touches.
  struct Frame {
    int64_t fp = 0, fp_orig = 0, fp_end = 0;
    size_t prefix = 0;
    bool has_terms = false, is_floor = false, is_last_in_floor = false, is_leaf_block = false;
    Bytes floor_data;
    size_t floor_pos = 0;
    int32_t num_follow_floor_blocks = 0, next_floor_label = 0;
    int32_t ent_count = 0, next_ent = -1, term_block_ord = 0;
    Bytes suffix_bytes, stat_bytes, meta_bytes;
  };

  // blocktree_reader.rs:1184-1232 + term_iter_frame.rs:147-160
  static Frame push_frame(const Bytes& frame_data, size_t length) {
    Frame f;
    ByteIn r(frame_data.data(), frame_data.size());
    int64_t code = r.read_vlong();
    f.fp = f.fp_orig = (int64_t)((uint64_t)code >> 2);
    f.has_terms = code & BT_OUTPUT_FLAGS_HAS_TERMS;
    f.is_floor = code & BT_OUTPUT_FLAGS_IS_FLOOR;
    f.prefix = length;
    if (f.is_floor) {
      f.floor_data.assign(frame_data.begin() + r.file_pointer(), frame_data.end());
      ByteIn fr(f.floor_data.data(), f.floor_data.size());
      f.num_follow_floor_blocks = fr.read_vint();
      f.next_floor_label = fr.read_byte();
      f.floor_pos = (size_t)fr.file_pointer();
    }
    return f;
  }

  // term_iter_frame.rs:334-372
  static void scan_to_floor_frame(Frame& f, const Bytes& target) {
    if (!f.is_floor || target.size() <= f.prefix) return;
    const int target_label = target[f.prefix];
    if (target_label < f.next_floor_label) return;
    ByteIn fr(f.floor_data.data(), f.floor_data.size());
    fr.seek((int64_t)f.floor_pos);
    int64_t new_fp;
    for (;;) {
      int64_t code = fr.read_vlong();
      new_fp = f.fp_orig + (int64_t)((uint64_t)code >> 1);
      f.has_terms = code & 1;
      f.is_last_in_floor = f.num_follow_floor_blocks == 1;
      f.num_follow_floor_blocks--;
      if (f.is_last_in_floor) { f.next_floor_label = 256; break; }
      f.next_floor_label = fr.read_byte();
      if (target_label < f.next_floor_label) break;
    }
    f.floor_pos = (size_t)fr.file_pointer();
    if (new_fp != f.fp) { f.next_ent = -1; f.fp = new_fp; }
  }

  // term_iter_frame.rs:176-232
  void load_block(Frame& f) const {
    if (f.next_ent != -1) return;
    ByteIn in(tim, tim_len);
    in.seek(f.fp);
    int32_t code = in.read_vint();
    f.ent_count = (int32_t)((uint32_t)code >> 1);
    if (f.ent_count <= 0) throw OracleError(E_CORRUPT_INDEX, "empty term block");
    f.is_last_in_floor = code & 1;
    code = in.read_vint();
    f.is_leaf_block = code & 1;
    auto read_blob = [&](Bytes& b, int32_t n) {
      if (n < 0) throw OracleError(E_CORRUPT_INDEX, "negative blob length");
      b.resize((size_t)n);
      in.read_exact(b.data(), b.size());
    };
    read_blob(f.suffix_bytes, (int32_t)((uint32_t)code >> 1));
    read_blob(f.stat_bytes, in.read_vint());
    f.term_block_ord = 0;
    f.next_ent = 0;
    read_blob(f.meta_bytes, in.read_vint());
    f.fp_end = in.file_pointer();
  }

  // term_iter_frame.rs:456-640 with exact_only = true. Returns true iff the term is in this block; on success the
  // frame's next_ent / term_block_ord say how many terms precede-and-include it (decode_metadata's limit).
  static bool scan_to_term(Frame& f, const Bytes& target) {
    ByteIn sr(f.suffix_bytes.data(), f.suffix_bytes.size());
    while (f.next_ent < f.ent_count) {
      f.next_ent++;
      int32_t code = sr.read_vint();
      size_t suffix;
      bool term_exists = true;
      size_t start_byte_pos;
      if (f.is_leaf_block) {
        suffix = (size_t)code;
        start_byte_pos = (size_t)sr.file_pointer();
        sr.get_and_advance(suffix);
      } else {
        suffix = (size_t)((uint32_t)code >> 1);
        start_byte_pos = (size_t)sr.file_pointer();
        sr.get_and_advance(suffix);
        term_exists = (code & 1) == 0;
        if (term_exists) f.term_block_ord++;
        else sr.read_vlong();  // sub_code
      }
      const size_t term_len = f.prefix + suffix;
      const size_t target_limit = std::min(target.size(), term_len);
      size_t target_pos = f.prefix, byte_pos = start_byte_pos;
      int cmp = 0;
      bool stop = false;
      for (;;) {
        if (target_pos < target_limit) {
          cmp = (int)f.suffix_bytes[byte_pos++] - (int)target[target_pos++];
        } else {
          cmp = term_len < target.size() ? -1 : (term_len > target.size() ? 1 : 0);
          stop = true;
        }
        if (cmp < 0) break;           // next entry
        if (cmp > 0) return false;    // NotFound
        if (stop) return term_exists; // Found (a sub-block entry can never equal a target that reached this block)
      }
    }
    return false;  // End
  }

  // term_iter_frame.rs:374-402 + posting_reader.rs:264-306
  static FullTermState decode_metadata(const Frame& f, const FieldReader& fr) {
    const int32_t limit = f.is_leaf_block ? f.next_ent : f.term_block_ord;
    FullTermState st;
    st.base.total_term_freq = -1;
    ByteIn stats(f.stat_bytes.data(), f.stat_bytes.size()), meta(f.meta_bytes.data(), f.meta_bytes.size());
    bool absolute = true;
    int64_t longs[3] = {0, 0, 0};
    for (int32_t upto = 0; upto < limit; upto++) {
      st.base.doc_freq = stats.read_vint();
      if (fr.info.index_options != IO_DOCS) st.base.total_term_freq = st.base.doc_freq + stats.read_vlong();
      for (int i = 0; i < fr.longs_size; i++) longs[i] = meta.read_vlong();
      if (absolute) st.base.doc_start_fp = st.pos_start_fp = st.pay_start_fp = 0;
      st.base.doc_start_fp += longs[0];
      if (fr.info.has_positions()) {
        st.pos_start_fp += longs[1];
        if (fr.info.has_offsets() || fr.info.has_payloads) st.pay_start_fp += longs[2];
      }
      st.base.singleton_doc_id = st.base.doc_freq == 1 ? meta.read_vint() : -1;
      if (fr.info.has_positions()) st.last_pos_block_offset = st.base.total_term_freq > BLOCK_SIZE ? meta.read_vlong() : -1;
      st.base.skip_offset = st.base.doc_freq > BLOCK_SIZE ? meta.read_vlong() : -1;
      absolute = false;
    }
    return st;
  }

  // blocktree_reader.rs:1364-1550, the `current_frame_ord == stack[0].ord` arm (fresh iterator) + term_state()
  bool seek_exact(int32_t field_number, const Bytes& target, FullTermState& out) const {
    auto it = fields.find(field_number);
    if (it == fields.end()) return false;
    const FieldReader& fr = it->second;
    const Fst& index = fr.index;
    RevReader r = index.reader();
    FstArc arc = index.root_arc();
    if (!arc.is_final()) throw OracleError(E_CORRUPT_INDEX, "terms index without a root output");
    Bytes output = arc.output;
    Frame frame = push_frame(bso_add(output, arc.next_final_output), 0);
    size_t target_upto = 0;
    while (target_upto < target.size()) {
      FstArc next;
      if (!index.find_target_arc(target[target_upto], arc, next, r)) break;
      arc = next;
      output = bso_add(output, arc.output);
      target_upto++;
      if (arc.is_final()) frame = push_frame(bso_add(output, arc.next_final_output), target_upto);
    }
    scan_to_floor_frame(frame, target);
    if (!frame.has_terms) return false;
    load_block(frame);
    if (!scan_to_term(frame, target)) return false;
    out = decode_metadata(frame, fr);
    return true;
  }
};

}  // namespace orc
