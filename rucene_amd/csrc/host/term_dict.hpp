// Host-side term dictionary: a segment's block-tree ".tim" (term blocks) + ".tip" (terms index) files ->
// term bytes -> BlockTermState, so that a query arrives as (field, term bytes) and leaves as the
// rgpu_term_state the search kernels take. Replaces, for exact lookups, the reference's
//   BlockTreeTermsReader::new            codec/postings/blocktree/blocktree_reader.rs:132-304   (open + field summary)
//   FieldReader (Terms statistics)       blocktree_reader.rs:390-548
//   SegmentTermIterator::seek_exact      blocktree_reader.rs:1364-1550 + blocktree/term_iter_frame.rs (frames, floor
//                                        blocks, block scan)
//   term_state() / decode_metadata       term_iter_frame.rs:374-402 + posting_reader.rs:264-306 lucene50_decode_term
// (paths relative to /root/reference/src/core).
//
// Design (not the reference's): the reference walks the index FST per lookup, then seeks to one block, scans it and
// decodes the metadata of every term before the hit — pointer chasing sized for an index on disk. Here the whole
// dictionary is resident and a batch carries thousands of lookups, so open() enumerates every block ONCE — starting at
// the root block of each field and following the sub-block pointers and floor-block chains that the ".tim" file
// itself carries — decodes every term's metadata in one forward pass per block, and files (term bytes -> state) in an
// open-addressing hash table. A lookup is then one hash + one compare, independent of block size and tree depth. The
// ".tip" FST only accelerates *finding* a block; since every block is visited anyway it is validated (header, segment
// identity, footer, per-field FST header) but not walked.
//
// Block layout consumed (blocktree_writer.rs:497-700):
//   vint (ent_count << 1 | is_last_in_floor)
//   vint (suffix_bytes << 1 | is_leaf), suffix blob: leaf   -> per entry vint len, bytes
//                                                    inner  -> per entry vint (len << 1 | is_sub_block), bytes,
//                                                              and for a sub-block vlong (this_block_fp - sub_block_fp)
//   vint stats_bytes, stats blob: per term vint doc_freq [, vlong total_term_freq - doc_freq unless IndexOptions::Docs]
//   vint meta_bytes,  meta blob:  per term longs_size vlongs (file-pointer deltas; absolute for a block's first term)
//                                 [vint singleton_doc_id if doc_freq == 1]
//                                 [vlong last_pos_block_offset if positions && total_term_freq > 128]
//                                 [vlong skip_offset if doc_freq > 128]
// Floor blocks of one prefix follow each other in the file; the last one has is_last_in_floor set
// (term_iter_frame.rs:168-174 load_next_floor_block: fp = fp_end).
//
// Error codes are rgpu_status values (include/rucene_gpu.h). No GPU involved.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "norms_format.hpp"

namespace rucene {

struct TermFieldInfo {      // the slice of FieldInfo (codec/field_infos/mod.rs) the dictionary needs
  int32_t number = 0;
  int32_t index_options = 2;  // doc::IndexOptions ordinal: 1 Docs, 2 DocsAndFreqs, 3 +Positions, 4 +Offsets
  int32_t has_payloads = 0;
};

struct TermState {          // == rgpu_term_state
  int64_t doc_start_fp;
  int64_t skip_offset;
  int64_t total_term_freq;
  int32_t doc_freq;
  int32_t singleton_doc_id;
};

struct TermPositions {      // == rgpu_term_positions: the pointers a positions field adds to BlockTermState
  int64_t pos_start_fp;           // where the term's positions start in .pos (0 for fields without positions)
  int64_t pay_start_fp;           // ... its payloads / offsets in .pay (0 unless the field stores them)
  int64_t last_pos_block_offset;  // -1 unless total_term_freq > 128
};

struct TermFieldStats {     // == rgpu_field_stats; Terms::{size, sum_total_term_freq, sum_doc_freq, doc_count}
  int64_t num_terms;
  int64_t sum_total_term_freq;  // -1 for IndexOptions::Docs
  int64_t sum_doc_freq;
  int32_t doc_count;
  int32_t longs_size;
};

class TermDictionary {
 public:
  // Returns 0 or a negative rgpu_status (*why explains).
  static int open(const uint8_t* tim, size_t tim_len, const uint8_t* tip, size_t tip_len, const TermFieldInfo* infos,
                  int32_t n_infos, int32_t max_doc, std::unique_ptr<TermDictionary>* out, std::string* why) {
    const int ERR_ARG = -2, ERR_EOF = -3, ERR_CORRUPT = -4, ERR_UNSUPPORTED = -5, ERR_STATE = -1;
    if (!tim || !tip || (!infos && n_infos > 0) || n_infos < 0 || max_doc < 0 || !out) { *why = "bad arguments"; return ERR_ARG; }
    auto dict = std::unique_ptr<TermDictionary>(new TermDictionary());
    detail::Cursor t{tim, tim_len}, x{tip, tip_len};
    int32_t version = 0, index_version = 0, postings_version = 0;
    const uint8_t *tim_id = nullptr, *tip_id = nullptr, *post_id = nullptr;
    std::string tim_suffix, tip_suffix, post_suffix;
    int rc = detail::read_index_header(t, "BlockTreeTermsDict", 0, 3, &version, &tim_id, &tim_suffix, why);
    if (rc) return rc;
    if (version == 1 || version == 2) { *why = "auto-prefix term dictionaries (format versions 1-2) are not supported"; return ERR_UNSUPPORTED; }
    rc = detail::read_index_header(x, "BlockTreeTermsIndex", version, version, &index_version, &tip_id, &tip_suffix, why);
    if (rc) return rc;
    if (std::memcmp(tim_id, tip_id, 16) != 0 || tim_suffix != tip_suffix) { *why = ".tim and .tip belong to different segments"; return ERR_CORRUPT; }
    // Lucene50PostingsReader::init (posting_reader.rs:160-180): the postings header sits inside .tim
    rc = detail::read_index_header(t, "Lucene50PostingsWriterTerms", 0, 1, &postings_version, &post_id, &post_suffix, why);
    if (rc) return rc;
    const uint32_t index_block_size = t.vint();
    if (!t.ok) { *why = "truncated postings header in .tim"; return ERR_EOF; }
    if (index_block_size != 128) { *why = "index-time BLOCK_SIZE (" + std::to_string(index_block_size) + ") != read-time BLOCK_SIZE (128)"; return ERR_STATE; }
    const size_t blocks_start = t.pos;
    uint64_t stored = 0;
    if (tim_len < 16 + 8) { *why = "misplaced codec footer (file truncated?)"; return ERR_CORRUPT; }
    rc = detail::read_footer(tim, tim_len, tim_len - 16, &stored, why);  // retrieve_checksum
    if (rc) return rc;
    if (tip_len < 16 + 8) { *why = "misplaced codec footer (file truncated?)"; return ERR_CORRUPT; }
    rc = detail::read_footer(tip, tip_len, tip_len - 16, &stored, why);
    if (rc) return rc;
    // seek_dir (blocktree_reader.rs:327-332)
    const uint64_t dir = detail::be64_at(tim + tim_len - 16 - 8), index_dir = detail::be64_at(tip + tip_len - 16 - 8);
    if (dir < blocks_start || dir > tim_len - 16 - 8) { *why = ".tim directory pointer out of range"; return ERR_CORRUPT; }
    if (index_dir > tip_len - 16 - 8) { *why = ".tip directory pointer out of range"; return ERR_CORRUPT; }
    const size_t blocks_end = (size_t)dir;
    t.pos = (size_t)dir;
    t.len = tim_len - 16 - 8;
    x.pos = (size_t)index_dir;
    x.len = tip_len - 16 - 8;
    const uint32_t num_fields = t.vint();
    if (!t.ok || (int32_t)num_fields < 0) { *why = "invalid num_fields"; return ERR_CORRUPT; }
    for (uint32_t i = 0; i < num_fields; ++i) {
      Field f;
      const uint32_t number = t.vint();
      f.stats.num_terms = (int64_t)vlong(t);
      if (!t.ok || f.stats.num_terms <= 0) { *why = "Illegal num_terms for field number: " + std::to_string(number); return ERR_CORRUPT; }
      const uint32_t root_len = t.vint();
      if (!t.ok || t.pos + root_len > t.len) { *why = "invalid root_code for field number: " + std::to_string(number); return ERR_CORRUPT; }
      detail::Cursor root{tim + t.pos, root_len};
      t.pos += root_len;
      const TermFieldInfo* info = nullptr;
      for (int32_t k = 0; k < n_infos; ++k) if ((uint32_t)infos[k].number == number) info = &infos[k];
      if (!info || info->index_options < 1 || info->index_options > 4) { *why = "invalid field number: " + std::to_string(number); return ERR_CORRUPT; }
      f.info = *info;
      f.stats.sum_total_term_freq = info->index_options == 1 ? -1 : (int64_t)vlong(t);
      f.stats.sum_doc_freq = (int64_t)vlong(t);
      f.stats.doc_count = (int32_t)t.vint();
      f.stats.longs_size = (int32_t)t.vint();
      if (!t.ok) { *why = "truncated field summary"; return ERR_EOF; }
      if (f.stats.longs_size < 0 || f.stats.longs_size > 3) { *why = "invalid longs_size for field number: " + std::to_string(number); return ERR_CORRUPT; }
      const bool has_pos = info->index_options >= 3, has_offs = info->index_options >= 4;
      const int want_longs = has_pos ? ((has_offs || info->has_payloads) ? 3 : 2) : 1;  // posting_writer.rs:720-733
      if (f.stats.longs_size != want_longs) { *why = "longs_size does not match the field's index options"; return ERR_CORRUPT; }
      for (int k = 0; k < 2; ++k) {  // min_term, max_term
        const uint32_t n = t.vint();
        if (!t.ok || t.pos + n > t.len) { *why = "truncated field summary"; return ERR_EOF; }
        t.pos += n;
      }
      if (f.stats.doc_count < 0 || f.stats.doc_count > max_doc) { *why = "invalid doc_count: " + std::to_string(f.stats.doc_count) + " max_doc: " + std::to_string(max_doc); return ERR_CORRUPT; }
      if (f.stats.sum_doc_freq < f.stats.doc_count) { *why = "invalid sum_doc_freq"; return ERR_CORRUPT; }
      if (f.stats.sum_total_term_freq != -1 && f.stats.sum_total_term_freq < f.stats.sum_doc_freq) { *why = "invalid sum_total_term_freq"; return ERR_CORRUPT; }
      const uint64_t index_start_fp = vlong(x);
      if (!x.ok) { *why = "truncated .tip directory"; return ERR_EOF; }
      for (const Field& g : dict->fields_) if ((uint32_t)g.info.number == number) { *why = "duplicated field: " + std::to_string(number); return ERR_CORRUPT; }
      // FieldReader::new: root block fp from the root code; the field's FST must start where .tip says it does
      const uint64_t root_code = vlong(root);
      if (!root.ok) { *why = "invalid root_code"; return ERR_CORRUPT; }
      static const uint8_t kFstHeader[] = {0x3F, 0xD7, 0x6C, 0x17, 3, 'F', 'S', 'T'};
      if (index_start_fp > index_dir || index_dir - index_start_fp < sizeof(kFstHeader) + 4 ||
          std::memcmp(tip + index_start_fp, kFstHeader, sizeof(kFstHeader)) != 0) {
        *why = "terms index FST header missing for field number: " + std::to_string(number);
        return ERR_CORRUPT;
      }
      rc = dict->enumerate(tim, blocks_start, blocks_end, root_code >> 2, &f, why);
      if (rc) return rc;
      dict->fields_.push_back(std::move(f));
    }
    *out = std::move(dict);
    return 0;
  }

  const TermFieldStats* field_stats(int32_t field_number) const {
    const Field* f = find_field(field_number);
    return f ? &f->stats : nullptr;
  }

  // seek_exact + term_state. Absent term (or field): returns false and *out is the "absent" state (doc_freq 0).
  bool lookup(int32_t field_number, const uint8_t* term, size_t len, TermState* out, TermPositions* pos_out = nullptr) const {
    const Field* f = find_field(field_number);
    if (pos_out) *pos_out = TermPositions{0, 0, -1};
    if (!f || f->slots.empty()) { *out = TermState{0, -1, 0, 0, -1}; return false; }
    const int64_t e = probe(*f, hash_bytes(term, len), term, len, out);
    if (e >= 0 && pos_out && !f->positions.empty()) *pos_out = f->positions[(size_t)e];
    return e >= 0;
  }

  // n lookups in one field; term i = bytes[offsets[i], offsets[i+1]). A resident dictionary is bound by cache misses
  // (slot -> entry -> term bytes), so the batch is software-pipelined: hash a window ahead and prefetch its slots while
  // the current window is probed.
  void lookup_batch(int32_t field_number, const uint8_t* bytes, const int64_t* offsets, int64_t n, TermState* out, uint8_t* found,
                    TermPositions* pos_out = nullptr) const {
    const Field* f = find_field(field_number);
    if (pos_out) for (int64_t i = 0; i < n; ++i) pos_out[i] = TermPositions{0, 0, -1};
    if (!f || f->slots.empty()) {
      for (int64_t i = 0; i < n; ++i) { out[i] = TermState{0, -1, 0, 0, -1}; if (found) found[i] = 0; }
      return;
    }
    // three stages, W lookups apart: (A) hash + prefetch the slot, (B) read the slot, prefetch the entry it names,
    // (C) probe for real (slot and entry now in cache; only the term bytes may still miss)
    constexpr int W = 16;
    uint64_t h[3][W];
    const size_t mask = f->slots.size() - 1;
    auto count = [&](int64_t base) { return base < n ? std::min<int64_t>(W, n - base) : 0; };
    auto stage_a = [&](int64_t base, uint64_t* hs) {
      for (int64_t j = 0; j < count(base); ++j) {
        hs[j] = hash_bytes(bytes + offsets[base + j], (size_t)(offsets[base + j + 1] - offsets[base + j]));
        __builtin_prefetch(&f->slots[(size_t)hs[j] & mask]);
      }
    };
    auto stage_b = [&](int64_t base, const uint64_t* hs) {
      for (int64_t j = 0; j < count(base); ++j) {
        const Slot& s = f->slots[(size_t)hs[j] & mask];
        if (s.tag != 0) __builtin_prefetch(&f->entries[s.entry]);
      }
    };
    stage_a(0, h[0]);
    stage_a(W, h[1]);
    stage_b(0, h[0]);
    for (int64_t base = 0, w = 0; base < n; base += W, w = (w + 1) % 3) {
      stage_a(base + 2 * W, h[(w + 2) % 3]);
      stage_b(base + W, h[(w + 1) % 3]);
      for (int64_t j = 0; j < count(base); ++j) {
        const int64_t i = base + j;
        const int64_t e = probe(*f, h[w][j], bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &out[i]);
        if (found) found[i] = e >= 0 ? 1 : 0;
        if (e >= 0 && pos_out && !f->positions.empty()) pos_out[i] = f->positions[(size_t)e];
      }
    }
  }

  size_t memory_bytes() const {
    size_t n = 0;
    for (const Field& f : fields_)
      n += f.pool.size() + f.entries.size() * sizeof(Entry) + f.positions.size() * sizeof(TermPositions) + f.slots.size() * sizeof(Slot);
    return n;
  }

 private:
  struct Entry { uint64_t offset; uint32_t len; TermState state; };
  struct Slot { uint32_t tag; uint32_t entry; };
  struct Field {
    TermFieldInfo info;
    TermFieldStats stats{};
    std::vector<uint8_t> pool;
    std::vector<Entry> entries;
    std::vector<TermPositions> positions;  // parallel to `entries`; only for fields indexed with positions
    std::vector<Slot> slots;
  };
  std::vector<Field> fields_;

  // -> index of the term's entry, or -1 (then *out is the "absent" state)
  static int64_t probe(const Field& f, uint64_t h, const uint8_t* term, size_t len, TermState* out) {
    const size_t mask = f.slots.size() - 1;
    const uint32_t tag = (uint32_t)(h >> 32) | 1u;
    for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
      const Slot& s = f.slots[i];
      if (s.tag == 0) { *out = TermState{0, -1, 0, 0, -1}; return -1; }
      if (s.tag == tag) {
        const Entry& e = f.entries[s.entry];
        if (e.len == len && std::memcmp(f.pool.data() + e.offset, term, len) == 0) { *out = e.state; return (int64_t)s.entry; }
      }
    }
  }

  const Field* find_field(int32_t number) const {
    for (const Field& f : fields_) if (f.info.number == number) return &f;
    return nullptr;
  }

  static uint64_t vlong(detail::Cursor& c) {
    uint64_t v = 0;
    for (int i = 0; i < 9; ++i) { uint8_t b = c.u8(); v |= (uint64_t)(b & 0x7f) << (7 * i); if (!(b & 0x80)) return v; }
    c.ok = false;  // data_input.rs:131-199: a 10th byte means a negative vLong, rejected
    return v;
  }

  static uint64_t hash_bytes(const uint8_t* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) { uint64_t w; std::memcpy(&w, p, 8); h = mix(h ^ w); p += 8; n -= 8; }
    uint64_t w = 0;
    std::memcpy(&w, p, n);
    return mix(h ^ w ^ ((uint64_t)n << 56));
  }
  static uint64_t mix(uint64_t x) {
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
  }

  // Visits every block reachable from the field's root block and files every term.
  int enumerate(const uint8_t* tim, size_t blocks_start, size_t blocks_end, uint64_t root_fp, Field* f, std::string* why) {
    const int ERR_EOF = -3, ERR_CORRUPT = -4;
    struct Work { uint64_t fp; uint64_t prefix_off; uint32_t prefix_len; };  // prefix bytes live in `prefixes`
    std::vector<Work> work;
    std::vector<uint8_t> prefixes;
    work.push_back({root_fp, 0, 0});
    const bool has_freqs = f->info.index_options != 1, has_pos = f->info.index_options >= 3;
    const bool has_pay = f->info.index_options >= 4 || f->info.has_payloads;
    const int longs_size = f->stats.longs_size;
    const uint64_t num_terms = (uint64_t)f->stats.num_terms;
    if (num_terms > 0xFFFFFFF0ull) { *why = "more than 2^32 terms in one field"; return -5; }
    if (num_terms > (uint64_t)(blocks_end - blocks_start)) { *why = "num_terms exceeds what the term blocks can hold"; return ERR_CORRUPT; }
    f->entries.reserve((size_t)num_terms);
    uint64_t blocks_seen = 0, sum_doc_freq = 0;
    std::vector<uint8_t> prefix;
    while (!work.empty()) {
      const Work w = work.back();
      work.pop_back();
      prefix.assign(prefixes.begin() + w.prefix_off, prefixes.begin() + w.prefix_off + w.prefix_len);
      uint64_t fp = w.fp;
      for (;;) {  // the floor-block chain of this prefix
        if (fp < blocks_start || fp >= blocks_end) { *why = "term block pointer out of range"; return ERR_CORRUPT; }
        if (++blocks_seen > (uint64_t)(blocks_end - blocks_start)) { *why = "term block graph is not a tree"; return ERR_CORRUPT; }
        detail::Cursor c{tim, blocks_end};
        c.pos = (size_t)fp;
        const uint32_t code = c.vint();
        const uint32_t ent_count = code >> 1;
        const bool is_last_in_floor = code & 1;
        const uint32_t code2 = c.vint();
        const bool is_leaf = code2 & 1;
        const uint32_t suffix_len = code2 >> 1;
        if (!c.ok || ent_count == 0 || c.pos + suffix_len > blocks_end) { *why = "truncated term block"; return ERR_EOF; }
        detail::Cursor sfx{tim + c.pos, suffix_len};
        c.pos += suffix_len;
        const uint32_t stats_len = c.vint();
        if (!c.ok || c.pos + stats_len > blocks_end) { *why = "truncated term block"; return ERR_EOF; }
        detail::Cursor stats{tim + c.pos, stats_len};
        c.pos += stats_len;
        const uint32_t meta_len = c.vint();
        if (!c.ok || c.pos + meta_len > blocks_end) { *why = "truncated term block"; return ERR_EOF; }
        detail::Cursor meta{tim + c.pos, meta_len};
        c.pos += meta_len;
        const uint64_t fp_end = c.pos;

        int64_t doc_fp = 0;
        uint64_t pos_fp = 0, pay_fp = 0;
        bool first_term = true;
        for (uint32_t e = 0; e < ent_count; ++e) {
          const uint32_t scode = sfx.vint();
          const uint32_t slen = is_leaf ? scode : scode >> 1;
          const bool is_sub_block = !is_leaf && (scode & 1);
          if (!sfx.ok || sfx.pos + slen > sfx.len) { *why = "truncated suffix blob in a term block"; return ERR_EOF; }
          const uint8_t* sbytes = sfx.p + sfx.pos;
          sfx.pos += slen;
          if (is_sub_block) {
            const uint64_t sub_code = vlong(sfx);
            if (!sfx.ok || slen == 0 || sub_code == 0 || sub_code > fp - blocks_start) { *why = "invalid sub-block pointer"; return ERR_CORRUPT; }
            Work sub{fp - sub_code, prefixes.size(), (uint32_t)(prefix.size() + slen)};
            prefixes.insert(prefixes.end(), prefix.begin(), prefix.end());
            prefixes.insert(prefixes.end(), sbytes, sbytes + slen);
            work.push_back(sub);
            continue;
          }
          // a term: stats + metadata (decode_metadata + lucene50_decode_term), cumulative within the block
          TermState st;
          st.doc_freq = (int32_t)stats.vint();
          // vlong() yields < 2^63; sums are formed unsigned and must stay below 2^62 (a file pointer / count a real
          // index can hold), so no signed overflow whatever the bytes say
          const uint64_t kMax = 1ull << 62;
          const uint64_t ttf_extra = has_freqs ? vlong(stats) : 0;
          uint64_t longs[3] = {0, 0, 0};
          for (int k = 0; k < longs_size; ++k) longs[k] = vlong(meta);
          const uint64_t fp = (first_term ? 0ull : (uint64_t)doc_fp) + longs[0];
          if (ttf_extra >= kMax || longs[0] >= kMax || longs[1] >= kMax || longs[2] >= kMax || fp >= kMax) { *why = "term statistics / file pointer out of range in a term block"; return ERR_CORRUPT; }
          st.total_term_freq = has_freqs ? (int64_t)((uint64_t)(uint32_t)st.doc_freq + ttf_extra) : -1;
          doc_fp = (int64_t)fp;
          if (has_pos) {  // lucene50_decode_term: longs[1] -> pos_start_fp, longs[2] -> pay_start_fp (payloads / offsets only)
            pos_fp = (first_term ? 0ull : pos_fp) + longs[1];
            pay_fp = has_pay ? (first_term ? 0ull : pay_fp) + longs[2] : 0ull;
            if (pos_fp >= kMax || pay_fp >= kMax) { *why = "position file pointer out of range in a term block"; return ERR_CORRUPT; }
          }
          first_term = false;
          st.doc_start_fp = doc_fp;
          st.singleton_doc_id = st.doc_freq == 1 ? (int32_t)meta.vint() : -1;
          const uint64_t last_pos_block = (has_pos && st.total_term_freq > 128) ? vlong(meta) : 0ull;
          if (last_pos_block >= kMax) { *why = "last position block offset out of range in a term block"; return ERR_CORRUPT; }
          st.skip_offset = st.doc_freq > 128 ? (int64_t)vlong(meta) : -1;
          if (!stats.ok || !meta.ok) { *why = "truncated stats/metadata blob in a term block"; return ERR_EOF; }
          if (st.doc_freq <= 0 || (has_freqs && st.total_term_freq < st.doc_freq)) { *why = "invalid term statistics in a term block"; return ERR_CORRUPT; }
          if (f->entries.size() >= num_terms) { *why = "more terms in the blocks than the field summary declares"; return ERR_CORRUPT; }
          sum_doc_freq += (uint64_t)st.doc_freq;
          Entry ent{f->pool.size(), (uint32_t)(prefix.size() + slen), st};
          f->pool.insert(f->pool.end(), prefix.begin(), prefix.end());
          f->pool.insert(f->pool.end(), sbytes, sbytes + slen);
          f->entries.push_back(ent);
          if (has_pos) f->positions.push_back(TermPositions{(int64_t)pos_fp, (int64_t)pay_fp, st.total_term_freq > 128 ? (int64_t)last_pos_block : -1});
        }
        if (sfx.pos != sfx.len || stats.pos != stats.len || meta.pos != meta.len) { *why = "term block blobs longer than their entries"; return ERR_CORRUPT; }
        if (is_last_in_floor) break;
        fp = fp_end;
      }
    }
    if (f->entries.size() != num_terms) { *why = "fewer terms in the blocks than the field summary declares"; return ERR_CORRUPT; }
    if (sum_doc_freq != (uint64_t)f->stats.sum_doc_freq) { *why = "sum_doc_freq does not match the term blocks"; return ERR_CORRUPT; }
    // hash table: power of two, load <= 0.5
    size_t cap = 16;
    while (cap < 2 * f->entries.size()) cap <<= 1;
    f->slots.assign(cap, Slot{0, 0});
    const size_t mask = cap - 1;
    for (size_t i = 0; i < f->entries.size(); ++i) {
      const Entry& e = f->entries[i];
      const uint64_t h = hash_bytes(f->pool.data() + e.offset, e.len);
      const uint32_t tag = (uint32_t)(h >> 32) | 1u;
      size_t s = (size_t)h & mask;
      for (;; s = (s + 1) & mask) {
        if (f->slots[s].tag == 0) break;
        if (f->slots[s].tag == tag) {
          const Entry& o = f->entries[f->slots[s].entry];
          if (o.len == e.len && std::memcmp(f->pool.data() + o.offset, f->pool.data() + e.offset, e.len) == 0) {
            *why = "the same term appears twice in the term blocks";
            return ERR_CORRUPT;
          }
        }
      }
      f->slots[s] = Slot{tag, (uint32_t)i};
    }
    return 0;
  }
};

}  // namespace rucene
