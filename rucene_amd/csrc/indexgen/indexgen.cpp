// Synthetic Lucene50 segment generator (host C++): deterministic Zipfian posting lists written as real ".doc"
// bytes (IndexHeader, ForUtil table, per-term FullBlocks / VInt tail / multi-level skip data, Footer), a
// 1-byte-per-doc norms array and a flat table of BlockTermState records standing in for the block-tree term
// dictionary (out of scope). This is the *write side* of the format — an independent,
// bulk-style implementation of the byte grammar in SURVEY.md §3.7; tests/test_format.py checks it byte for
// byte against the oracle's line-faithful Lucene50PostingsWriter restatement.
//
// Format spec followed (paths relative to /root/reference/src/core):
//   codec/codec_util.rs:46-120                  index header / footer (+ CRC32)
//   codec/postings/for_util.rs:150-185,396-478  ForUtil table, block = [hdr][payload] | [0][vint]
//   util/packed/packed_simd.rs:81-108           BP128 vertical layout (version 1)
//   util/packed/packed_misc.rs:2556-2582,2757-2777  legacy Packed / PackedSingleBlock (version 0)
//   codec/postings/posting_writer.rs:289-361,477-591  term layout, VInt tail, skip offset
//   codec/postings/skip_writer.rs:187-289       skip entries: vint docDelta, vlong fpDelta (+vlong child)
// Corpus spec: SURVEY.md §8(d) (splitmix64, seed 0x527563656E65 ^ purpose tag).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rucene_gpu.h"
#include "../host/bm25_similarity.hpp"

namespace {

struct Bytes {
  std::vector<uint8_t> b;
  size_t size() const { return b.size(); }
  void u8(uint8_t v) { b.push_back(v); }
  void raw(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
  void be32(uint32_t v) { uint8_t t[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; raw(t, 4); }
  void be64(uint64_t v) { be32((uint32_t)(v >> 32)); be32((uint32_t)v); }
  void vint(uint32_t v) { while (v >= 0x80) { u8((uint8_t)(v | 0x80)); v >>= 7; } u8((uint8_t)v); }
  void vlong(uint64_t v) { while (v >= 0x80) { u8((uint8_t)(v | 0x80)); v >>= 7; } u8((uint8_t)v); }
};

uint32_t crc32_of(const uint8_t* p, size_t n) {
  static uint32_t T[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1))); T[i] = c; }
    ready = true;
  }
  uint32_t c = ~0u;
  for (size_t i = 0; i < n; i++) c = T[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  double unit_open() { return ((double)(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0); }  // (0, 1]
};

// ---- block payload encoders (by layout formula, not by the reference's macro) ----------------------------------

// BP128: value i = 4r + l sits at bits [r*b, (r+1)*b) of stream l; stream word w is dword 4w + l.
void put_bp128(const uint32_t* v, int b, Bytes& out) {
  uint32_t words[128];
  std::memset(words, 0, sizeof(uint32_t) * (size_t)(4 * b));
  for (int i = 0; i < 128; i++) {
    int r = i >> 2, l = i & 3, p = r * b, w = p >> 5, s = p & 31;
    words[4 * w + l] |= v[i] << s;
    if (s + b > 32) words[4 * (w + 1) + l] |= v[i] >> (32 - s);
  }
  out.raw(words, (size_t)(16 * b));  // little-endian dwords (x86 store in the reference)
}
// legacy Packed: one MSB-first big-endian bitstream
void put_packed(const uint32_t* v, int b, Bytes& out) {
  uint8_t buf[512];
  std::memset(buf, 0, (size_t)(16 * b));
  for (int i = 0; i < 128; i++)
    for (int k = 0; k < b; k++)
      if ((v[i] >> (b - 1 - k)) & 1) { int pos = i * b + k; buf[pos >> 3] |= (uint8_t)(0x80 >> (pos & 7)); }
  out.raw(buf, (size_t)(16 * b));
}
// legacy PackedSingleBlock (b in {1,2,4}): big-endian u64 blocks, 64/b values each, first value in the low bits
void put_psb(const uint32_t* v, int b, Bytes& out) {
  int per = 64 / b;
  for (int blk = 0; blk * per < 128; blk++) {
    uint64_t w = 0;
    for (int j = 0; j < per; j++) w |= (uint64_t)v[blk * per + j] << (j * b);
    out.be64(w);
  }
}

struct BlockCodec {
  int version;
  void put_block(const uint32_t* v, Bytes& out) const {
    bool same = true;
    uint32_t o = 0;
    for (int i = 0; i < 128; i++) { same &= (v[i] == v[0]); o |= v[i]; }
    if (same) { out.u8(0); out.vint(v[0]); return; }
    int b = 32 - __builtin_clz(o);
    out.u8((uint8_t)b);
    if (version >= 1) put_bp128(v, b, out);
    else if (b == 1 || b == 2 || b == 4) put_psb(v, b, out);
    else put_packed(v, b, out);
  }
};

int skip_levels_for(int64_t n, int cap) {  // 1 + floor(log8(n / 128)), capped
  int levels = 1;
  for (int64_t x = n / 128; x >= 8; x /= 8) levels++;
  return std::min(levels, cap);
}

}  // namespace

struct rgen_index {
  Bytes doc;
  std::vector<uint8_t> norms;
  std::vector<rgpu_term_state> terms;
  int32_t max_doc = 0;
  int32_t version = 1;
  int64_t sum_total_term_freq = 0;  // collection statistic (sum of doc lengths)
  int64_t sum_doc_freq = 0;
  int64_t total_postings = 0;
  int64_t full_blocks = 0;
  int64_t block_payload_bytes = 0;  // bytes of all FullBlocks (headers + payloads)
  int64_t tail_bytes = 0;
  int64_t skip_bytes = 0;
  std::string error;
  BlockCodec codec{1};
  int writer_skip_levels = 1;

  void begin(int32_t max_doc_, int32_t version_, const uint8_t seg_id[16]) {
    max_doc = max_doc_;
    version = version_;
    codec.version = version_;
    writer_skip_levels = skip_levels_for(max_doc_, 10);
    // IndexHeader
    doc.be32(0x3FD76C17u);
    const char* name = "Lucene50PostingsWriterDoc";
    doc.vint((uint32_t)std::strlen(name));
    doc.raw(name, std::strlen(name));
    doc.be32((uint32_t)version_);
    doc.raw(seg_id, 16);
    const char* suffix = "Lucene50_0";
    doc.u8((uint8_t)std::strlen(suffix));
    doc.raw(suffix, std::strlen(suffix));
    // ForUtil table: PackedInts version 2, then (formatId << 5 | bpv - 1) for bpv 1..32 (COMPACT ->
    // PackedSingleBlock only where it wastes no bits: 1, 2, 4)
    doc.vint(2);
    for (int bpv = 1; bpv <= 32; bpv++) {
      int fmt = (bpv == 1 || bpv == 2 || bpv == 4) ? 1 : 0;
      doc.vint((uint32_t)(fmt << 5 | (bpv - 1)));
    }
  }

  // Append one term (docs strictly increasing, freqs >= 1); returns its BlockTermState.
  rgpu_term_state add_term(const int32_t* docs, const int32_t* freqs, int64_t df) {
    rgpu_term_state st;
    st.doc_start_fp = (int64_t)doc.size();
    st.skip_offset = -1;
    st.singleton_doc_id = -1;
    st.doc_freq = (int32_t)df;
    int64_t ttf = 0;
    for (int64_t i = 0; i < df; i++) ttf += freqs[i];
    st.total_term_freq = ttf;
    total_postings += df;
    sum_doc_freq += df;
    if (df == 1) { st.singleton_doc_id = docs[0]; return st; }

    // per-level skip buffers; an entry is emitted when the first doc AFTER a full block arrives
    struct Level { Bytes buf; int32_t last_doc = 0; int64_t last_fp = 0; };
    std::vector<Level> levels;
    const int64_t nfull = df / 128;
    int32_t prev = 0;
    uint32_t dbuf[128], fbuf[128];
    for (int64_t blk = 0; blk < nfull; blk++) {
      if (blk > 0) {  // doc #(128*blk) is about to be written: record the boundary after block blk-1
        if (levels.empty()) { levels.resize((size_t)writer_skip_levels); for (auto& L : levels) L.last_fp = st.doc_start_fp; }
        const int32_t boundary_doc = docs[blk * 128 - 1];
        const int64_t boundary_fp = (int64_t)doc.size();
        int nlev = 1;
        for (int64_t e = blk; e % 8 == 0 && nlev < writer_skip_levels; e /= 8) nlev++;
        int64_t child = 0;
        for (int lv = 0; lv < nlev; lv++) {
          Level& L = levels[(size_t)lv];
          L.buf.vint((uint32_t)(boundary_doc - L.last_doc));
          L.buf.vlong((uint64_t)(boundary_fp - L.last_fp));
          L.last_doc = boundary_doc;
          L.last_fp = boundary_fp;
          int64_t here = (int64_t)L.buf.size();
          if (lv > 0) L.buf.vlong((uint64_t)child);
          child = here;
        }
      }
      for (int i = 0; i < 128; i++) {
        int32_t d = docs[blk * 128 + i];
        dbuf[i] = (uint32_t)(d - prev);
        prev = d;
        fbuf[i] = (uint32_t)freqs[blk * 128 + i];
      }
      size_t before = doc.size();
      codec.put_block(dbuf, doc);
      codec.put_block(fbuf, doc);
      block_payload_bytes += (int64_t)(doc.size() - before);
      full_blocks++;
    }
    const int64_t rem = df - nfull * 128;
    if (rem > 0 && nfull > 0) {  // the tail's first doc also closes the last full block
      if (levels.empty()) { levels.resize((size_t)writer_skip_levels); for (auto& L : levels) L.last_fp = st.doc_start_fp; }
      const int32_t boundary_doc = docs[nfull * 128 - 1];
      const int64_t boundary_fp = (int64_t)doc.size();
      int nlev = 1;
      for (int64_t e = nfull; e % 8 == 0 && nlev < writer_skip_levels; e /= 8) nlev++;
      int64_t child = 0;
      for (int lv = 0; lv < nlev; lv++) {
        Level& L = levels[(size_t)lv];
        L.buf.vint((uint32_t)(boundary_doc - L.last_doc));
        L.buf.vlong((uint64_t)(boundary_fp - L.last_fp));
        L.last_doc = boundary_doc;
        L.last_fp = boundary_fp;
        int64_t here = (int64_t)L.buf.size();
        if (lv > 0) L.buf.vlong((uint64_t)child);
        child = here;
      }
    }
    size_t tail_start = doc.size();
    for (int64_t i = nfull * 128; i < df; i++) {
      uint32_t delta = (uint32_t)(docs[i] - prev);
      prev = docs[i];
      if (freqs[i] == 1) doc.vint(delta << 1 | 1);
      else { doc.vint(delta << 1); doc.vint((uint32_t)freqs[i]); }
    }
    tail_bytes += (int64_t)(doc.size() - tail_start);
    if (df > 128) {
      st.skip_offset = (int64_t)doc.size() - st.doc_start_fp;
      size_t skip_start = doc.size();
      for (int lv = (int)levels.size() - 1; lv >= 1; lv--) {
        if (levels[(size_t)lv].buf.size() > 0) {
          doc.vlong(levels[(size_t)lv].buf.size());
          doc.raw(levels[(size_t)lv].buf.b.data(), levels[(size_t)lv].buf.size());
        }
      }
      if (!levels.empty()) doc.raw(levels[0].buf.b.data(), levels[0].buf.size());
      skip_bytes += (int64_t)(doc.size() - skip_start);
    }
    return st;
  }

  void finish() {
    doc.be32(~0x3FD76C17u);
    doc.be32(0);
    doc.be64((uint64_t)crc32_of(doc.b.data(), doc.size()));
    if (has_positions) {
      pos.be32(~0x3FD76C17u);
      pos.be32(0);
      pos.be64((uint64_t)crc32_of(pos.b.data(), pos.size()));
    }
    if (has_pay()) {
      pay.be32(~0x3FD76C17u);
      pay.be32(0);
      pay.be64((uint64_t)crc32_of(pay.b.data(), pay.size()));
    }
  }

  // ---- positions (IndexOptions::DocsAndFreqsAndPositions, no payloads / offsets) ---------------------------------------
  // ".pos" = index header "Lucene50PostingsWriterPos" + per term: floor(ttf / 128) ForUtil blocks of position deltas (the
  // delta restarts at every doc) + (ttf % 128) vints (posting_writer.rs:363-455, 505-560). The ".doc" skip entries of such a
  // field carry two more values: the .pos pointer and the number of buffered positions at the block boundary
  // (skip_writer.rs:261-289). Synthetic data for the positions / phrase row.
  // A field that also stores OFFSETS (IndexOptions::DocsAndFreqsAndPositionsAndOffsets) and / or PAYLOADS
  // (FieldInfo::has_store_payloads) gets a third file, ".pay": per 128 positions of a term
  // [payload-length block][vint byte count][the payload bytes] (payloads) then [offset start-delta block][offset length block]
  // (offsets) (posting_writer.rs:404-451); the trailing (ttf % 128) positions keep theirs inside ".pos", woven into the VInts
  // (:505-560); every skip entry grows by the payload bytes buffered at the boundary (payloads) and the .pay pointer
  // (skip_writer.rs:276-286).
  Bytes pos, pay;
  bool has_positions = false, has_offsets = false, has_payloads = false;
  std::vector<int64_t> term_pos_start, term_last_pos_block_offset, term_pay_start;
  bool has_pay() const { return has_offsets || has_payloads; }

  static void index_header(Bytes& out, const char* name, int32_t version, const uint8_t seg_id[16]) {
    out.be32(0x3FD76C17u);
    out.vint((uint32_t)std::strlen(name));
    out.raw(name, std::strlen(name));
    out.be32((uint32_t)version);
    out.raw(seg_id, 16);
    const char* suffix = "Lucene50_0";
    out.u8((uint8_t)std::strlen(suffix));
    out.raw(suffix, std::strlen(suffix));
  }
  void begin_positions(const uint8_t seg_id[16], bool offsets = false, bool payloads = false) {
    has_positions = true;
    has_offsets = offsets;
    has_payloads = payloads;
    index_header(pos, "Lucene50PostingsWriterPos", version, seg_id);
    if (has_pay()) index_header(pay, "Lucene50PostingsWriterPay", version, seg_id);
  }

  // positions: flat, doc j of the term owns freqs[j] of them (ascending within a doc). With offsets: start / end per position
  // (starts non-decreasing within a doc); with payloads: position q's bytes = payload_bytes[payload_offs[q], payload_offs[q + 1]).
  rgpu_term_state add_term_positions(const int32_t* docs, const int32_t* freqs, const int32_t* positions, int64_t df,
                                     const int32_t* starts = nullptr, const int32_t* ends = nullptr, const int64_t* payload_offs = nullptr,
                                     const uint8_t* payload_bytes = nullptr) {
    rgpu_term_state st;
    st.doc_start_fp = (int64_t)doc.size();
    st.skip_offset = -1;
    st.singleton_doc_id = -1;
    st.doc_freq = (int32_t)df;
    const int64_t pos_start = (int64_t)pos.size();
    int64_t ttf = 0;
    for (int64_t i = 0; i < df; i++) ttf += freqs[i];
    st.total_term_freq = ttf;
    total_postings += df;
    sum_doc_freq += df;
    // .pos first: deltas of the whole term, cut into 128-blocks; remember where each block ends
    std::vector<uint32_t> deltas((size_t)ttf);
    {
      int64_t at = 0;
      for (int64_t j = 0; j < df; j++) {
        int32_t last = 0;
        for (int32_t q = 0; q < freqs[j]; q++, at++) { deltas[(size_t)at] = (uint32_t)(positions[at] - last); last = positions[at]; }
      }
    }
    // per position: payload length, offset start delta (restarts at every doc) and offset length
    std::vector<uint32_t> plen, odelta, olen;
    if (has_payloads) {
      plen.resize((size_t)ttf);
      for (int64_t i = 0; i < ttf; i++) plen[(size_t)i] = (uint32_t)(payload_offs[i + 1] - payload_offs[i]);
    }
    if (has_offsets) {
      odelta.resize((size_t)ttf);
      olen.resize((size_t)ttf);
      int64_t at = 0;
      for (int64_t j = 0; j < df; j++) {
        int32_t last = 0;
        for (int32_t q = 0; q < freqs[j]; q++, at++) {
          odelta[(size_t)at] = (uint32_t)(starts[at] - last);
          olen[(size_t)at] = (uint32_t)(ends[at] - starts[at]);
          last = starts[at];
        }
      }
    }
    const int64_t n_pos_blocks = ttf / 128;
    const int64_t pay_start = (int64_t)pay.size();
    std::vector<int64_t> pos_fp_after((size_t)n_pos_blocks + 1, pos_start);  // [b] = .pos pointer once b blocks are written
    std::vector<int64_t> pay_fp_after((size_t)n_pos_blocks + 1, pay_start);  // the same for .pay
    for (int64_t b = 0; b < n_pos_blocks; b++) {
      codec.put_block(deltas.data() + b * 128, pos);
      pos_fp_after[(size_t)b + 1] = (int64_t)pos.size();
      if (has_payloads) {
        codec.put_block(plen.data() + b * 128, pay);
        const int64_t from = payload_offs[b * 128], to = payload_offs[b * 128 + 128];
        pay.vint((uint32_t)(to - from));
        pay.raw(payload_bytes + from, (size_t)(to - from));
      }
      if (has_offsets) {
        codec.put_block(odelta.data() + b * 128, pay);
        codec.put_block(olen.data() + b * 128, pay);
      }
      pay_fp_after[(size_t)b + 1] = (int64_t)pay.size();
    }
    term_pos_start.push_back(pos_start);
    term_pay_start.push_back(has_pay() ? pay_start : 0);
    term_last_pos_block_offset.push_back(ttf > 128 ? (int64_t)pos.size() - pos_start : -1);
    {
      // the trailing positions: a payload length / an offset length is written only where it differs from the one before
      int64_t last_plen = -1, last_olen = -1;
      for (int64_t i = n_pos_blocks * 128; i < ttf; i++) {
        if (has_payloads) {
          if ((int64_t)plen[(size_t)i] != last_plen) {
            last_plen = plen[(size_t)i];
            pos.vint(deltas[(size_t)i] << 1 | 1u);
            pos.vint(plen[(size_t)i]);
          } else {
            pos.vint(deltas[(size_t)i] << 1);
          }
          if (plen[(size_t)i]) pos.raw(payload_bytes + payload_offs[i], plen[(size_t)i]);
        } else {
          pos.vint(deltas[(size_t)i]);
        }
        if (has_offsets) {
          if ((int64_t)olen[(size_t)i] == last_olen) {
            pos.vint(odelta[(size_t)i] << 1);
          } else {
            pos.vint(odelta[(size_t)i] << 1 | 1u);
            pos.vint(olen[(size_t)i]);
            last_olen = olen[(size_t)i];
          }
        }
      }
    }
    if (df == 1) { st.singleton_doc_id = docs[0]; return st; }

    struct Level { Bytes buf; int32_t last_doc = 0; int64_t last_fp = 0, last_pos_fp = 0, last_pay_fp = 0; };
    std::vector<Level> levels;
    const int64_t nfull = df / 128;
    int64_t freq_sum = 0;  // positions of the docs written so far
    auto boundary = [&](int64_t blocks_done) {  // the first doc after `blocks_done` full doc blocks is about to be written
      if (levels.empty()) {
        levels.resize((size_t)writer_skip_levels);
        for (auto& L : levels) { L.last_fp = st.doc_start_fp; L.last_pos_fp = pos_start; L.last_pay_fp = pay_start; }
      }
      const int32_t boundary_doc = docs[blocks_done * 128 - 1];
      const int64_t boundary_fp = (int64_t)doc.size();
      const int64_t boundary_pos_fp = pos_fp_after[(size_t)(freq_sum / 128)];
      const int64_t boundary_pay_fp = pay_fp_after[(size_t)(freq_sum / 128)];
      const uint32_t pos_buffer_upto = (uint32_t)(freq_sum % 128);
      // payload bytes of the positions buffered since the last full position block
      const uint32_t payload_byte_upto = has_payloads ? (uint32_t)(payload_offs[freq_sum] - payload_offs[freq_sum - freq_sum % 128]) : 0u;
      int nlev = 1;
      for (int64_t e = blocks_done; e % 8 == 0 && nlev < writer_skip_levels; e /= 8) nlev++;
      int64_t child = 0;
      for (int lv = 0; lv < nlev; lv++) {
        Level& L = levels[(size_t)lv];
        L.buf.vint((uint32_t)(boundary_doc - L.last_doc));
        L.buf.vlong((uint64_t)(boundary_fp - L.last_fp));
        L.buf.vlong((uint64_t)(boundary_pos_fp - L.last_pos_fp));
        L.buf.vint(pos_buffer_upto);
        if (has_payloads) L.buf.vint(payload_byte_upto);
        if (has_pay()) { L.buf.vlong((uint64_t)(boundary_pay_fp - L.last_pay_fp)); L.last_pay_fp = boundary_pay_fp; }
        L.last_doc = boundary_doc;
        L.last_fp = boundary_fp;
        L.last_pos_fp = boundary_pos_fp;
        const int64_t here = (int64_t)L.buf.size();
        if (lv > 0) L.buf.vlong((uint64_t)child);
        child = here;
      }
    };
    int32_t prev = 0;
    uint32_t dbuf[128], fbuf[128];
    for (int64_t blk = 0; blk < nfull; blk++) {
      if (blk > 0) boundary(blk);
      for (int i = 0; i < 128; i++) {
        const int32_t d = docs[blk * 128 + i];
        dbuf[i] = (uint32_t)(d - prev);
        prev = d;
        fbuf[i] = (uint32_t)freqs[blk * 128 + i];
        freq_sum += freqs[blk * 128 + i];
      }
      const size_t before = doc.size();
      codec.put_block(dbuf, doc);
      codec.put_block(fbuf, doc);
      block_payload_bytes += (int64_t)(doc.size() - before);
      full_blocks++;
    }
    if (df - nfull * 128 > 0 && nfull > 0) boundary(nfull);
    const size_t tail_start = doc.size();
    for (int64_t i = nfull * 128; i < df; i++) {
      const uint32_t delta = (uint32_t)(docs[i] - prev);
      prev = docs[i];
      if (freqs[i] == 1) doc.vint(delta << 1 | 1);
      else { doc.vint(delta << 1); doc.vint((uint32_t)freqs[i]); }
    }
    tail_bytes += (int64_t)(doc.size() - tail_start);
    if (df > 128) {
      st.skip_offset = (int64_t)doc.size() - st.doc_start_fp;
      const size_t skip_start = doc.size();
      for (int lv = (int)levels.size() - 1; lv >= 1; lv--) {
        if (levels[(size_t)lv].buf.size() > 0) {
          doc.vlong(levels[(size_t)lv].buf.size());
          doc.raw(levels[(size_t)lv].buf.b.data(), levels[(size_t)lv].buf.size());
        }
      }
      if (!levels.empty()) doc.raw(levels[0].buf.b.data(), levels[0].buf.size());
      skip_bytes += (int64_t)(doc.size() - skip_start);
    }
    return st;
  }
};

extern "C" {

typedef struct rgen_config {
  int32_t max_doc;       // N
  int32_t version;       // 1 = BP128 (live format), 0 = legacy PackedInts
  int64_t n_terms;       // V
  double zipf_scale;     // df(r) = clamp(round(zipf_scale * N / r), 1, N / 2), r = 1..V
  uint64_t seed;         // 0 -> 0x527563656E65 ("Rucene")
  int32_t shard;         // shard ordinal mixed into every stream (multi-GPU: one segment per shard)
  int32_t positions;     // 1: the field is indexed with positions (IndexOptions::DocsAndFreqsAndPositions): ".pos" + position pointers
                         // in the skip entries; a posting's `freq` positions start at 0..63 and step by 1..16
} rgen_config;

// Zipfian corpus of SURVEY.md §8(d).
rgen_index* rgen_build_zipf(const rgen_config* cfg) {
  rgen_index* ix = new rgen_index();
  const int32_t N = cfg->max_doc;
  const uint64_t seed = (cfg->seed ? cfg->seed : 0x527563656E65ULL) ^ ((uint64_t)cfg->shard * 0xD1B54A32D192ED03ULL);
  uint8_t seg_id[16];
  { SplitMix64 r(seed ^ 0x1D); for (int i = 0; i < 16; i += 8) { uint64_t x = r.next(); std::memcpy(seg_id + i, &x, 8); } }
  ix->begin(N, cfg->version, seg_id);
  if (cfg->positions) ix->begin_positions(seg_id);

  // norms: doc length ~ LogNormal(ln 100, 0.5) clamped to [1, 10000] -> float_to_byte315(1/sqrt(len))
  ix->norms.resize((size_t)N);
  {
    SplitMix64 r(seed ^ 0x4E4F524DULL /* "NORM" */);
    const double kPi = 3.14159265358979323846;
    int64_t sum_len = 0;
    for (int32_t d = 0; d < N; d += 2) {
      double u1 = r.unit_open(), u2 = r.unit_open();
      double rad = std::sqrt(-2.0 * std::log(u1));
      double z[2] = {rad * std::cos(2.0 * kPi * u2), rad * std::sin(2.0 * kPi * u2)};
      for (int j = 0; j < 2 && d + j < N; j++) {
        double len = std::nearbyint(std::exp(std::log(100.0) + 0.5 * z[j]));
        int32_t l = (int32_t)std::min(10000.0, std::max(1.0, len));
        sum_len += l;
        ix->norms[(size_t)(d + j)] = rucene::BM25Similarity::encode_norm_value(1.0f, l);
      }
    }
    ix->sum_total_term_freq = sum_len;
  }

  ix->terms.resize((size_t)cfg->n_terms);
  std::vector<int32_t> docs, freqs, positions;
  for (int64_t r = 1; r <= cfg->n_terms; r++) {
    double want = std::nearbyint(cfg->zipf_scale * (double)N / (double)r);
    int64_t df_nom = (int64_t)std::min((double)(N / 2), std::max(1.0, want));
    SplitMix64 rng(seed ^ (0x5445524DULL /* "TERM" */ + (uint64_t)r * 0x9E3779B97F4A7C15ULL));
    const double p = (double)df_nom / (double)N;
    const double inv_log1mp = 1.0 / std::log1p(-p);
    docs.clear();
    freqs.clear();
    int64_t d = -1;
    while ((int64_t)docs.size() < df_nom) {
      double g = 1.0 + std::floor(std::log(rng.unit_open()) * inv_log1mp);  // Geometric(p), >= 1
      if (g > (double)N) break;
      d += (int64_t)g;
      if (d >= N) break;
      docs.push_back((int32_t)d);
      uint64_t bits = rng.next();
      int f = 1 + (bits ? __builtin_ctzll(bits) : 64);  // 1 + Geometric(0.5)
      freqs.push_back(std::min(10, f));                  // write-time clamp, postings/mod.rs:82
    }
    if (docs.empty()) { docs.push_back((int32_t)(rng.next() % (uint64_t)N)); freqs.push_back(1); }
    if (cfg->positions) {
      SplitMix64 prng(seed ^ (0x504F53ULL /* "POS" */ + (uint64_t)r * 0xD6E8FEB86659FD93ULL));
      positions.clear();
      for (size_t j = 0; j < docs.size(); j++) {
        uint64_t bits = prng.next();
        int32_t at = (int32_t)(bits & 63u);
        bits >>= 6;
        for (int32_t q = 0; q < freqs[j]; q++) {
          positions.push_back(at);
          at += 1 + (int32_t)(bits & 15u);
          bits >>= 4;
        }
      }
      ix->terms[(size_t)(r - 1)] = ix->add_term_positions(docs.data(), freqs.data(), positions.data(), (int64_t)docs.size());
    } else {
      ix->terms[(size_t)(r - 1)] = ix->add_term(docs.data(), freqs.data(), (int64_t)docs.size());
    }
  }
  ix->finish();
  return ix;
}

// Explicit postings (tests): term t owns docs/freqs[offsets[t] .. offsets[t+1]).
rgen_index* rgen_build_explicit(int32_t max_doc, int32_t version, int64_t n_terms, const int64_t* offsets,
                                const int32_t* docs, const int32_t* freqs, const uint8_t* norms_or_null,
                                const uint8_t* seg_id16_or_null) {
  rgen_index* ix = new rgen_index();
  uint8_t seg_id[16];
  for (int i = 0; i < 16; i++) seg_id[i] = seg_id16_or_null ? seg_id16_or_null[i] : (uint8_t)i;
  ix->begin(max_doc, version, seg_id);
  ix->norms.assign((size_t)max_doc, 0);
  if (norms_or_null) std::memcpy(ix->norms.data(), norms_or_null, (size_t)max_doc);
  ix->terms.resize((size_t)n_terms);
  for (int64_t t = 0; t < n_terms; t++) {
    int64_t df = offsets[t + 1] - offsets[t];
    if (df <= 0) { rgpu_term_state z; std::memset(&z, 0, sizeof z); z.skip_offset = -1; z.singleton_doc_id = -1; ix->terms[(size_t)t] = z; continue; }
    ix->terms[(size_t)t] = ix->add_term(docs + offsets[t], freqs + offsets[t], df);
  }
  ix->finish();
  return ix;
}

// Explicit postings WITH positions (a DocsAndFreqsAndPositions field): doc slot j (flat over all terms) owns positions
// [pos_offsets[j], pos_offsets[j+1]) and freqs[j] must equal that count. Produces ".doc" (skip entries with position pointers)
// and ".pos"; rgen_pos_* below expose the extra bytes and per-term pointers.
rgen_index* rgen_build_explicit_positions(int32_t max_doc, int32_t version, int64_t n_terms, const int64_t* offsets, const int32_t* docs,
                                          const int32_t* freqs, const int64_t* pos_offsets, const int32_t* positions,
                                          const uint8_t* norms_or_null, const uint8_t* seg_id16_or_null) {
  rgen_index* ix = new rgen_index();
  uint8_t seg_id[16];
  for (int i = 0; i < 16; i++) seg_id[i] = seg_id16_or_null ? seg_id16_or_null[i] : (uint8_t)i;
  ix->begin(max_doc, version, seg_id);
  ix->begin_positions(seg_id);
  ix->norms.assign((size_t)max_doc, 0);
  if (norms_or_null) std::memcpy(ix->norms.data(), norms_or_null, (size_t)max_doc);
  ix->terms.resize((size_t)n_terms);
  for (int64_t t = 0; t < n_terms; t++) {
    const int64_t df = offsets[t + 1] - offsets[t];
    if (df <= 0) {
      rgpu_term_state z;
      std::memset(&z, 0, sizeof z);
      z.skip_offset = -1;
      z.singleton_doc_id = -1;
      ix->terms[(size_t)t] = z;
      ix->term_pos_start.push_back((int64_t)ix->pos.size());
      ix->term_pay_start.push_back(0);
      ix->term_last_pos_block_offset.push_back(-1);
      continue;
    }
    for (int64_t j = offsets[t]; j < offsets[t + 1]; j++)
      if (pos_offsets[j + 1] - pos_offsets[j] != freqs[j]) { ix->error = "positions per doc must equal freq"; return ix; }
    ix->terms[(size_t)t] = ix->add_term_positions(docs + offsets[t], freqs + offsets[t], positions + pos_offsets[offsets[t]], df);
  }
  ix->finish();
  return ix;
}
// The same for a field that also stores offsets (field_flags bit 0: starts / ends per position) and / or payloads (bit 1:
// position q's payload = payload_bytes[payload_offs[q], payload_offs[q + 1])): ".doc", ".pos" and ".pay".
rgen_index* rgen_build_explicit_positions_ex(int32_t max_doc, int32_t version, int64_t n_terms, const int64_t* offsets, const int32_t* docs,
                                             const int32_t* freqs, const int64_t* pos_offsets, const int32_t* positions, int32_t field_flags,
                                             const int32_t* starts, const int32_t* ends, const int64_t* payload_offs, const uint8_t* payload_bytes,
                                             const uint8_t* norms_or_null, const uint8_t* seg_id16_or_null) {
  rgen_index* ix = new rgen_index();
  uint8_t seg_id[16];
  for (int i = 0; i < 16; i++) seg_id[i] = seg_id16_or_null ? seg_id16_or_null[i] : (uint8_t)i;
  const bool offs = (field_flags & 1) != 0, pays = (field_flags & 2) != 0;
  ix->begin(max_doc, version, seg_id);
  ix->begin_positions(seg_id, offs, pays);
  ix->norms.assign((size_t)max_doc, 0);
  if (norms_or_null) std::memcpy(ix->norms.data(), norms_or_null, (size_t)max_doc);
  ix->terms.resize((size_t)n_terms);
  for (int64_t t = 0; t < n_terms; t++) {
    const int64_t df = offsets[t + 1] - offsets[t];
    if (df <= 0) {
      rgpu_term_state z;
      std::memset(&z, 0, sizeof z);
      z.skip_offset = -1;
      z.singleton_doc_id = -1;
      ix->terms[(size_t)t] = z;
      ix->term_pos_start.push_back((int64_t)ix->pos.size());
      ix->term_pay_start.push_back(ix->has_pay() ? (int64_t)ix->pay.size() : 0);
      ix->term_last_pos_block_offset.push_back(-1);
      continue;
    }
    for (int64_t j = offsets[t]; j < offsets[t + 1]; j++)
      if (pos_offsets[j + 1] - pos_offsets[j] != freqs[j]) { ix->error = "positions per doc must equal freq"; return ix; }
    const int64_t p0 = pos_offsets[offsets[t]];
    ix->terms[(size_t)t] = ix->add_term_positions(docs + offsets[t], freqs + offsets[t], positions + p0, df, offs ? starts + p0 : nullptr,
                                                  offs ? ends + p0 : nullptr, pays ? payload_offs + p0 : nullptr, payload_bytes);
  }
  ix->finish();
  return ix;
}
int64_t rgen_pay_len(const rgen_index* ix) { return (int64_t)ix->pay.size(); }
const uint8_t* rgen_pay_bytes(const rgen_index* ix) { return ix->pay.b.data(); }
const int64_t* rgen_pay_start_fps(const rgen_index* ix) { return ix->term_pay_start.data(); }
int64_t rgen_pos_len(const rgen_index* ix) { return (int64_t)ix->pos.size(); }
const uint8_t* rgen_pos_bytes(const rgen_index* ix) { return ix->pos.b.data(); }
const int64_t* rgen_pos_start_fps(const rgen_index* ix) { return ix->term_pos_start.data(); }
const int64_t* rgen_last_pos_block_offsets(const rgen_index* ix) { return ix->term_last_pos_block_offset.data(); }
const char* rgen_error(const rgen_index* ix) { return ix->error.c_str(); }

void rgen_free(rgen_index* ix) { delete ix; }
int64_t rgen_doc_len(const rgen_index* ix) { return (int64_t)ix->doc.size(); }
const uint8_t* rgen_doc_bytes(const rgen_index* ix) { return ix->doc.b.data(); }
const uint8_t* rgen_norms(const rgen_index* ix) { return ix->norms.data(); }
const rgpu_term_state* rgen_terms(const rgen_index* ix) { return ix->terms.data(); }
int64_t rgen_n_terms(const rgen_index* ix) { return (int64_t)ix->terms.size(); }
int32_t rgen_max_doc(const rgen_index* ix) { return ix->max_doc; }
// stats[0..7] = sum_total_term_freq, sum_doc_freq, total_postings, full_blocks, block_bytes, tail_bytes, skip_bytes, doc_len
void rgen_stats(const rgen_index* ix, int64_t* stats) {
  stats[0] = ix->sum_total_term_freq; stats[1] = ix->sum_doc_freq; stats[2] = ix->total_postings; stats[3] = ix->full_blocks;
  stats[4] = ix->block_payload_bytes; stats[5] = ix->tail_bytes; stats[6] = ix->skip_bytes; stats[7] = (int64_t)ix->doc.size();
}

}  // extern "C"
