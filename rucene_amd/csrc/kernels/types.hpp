// Device-visible plain structs shared by the host side of the C ABI and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace rgpu {

// One segment's device-resident data (passed by value to kernels).
struct SegView {
  const uint8_t* doc;        // raw .doc bytes (+ zero padding so speculative 16-byte row loads never fault)
  const uint8_t* norms;      // 1 byte per doc, or null (compute_score falls back to k1)
  const uint64_t* live;      // FixedBitSet words or null (MatchAllBits)
  const int32_t* dir_last;   // block directory: last doc id of block i            (skip level 0 docs, summed)
  const uint32_t* dir_off;   // block directory: byte offset of block i from the term's doc_start_fp (slot nblocks: the tail)
  const uint32_t* dir_row;   // block directory: first 16-byte row of block i in the block store, from the term's bs_base
  const uint16_t* dir_hdr;   // block directory: b_doc | vint_len << 6 | b_freq << 9
  // Block directory: the block's (freq, norm rank) frontier — bits 0..3 = its largest freq (15: some freq > 10, "no
  // bound"), then for freq f = 1..10 six bits with the largest norm RANK among the block's postings of that freq (0 when
  // there is none). With a similarity whose score grows with the norm rank (BM25: shorter doc, larger norm byte) the
  // block's best possible score under ANY clause weight is max over f <= largest freq of table[rank_f][f]: the TERM
  // kernel skips a block outright when that bound cannot enter the top-k (search_term.hpp). Built by k_prepare_blocks.
  const uint64_t* dir_bmax;
  // One level up (round 6): the same word for every WHOLE chunk of 64 consecutive blocks of a term (blocks 64 j .. 64 j + 63 counted
  // from the term's first) — the field-wise maximum of the chunk's frontier words, 15 if any of them is. Chunk j of a term whose
  // directory starts at slot `dir_base` sits at dir_sum[(dir_base + 63) / 64 + j]: a term owns nblocks + 1 consecutive slots, so
  // these ranges never overlap and no term needs a second base. A term's last, partial chunk has no word. Built by
  // k_chunk_frontiers right behind k_prepare_norms; read by k_search_term (one lane per chunk).
  const uint64_t* dir_sum;
  // Block store: the FullBlock payloads of every prepared term, copied once (k_prepare_terms) to 16-byte aligned
  // rows: block i = [max(b_doc,1) doc rows][max(b_freq,1) freq rows]; a row is 16 bytes of the BP128 / packed
  // stream exactly as in the .doc file, an all-equal stream (b == 0) is one row holding its value as a u32.
  // In the file a payload starts right after a 1-byte header, and byte-misaligned 16-byte loads run at a third
  // of the aligned rate on gfx950 (scripts/microbench/unaligned_rows.hip) — the scoring kernels were bound by it.
  const uint8_t* bstore;
  const float* sim_tables;   // n x 257 floats: cache[256] then k1
  // Norm ranks: when a segment uses <= 64 distinct norm bytes (SmallFloat lengths: the usual case) the HBM copy of
  // the norms holds each byte's RANK among the used values and rank_to_norm maps back, so a clause's whole
  // (norm, freq <= 10) score table fits in a wave's LDS slice. n_norm_ranks == 0 means raw norm bytes.
  const uint8_t* rank_to_norm;
  // Posting-order norms: for every FullBlock posting of a prepared term, the norm byte (or rank) of its doc, laid
  // out in posting order (built once per term by k_prepare_terms). Turns the per-posting norm gather — 128
  // cache-line lookups per block, the measured bottleneck of the scoring kernels — into one coalesced 128-byte
  // read per block. Null when the segment has no norms.
  const uint8_t* pnorm;
  int32_t n_norm_ranks;
  int32_t max_doc;
  int32_t doc_base;
  // 0: the field was indexed with IndexOptions::Docs — no freq block follows a doc block and tail VInts are plain
  // deltas; every freq is 1 (posting_reader.rs:532-557: the freq buffer is filled with 1). The block store then carries
  // one synthetic all-equal freq row (value 1) per block, so every block decoder works unchanged.
  int32_t has_freqs;
  // Positions fields (IndexOptions::DocsAndFreqsAndPositions): the raw .pos bytes and, per directory slot i (block i of
  // a term), where the position stream stands when block i starts — low word: byte offset from the term's
  // pos_start_fp of the position block that holds the block's first position, high word: positions of EARLIER docs
  // buffered in that block (the skip entry's posFP / posBufferUpto, skip_writer.rs:187-205; slot 0 = {0, 0}).
  const uint8_t* pos;
  const uint64_t* dir_pos;
  // bit 0: the field stores payloads, bit 1: offsets — the trailing VInt position block of a term then carries payload
  // bytes / offset words between its position deltas (decode.hpp decode_vint_block_everything)
  int32_t pos_tail_flags;
  int32_t pad_;
  // Block-max sketches of the long terms single-term queries have named (search_term.hpp k_term_sketch): per sketch
  // TERM_SKETCH_K entries `largest freq | norm rank << 4` — the (freq, rank) of a real posting of each of the term's K best blocks,
  // best first (0: no such block). DevTerm::sketch points into it.
  const uint16_t* sketch;
};

// One term as the kernels see it (built on the host from rgpu_term_state + the directory cache).
struct DevTerm {
  uint64_t start_fp;      // doc_start_fp
  uint64_t pn_base;       // first byte of this term's posting-order norms in SegView::pnorm
  uint64_t bs_base;       // first byte of this term's rows in SegView::bstore (16-byte aligned)
  uint32_t dir_base;      // first directory slot (nblocks + 1 slots)
  int32_t nblocks;        // full 128-posting blocks
  int32_t df;
  int32_t tail_n;         // df % 128 when df > 1, else 0
  int32_t singleton_doc;  // df == 1
  int32_t singleton_freq;
  float weight;           // idf * boost
  int32_t sim_table;
  uint32_t flags;         // bit 0: the sim table's norm cache is non-increasing in the norm byte (block-max bounds hold)
  uint32_t sketch;        // 1 + index of the term's block-max sketch in SegView::sketch (TERM_SKETCH_K entries each); 0: none
};
constexpr uint32_t TERM_FLAG_MONOTONE = 1u;
constexpr uint32_t TERM_FLAG_OR_DENSE = 2u;  // OR: this clause's FullBlocks are decoded inside the window kernel; only its tail runs through k_score_terms

// Work description of one term for the skip-decode ("prepare") kernel.
struct PrepTerm {
  uint64_t start_fp;
  uint64_t pn_base;    // where this term's posting-order norms go
  uint64_t bs_base;    // where this term's block-store rows go (bytes, 16-byte aligned)
  int64_t skip_fp;     // absolute, -1 when df <= 128
  uint32_t dir_base;
  int32_t nblocks;
  int32_t n_entries;   // level-0 skip entries = ceil(df / 128) - 1
  int32_t n_levels;    // 1 + floor(log8(trim(df) / 128)), capped at 10
  int32_t df;
  uint32_t bs_rows;    // rows reserved for this term in the block store
  int64_t out_base;    // >= 0: a decode is waiting for this term — k_prepare_blocks, which unpacks every block once anyway (to
                       // validate it), also leaves the postings at docs_out / freqs_out + out_base; -1: nobody is
};

// One phrase clause's position-stream pointers (parallel to the DevTerm array of a phrase launch).
struct PosTerm {
  uint64_t pos_start_fp;       // where the term's positions start in .pos
  int64_t last_pos_block_fp;   // absolute fp of the trailing VInt block; -1: none (total_term_freq == 128)
  int64_t total_term_freq;
  int32_t phrase_pos;          // the term's position inside the phrase (PhraseQuery::build: 0, 1, 2, ...)
  int32_t query_ord;           // the term's index in the QUERY's term list (device clauses are in cost order): PhrasePositions::ord
  int32_t same_as;             // query-order index of the first term of the phrase that is this very term (itself when none before)
  int32_t pad;
};

// A clause's doc bitmap (doc_bitmap.hpp), parallel to the DevTerm array of a conjunction launch: words == null = the clause is
// walked through its block directory
struct TermBitmap {
  const uint2* words;     // {any, hi} per 32 docs
  const uint32_t* ranks;  // postings before each word
  const uint8_t* freqs;   // min(freq, 255) by posting index
  const uint32_t* ovf;    // {posting index, freq} of the freqs >= 255
  const uint32_t* nib;    // four bits per doc (0 absent, 1..14 the freq, 15 look it up), or null
  const uint32_t* memb;   // one bit per doc: membership alone (k_search_and's batched first probe)
  int32_t n_ovf;
  int32_t pad;
};

struct DevQuery {
  int32_t op;
  int32_t n_terms;     // positive clauses (MUST or SHOULD) that exist in this leaf
  int32_t first_term;
  int32_t pad;         // n_not: MUST_NOT clauses, stored right after the positive ones
};

}  // namespace rgpu
