/*
 * rucene_gpu.h — C ABI of the MI355X (gfx950) query-evaluation path for Rucene.
 *
 * This is the drop-in boundary: a Rust shim inside Rucene binds exactly these symbols over `extern "C"`
 *  to replace, for Lucene50 docs+freqs postings,
 *   core::codec::postings::{ForUtil, Lucene50PostingsReader (BlockDocIterator), Lucene50SkipReader}
 *   core::search::{TermScorer, ConjunctionScorer, DisjunctionSumScorer, BM25 SimScorer, TopDocsCollector}.
 * The reference has no FFI of its own; each entry point cites the reference interface it
 * stands in for. Paths are relative to /root/reference/src/core.
 *
 * Conventions (mirroring src/error.rs and the trait surface):
 *   - plain C, opaque handles, caller-owned output buffers, no exceptions/panics across the boundary;
 *   - every fallible call returns 0 or a negative rgpu_status that maps 1:1 onto error.rs ErrorKind;
 *   - doc ids are int32 per segment (DocId), NO_MORE_DOCS = INT32_MAX; hits carry doc + doc_base;
 *   - one rgpu_ctx per process per GPU; calls on one ctx are serialised internally (safe from many
 *     threads, like &self methods of IndexSearcher);
 *   - there is NO CPU fallback: if no gfx950 device is present rgpu_init fails with RGPU_ERR_RUNTIME.
 */
#ifndef RUCENE_GPU_H
#define RUCENE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGPU_ABI_VERSION 6
#define RGPU_NO_MORE_DOCS 0x7fffffff /* search/mod.rs:59 */
#define RGPU_BLOCK_SIZE 128          /* codec/postings/posting_format.rs:36 */
#define RGPU_MAX_QUERY_TERMS 64  /* clauses of one query, MUST + SHOULD + MUST_NOT together: a clause's cursor lives in a lane of the
                                   wavefront. Disjunctions of 10..16 SHOULD clauses take the fixed-point kernels, longer ones the
                                   clause-order kernel */
#define RGPU_MAX_PHRASE_TERMS 16 /* terms of one phrase */
#define RGPU_MAX_K 1024   /* k above 128 costs ceil(k / 128) passes of the search (phrases: of the collector only); rescoring: k <= 128 */

/* error.rs:24-91 ErrorKind */
typedef enum rgpu_status {
  RGPU_OK = 0,
  RGPU_ERR_ILLEGAL_STATE = -1,     /* ErrorKind::IllegalState */
  RGPU_ERR_ILLEGAL_ARGUMENT = -2,  /* ErrorKind::IllegalArgument */
  RGPU_ERR_UNEXPECTED_EOF = -3,    /* ErrorKind::UnexpectedEOF */
  RGPU_ERR_CORRUPT_INDEX = -4,     /* ErrorKind::CorruptIndex */
  RGPU_ERR_UNSUPPORTED = -5,       /* ErrorKind::UnsupportedOperation (FULL-encoded doc blocks, payload / offset fields, k > RGPU_MAX_K ...) */
  RGPU_ERR_IO = -6,                /* ErrorKind::IOError */
  RGPU_ERR_RUNTIME = -7            /* ErrorKind::RuntimeError (HIP failure, no device, out of HBM) */
} rgpu_status;

typedef struct rgpu_ctx rgpu_ctx;
typedef struct rgpu_segment rgpu_segment;

/* Knobs (the reference has plain config structs only: SURVEY.md §5). Zero-initialise for defaults.
 * Work-partitioning knobs — blocks_per_item, and_blocks_per_item, or_window_docs, or_dense_clauses, or_wide_window_docs,
 * or_lazy_cells, and_bitmaps, profile_kernels: results never depend on them
 * (tests/test_gpu_parity.py::test_work_partitioning_knobs_do_not_change_answers; a conjunction clause answered through its
 * doc bitmap yields the same freq and the same f32 score as one walked through its blocks).
 * Knobs that select another ARITHMETIC and so may change scores (never doc-id sets beyond the documented tolerance, never
 * hit counts): or_wide (-1: f32 clause-order sums instead of order-free fixed-point sums for >= 10 SHOULD clauses),
 * req_opt_rule (-1: scores >= the reference's for MUST + SHOULD trees), raw_norms (1: no score tables; also routes
 * >= 10-clause disjunctions to the clause-order kernel, as or_wide = -1 does), or_bitmaps (which of the two fixed-point
 * kernels takes a 10..16-clause disjunction: both sum the same fixed-point steps, but a posting's score is rounded to a step
 * from a table entry in one and from a reciprocal in the other — totals may differ by a few steps of 2^-e, far inside 1e-5). */
typedef struct rgpu_config {
  int32_t abi_version;          /* must be RGPU_ABI_VERSION */
  int32_t blocks_per_item;      /* 128-posting blocks per wave work item in the TERM kernel (0 = auto: 8..2048 by the batch's blocks) */
  int32_t and_blocks_per_item;  /* lead blocks per wave work item of the AND kernel (0 = auto: 8..72, ~28 k items per launch) */
  int32_t profile_kernels;      /* 1 = bracket every kernel with HIP events from the start (see rgpu_set_profiling) */
  int32_t or_window_docs;       /* docs per wave window of the ordered OR kernel (0 = default 1024; 256..4096, rounded to 256) */
  int32_t or_dense_clauses;     /* OR: clauses decoded inside the window kernel instead of through a scored run
                                   (0 = default 4, -1 = none; at most 4) */
  int32_t raw_norms;            /* 1 keeps raw norm bytes in HBM even when <= 64 distinct values exist (disables the
                                   per-clause LDS score table and everything built on it; A/B testing) */
  int32_t or_wide;              /* disjunctions of >= 10 SHOULD clauses (where the reference itself sums in heap order) through the
                                   order-free workgroup-window kernel: 0 = yes (default), -1 = no (clause-order kernel for all) */
  int32_t or_wide_window_docs;  /* docs per workgroup window of that kernel (0 = default 16384; 4096..20480, rounded up to a multiple of 4096) */
  int32_t req_opt_rule;         /* MUST + SHOULD trees: 0 = the reference's ReqOptScorer, skipping rule included (default: exact
                                   scores, a sequential pass per query); -1 = always add the optional sums (faster, scores >= the
                                   reference's) */
  int32_t or_bitmaps;           /* doc bitmaps for dense terms + k_or_lazy for the >= 10-clause disjunctions that name one
                                   (kernels/search_or_lazy.hpp): 0 = terms holding >= 1 doc in 64 (default), n > 0 = >= 1 doc in n,
                                   -1 = off (k_or_wide walks every clause). A bitmap costs 3 max_doc / 8 + doc_freq bytes of HBM (+ max_doc / 2
                                   for a term that holds a doc in 128 or more: four bits per doc — absent / the freq) */
  int32_t or_lazy_cells;        /* accumulator cells (touched docs) per window of k_or_lazy (0 = default 512; 512..4096) */
  int32_t and_bitmaps;          /* conjunctions: a clause behind the lead whose term has a doc bitmap answers a candidate with one bit
                                   instead of a walk through its blocks. 0 = terms holding >= 1 doc in 256 get a bitmap (default;
                                   measured on the 3-term batch: 1 in 64 0.59 ms, 256 0.51, 1024 0.54, no bitmaps 1.55),
                                   n > 0 = >= 1 doc in n, -1 = off */
  int32_t bitmap_budget_mib;    /* HBM the doc bitmaps of ALL segments of this context may hold together, in MiB: 0 = an eighth of
                                   the device's memory (default), n > 0 = n MiB. Bitmaps are an accelerator, never a requirement: a
                                   term the budget has no room for (or whose allocation fails) is answered by walking its blocks,
                                   with the same results; rgpu_segment_footprint.doc_bitmap_refused counts such terms and
                                   rgpu_segment_release_prepared_terms gives the budget back */
  int32_t prepared_budget_mib;  /* HBM one segment's prepared-term store (block directory + aligned block store + posting-order norms:
                                   rgpu_segment_footprint) may hold when a batch arrives, in MiB; 0 = no ceiling (default: every
                                   term ever queried stays prepared for the life of the segment — 2.1..2.4 HBM bytes per .doc byte
                                   when the whole vocabulary has been touched). Over the ceiling the store is dropped as a whole
                                   between two batches and refilled by what the next batches name (the store is one dense region
                                   per preparing call: keeping the recently used terms means preparing them again, which is what
                                   their next use does anyway). The reference holds nothing per term (posting_reader.rs:460-500).
                                   Results never depend on it; doc bitmaps have their own budget (bitmap_budget_mib) and stay */
  int32_t or_deferred;          /* rgpu_search_batch_device with >= 10-clause disjunctions: 0 (default) = the call returns once the batch's
                                   hand-back flags have been looked at (a top-k below the fixed-point floor, a window that did not fit:
                                   those queries run again through another kernel) — one stream synchronisation per OR group; 1 = the
                                   call only enqueues, like TERM / AND batches: the flags are looked at by the NEXT call that needs the
                                   scratch slot, or by rgpu_synchronize. The caller must then call rgpu_synchronize(ctx) — not just
                                   synchronise its stream — before it reads such a batch's rows. The host's planning of batch i + 1
                                   then runs under the kernels of batch i. Results are the same either way */
  int32_t comm_force_gather;    /* communicators of ONE rank: 0 (default) = nothing to gather, the record is merged where the search
                                   left it; 1 = issue the in-place ncclAllGather all the same (and the cross-stream ordering around
                                   it) — the N > 1 data path end to end on a box with one GPU: tests and `bench.py --force-dist`
                                   (the environment variable RGPU_COMM_FORCE_GATHER=1 does the same). Results never depend on it */
} rgpu_config;

/* blocktree/mod.rs:33-59 BlockTermState, as filled by posting_reader.rs:264-306 lucene50_decode_term.
 * doc_freq == 0 means "term absent from this segment" (TermWeight::create_scorer -> None). */
typedef struct rgpu_term_state {
  int64_t doc_start_fp;      /* where this term's postings start in .doc */
  int64_t skip_offset;       /* skip data at doc_start_fp + skip_offset, -1 unless doc_freq > 128 */
  int64_t total_term_freq;   /* freq of the singleton posting when doc_freq == 1 */
  int32_t doc_freq;
  int32_t singleton_doc_id;  /* -1 unless doc_freq == 1 */
} rgpu_term_state;

/* One scored clause: TermQuery -> TermWeight (term_query.rs:58-95). `weight` and the 256-entry norm
 * cache are computed on the host exactly as bm25_similarity.rs:151-177 (idf in f64 -> f32, weight =
 * idf * boost) so the device never re-derives idf. `sim_table` is a handle from rgpu_sim_table_upload. */
typedef struct rgpu_query_term {
  rgpu_term_state state;
  float weight;
  int32_t sim_table;
} rgpu_query_term;

typedef enum rgpu_query_op {
  RGPU_OP_TERM = 0, /* TermQuery                      -> TermScorer (term_scorer.rs:43-67) */
  RGPU_OP_AND = 1,  /* BooleanQuery, all MUST         -> ConjunctionScorer (conjunction_scorer.rs:26-128) */
  RGPU_OP_OR = 2    /* BooleanQuery, all SHOULD         -> DisjunctionSumScorer (disjunction_scorer.rs:24-104) */
} rgpu_query_op;
/* rgpu_query.op for an OR query may carry BooleanQuery's min_should_match in its second byte:
 * RGPU_OP_OR | (msm << 8). 0 and 1 are the default (any clause matches); msm >= 2 collects only docs held by at least
 * that many SHOULD clauses (disjunction_scorer.rs:317-329) and, as in the reference, sums in clause order whatever
 * the clause count (SimpleQueue is forced, :41), so those scores are bit-exact. */
#define RGPU_OP_OR_MSM(msm) ((int32_t)RGPU_OP_OR | ((int32_t)(msm) << 8))
/* A TERM / AND query may carry optional SHOULD TermQuery clauses in its third byte: RGPU_OP_WITH_SHOULD(op, n). They are
 * stored between the MUST clauses and the MUST_NOT ones. BooleanWeight::create_scorer builds ReqOptScorer(must,
 * DisjunctionSumScorer(should)) for such a tree (query/boolean_query.rs:217-233, 253-262; scorer/req_opt_scorer.rs): the
 * SHOULD clauses never change which docs match, each one found on a matching doc adds its score (summed on their own in
 * clause order, then added to the MUST sum). ReqOptScorer::score carries state from doc to doc: once more than 100 docs
 * took the optional path, a doc whose MUST score is under half the running mean of those docs' MUST scores returns the
 * MUST score alone (:46-50). That rule is applied (default): the conjunction kernel leaves one record per lead posting and a
 * second kernel walks a query's matches in doc order (a sequential pass: a few ms per million matches) — scores equal the
 * reference's bit for bit. rgpu_config.req_opt_rule = -1 always adds the optional sums instead (one pass; doc ids and hit
 * counts equal the reference's, scores >= its and equal wherever it did not skip). */
#define RGPU_OP_WITH_SHOULD(op, n_should) ((int32_t)(op) | ((int32_t)(n_should) << 16))
/* RGPU_OP_SHOULD_REQUIRED on top of RGPU_OP_WITH_SHOULD: the n SHOULD clauses are a MUST clause of their own — the tree
 * "+a +(b c)", a should-only BooleanQuery (min_should_match <= 1) nested under MUST, which BooleanWeight::create_scorer turns into
 * ConjunctionScorer([TermScorer(a) ..., DisjunctionSumScorer(b, c)]) (query/boolean_query.rs:200-215, 217-233;
 * scorer/conjunction_scorer.rs:27-43). A doc matches when every MUST clause AND at least one of the n SHOULD clauses hold it; its
 * score is ConjunctionScorer::score (:87-95) over the children in their stable cost order, the disjunction contributing its own
 * sum (clause order) at its place — see RGPU_OP_NESTED_AT: bit-equal to the reference. ReqOptScorer's doc-to-doc rule does not
 * apply (there is no ReqOptScorer in this tree), rgpu_config.req_opt_rule is not consulted. 1 <= n <= 9 (ten or more children
 * sum in heap order: disjunction_scorer.rs:41-45); every SHOULD clause absent from the leaf = the nested weight has no scorer =
 * the query matches nothing there (boolean_query.rs:203-207). */
#define RGPU_OP_SHOULD_REQUIRED ((int32_t)1 << 24)
/* RGPU_OP_NESTED_MUST on top of RGPU_OP_WITH_SHOULD: the n clauses behind the MUST clauses are a must-only BooleanQuery nested
 * under MUST — "+a +(+b +c)", ConjunctionScorer([TermScorer(a) ..., ConjunctionScorer(b, c)]). A doc matches when every MUST clause
 * and every one of the n nested clauses hold it (the same docs and hit count as the flat conjunction); its score is the MUST sum
 * (cost order) plus the nested conjunction's own sum — lead1 + lead2 + others over the n clauses in THEIR cost order, formed first
 * (conjunction_scorer.rs:27-43, 87-95; the library sorts them by doc_freq per leaf, stable), added at the nested child's place
 * in the outer cost order — see RGPU_OP_NESTED_AT: bit-equal to the reference. n >= 2 (a nested
 * query of one clause is that clause: boolean_query.rs:56-68); a nested clause absent from the leaf = no scorer = nothing matches.
 * Not combined with RGPU_OP_SHOULD_REQUIRED. */
#define RGPU_OP_NESTED_MUST ((int32_t)1 << 25)
/* With either flag, where the nested clause stands among the MUST clauses: RGPU_OP_NESTED_AT(i) = it is the caller's (i + 1)-th
 * MUST child, i of the n_terms MUST term clauses precede it (0 .. n_terms; 0 when omitted). ConjunctionScorer::new sorts ALL its
 * children by cost(), stable — a term's doc_freq in the leaf, a nested disjunction's the sum of its clauses' doc freqs, a nested
 * conjunction's its cheapest clause's (conjunction_scorer.rs:30, 111-113) — and score() adds them in that order (:87-95). The
 * library does that sort per leaf: the MUST terms that sort before the nested child are summed first, the nested sum is added to
 * that, the MUST terms behind it are added one by one — the reference's f32 sum for EVERY such tree, bit for bit, whatever the
 * costs (the index only breaks ties, as the stable sort does). */
#define RGPU_OP_NESTED_AT(i) ((int32_t)((uint32_t)(i) << 26))

typedef struct rgpu_query {
  int32_t op;          /* rgpu_query_op (OR: optionally RGPU_OP_OR_MSM(msm); TERM / AND: optionally RGPU_OP_WITH_SHOULD(op, n) —
                          a min_should_match in the second byte is accepted there too and, as in the reference, has no effect:
                          ReqOptScorer only advance()s the optional scorer) */
  int32_t n_terms;     /* required / scored clauses: 1 for TERM, 1..RGPU_MAX_QUERY_TERMS MUST (AND) / SHOULD (OR) clauses; 0 for an OR
                          query of MUST_NOT clauses only, which matches nothing (BooleanWeight::create_scorer -> None);
                          the n optional SHOULD clauses of RGPU_OP_WITH_SHOULD follow them and are not counted here */
  int32_t first_term;  /* index of this query's first clause in the `terms` array */
  int32_t n_must_not;  /* MUST_NOT TermQuery clauses, stored right after the positive ones (weight / sim_table unused):
                          BooleanWeight::create_scorer wraps the positive scorer in a ReqNotScorer over their union
                          (query/boolean_query.rs:235-273, scorer/req_not_scorer.rs:20-120). 0 = none;
                          n_terms + n_should + n_must_not <= RGPU_MAX_QUERY_TERMS */
} rgpu_query;

/* sort_field/collapse_top_docs.rs:22-36 ScoreDoc */
typedef struct rgpu_hit {
  int32_t doc;  /* doc + doc_base (collector/top_docs.rs:89); -1 pads unused slots */
  float score;
} rgpu_hit;

/* ---- context ---------------------------------------------------------------------------------------------- */
/* No reference counterpart (Rucene is single-process CPU); created once by the shim. */
int32_t rgpu_init(int32_t device_ordinal, const rgpu_config* cfg_or_null, rgpu_ctx** out_ctx);
void rgpu_shutdown(rgpu_ctx* ctx);
const char* rgpu_last_error(rgpu_ctx* ctx_or_null); /* message of the last failing call on this thread */
int32_t rgpu_abi_version(void);
int32_t rgpu_device_name(rgpu_ctx* ctx, char* buf, size_t buf_len);

/* ---- segment (per-leaf, immutable) ------------------------------------------------------------------------ */
/* Stands in for Lucene50PostingsReader::open (posting_reader.rs:85-158): validates the IndexHeader
 * ("Lucene50PostingsWriterDoc", version 0..1), parses the ForUtil table (for_util.rs:120-148), checks the
 * footer magic, then copies the raw .doc bytes, the 1-byte-per-doc norms (norms_producer.rs:146-154) and
 * the optional live-docs bitset (util/bit_set.rs:453-460: bit doc&63 of word doc>>6; NULL = MatchAllBits)
 * into HBM. Version 1 selects the SIMD-BP128 block layout, version 0 the legacy PackedInts layout. */
int32_t rgpu_segment_upload(rgpu_ctx* ctx, const uint8_t* doc_file, size_t doc_len, const uint8_t* norms_or_null,
                            int32_t max_doc, int32_t doc_base, const uint64_t* live_docs_or_null,
                            rgpu_segment** out_seg);
/* The same for a field of the given doc::IndexOptions ordinal: 1 = Docs (no freq block follows a doc block, the VInt
 * tail holds plain deltas, every freq reads as 1 and a FREQS-less iterator's skip_block has nothing to skip:
 * posting_reader.rs:532-557, for_util.rs:263-272), 2 = DocsAndFreqs (what rgpu_segment_upload assumes).
 * 3 = DocsAndFreqsAndPositions (skip entries carry position pointers; see rgpu_segment_attach_positions /
 * rgpu_search_phrase_batch), 4 = DocsAndFreqsAndPositionsAndOffsets; a field whose FieldInfo::has_store_payloads is set
 * passes index_options | RGPU_FIELD_STORES_PAYLOADS (3 or 4). Such fields keep payloads and offsets in a third file
 * (".pay", rgpu_segment_attach_payloads), their skip entries carry one or two more words (skip_writer.rs:276-286) and the
 * trailing VInt block of a term's positions carries payload bytes and offset words between the position deltas
 * (posting_writer.rs:505-560): every search entry point — TERM / AND / OR, exact and sloppy phrases — serves them; what the
 * reference's scorers never ask for (EverythingIterator's payload() / start_offset() / end_offset(), posting_reader.rs:
 * 1595-2337) is not decoded on the GPU. For a Docs field rgpu_term_state.total_term_freq is ignored. */
#define RGPU_FIELD_STORES_PAYLOADS 0x100
int32_t rgpu_segment_upload_field(rgpu_ctx* ctx, const uint8_t* doc_file, size_t doc_len, const uint8_t* norms_or_null,
                                  int32_t max_doc, int32_t doc_base, const uint64_t* live_docs_or_null, int32_t index_options,
                                  rgpu_segment** out_seg);
void rgpu_segment_free(rgpu_segment* seg);
int32_t rgpu_segment_version(const rgpu_segment* seg); /* .doc format version (0 legacy, 1 BP128) */

/* Skip-list decode (Lucene50SkipReader::init/load_skip_levels/read_skip_data, skip_reader.rs:315-511):
 * decodes each term's level-0 skip entries on the GPU into a flat per-term block directory
 * {last doc id, file offset, header bytes} cached in HBM for the life of the segment — the GPU analogue
 * of opening a term's skipper — and re-lays the term's blocks as 16-byte aligned rows (stage A: all a decode needs);
 * then, for scoring, gathers the norm byte of every posting in posting order and bounds every block's best score
 * (stage B). Called implicitly by every entry point below for terms it has not seen (rgpu_decode_terms* and
 * rgpu_advance_batch: stage A only); exposed so a caller can pay both stages at segment-open time. */
int32_t rgpu_segment_prepare_terms(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms);
/* The per-term structures (directory, 16-byte aligned block copies, posting-order norms: about the term's share of the
 * .doc file again plus 1 byte per posting) are kept for every distinct term ever queried, for the life of the segment;
 * HBM use therefore grows towards ~2x the .doc size as the query vocabulary widens, and running out surfaces as
 * RGPU_ERR_RUNTIME from the call that needed the space. This drops them all (after waiting for work in flight); terms
 * are prepared again the next time they are used. */
int32_t rgpu_segment_release_prepared_terms(rgpu_segment* seg);
/* What a segment holds in HBM right now, by part (bytes in use). */
typedef struct rgpu_segment_footprint {
  int64_t doc_file_bytes;        /* the .doc bytes, verbatim */
  int64_t norms_bytes;           /* 1 byte per doc */
  int64_t live_docs_bytes;
  int64_t positions_file_bytes;  /* the .pos bytes (rgpu_segment_attach_positions) */
  int64_t directory_bytes;       /* prepared terms: 22 bytes per block (+ 8 for a positions field), + 8 per 64 blocks (chunk frontiers) */
  int64_t block_store_bytes;     /* prepared terms: 16-byte aligned payload rows + decoded tails */
  int64_t posting_norms_bytes;   /* prepared terms: 1 byte per posting */
  int64_t prepared_terms;        /* how many distinct terms are prepared */
  int64_t doc_bitmap_bytes;      /* doc bitmaps of the dense terms (rgpu_config.or_bitmaps): words + ranks + freq bytes */
  int64_t doc_bitmap_terms;      /* terms that hold one */
  int64_t doc_bitmap_refused;    /* dense terms that got none (rgpu_config.bitmap_budget_mib spent, allocation failed, or a list a
                                    bitmap cannot express): their clauses are walked */
} rgpu_segment_footprint;
int32_t rgpu_segment_get_footprint(rgpu_segment* seg, rgpu_segment_footprint* out);

/* BlockDocIterator over whole terms (posting_reader.rs:501-647: refill_docs + next, i.e. ForUtil
 * read_block for docs and freqs, VInt tail, singleton) — decodes every posting of every given term into
 * caller-owned HOST buffers, terms concatenated in order (sum of doc_freq entries each). The parity and
 * micro-benchmark surface for block decode. */
int32_t rgpu_decode_terms(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, int32_t* docs_out,
                          int32_t* freqs_out);
/* Same, results left in device memory (docs_dev / freqs_dev are hipMalloc'd by the caller or a torch
 * tensor's data_ptr); asynchronous on `hip_stream` (a hipStream_t, NULL = the ctx stream). */
int32_t rgpu_decode_terms_device(rgpu_segment* seg, const rgpu_term_state* terms, int64_t n_terms, void* docs_dev,
                                 void* freqs_dev, void* hip_stream);

/* BlockDocIterator::advance (posting_reader.rs:649-789) for a batch of independent probes on one term:
 * out_docs[i] = first doc >= targets[i] (RGPU_NO_MORE_DOCS if none), out_freqs[i] its freq. */
int32_t rgpu_advance_batch(rgpu_segment* seg, const rgpu_term_state* term, const int32_t* targets, int64_t n_targets,
                           int32_t* out_docs, int32_t* out_freqs);

/* ---- similarity tables ------------------------------------------------------------------------------------ */
/* BM25SimWeight::cache (bm25_similarity.rs:158-165): 256 f32 = k1*((1-b) + b*NORM_TABLE[i]/avgdl),
 * plus k1 for the final formula and the no-norms case. Returns a handle >= 0 or a negative status. */
int32_t rgpu_sim_table_upload(rgpu_ctx* ctx, const float cache[256], float k1);

/* ---- search ----------------------------------------------------------------------------------------------- */
/* One leaf of IndexSearcher::search (searcher.rs:487-525): for each query, create_scorer + BulkScorer::score
 * (bulk_scorer.rs:57-154) into a TopDocsCollector(k) (collector/top_docs.rs:28-95) — all on the GPU.
 *   hits_out        n_queries x k, best first; order = score desc, then doc asc (the canonical tie rule of
 *                   SURVEY.md §8(c)); unused slots {-1, 0}
 *   total_hits_out  n_queries, TopDocs::total_hits (every collected live doc)
 * Scores are f32 computed in the reference's operation order: TERM, AND and OR with < 10 clauses bit-exact with the
 * CPU scorers. OR with >= 10 clauses: within 1e-5 relative — the reference sums those in heap order
 * (disjunction_scorer.rs:41-45), so it pins a score no tighter itself; here such a doc's score is the exact sum of its
 * clause scores in fixed point (2^-e steps, e per query), rounded to f32 once: deterministic, independent of any
 * order, hit counts exact (kernels/search_or_wide.hpp; rgpu_config.or_wide = -1 sums in f32 in clause order instead). MUST + SHOULD
 * trees follow the reference's ReqOptScorer including its sequential skipping rule (see RGPU_OP_WITH_SHOULD). */
int32_t rgpu_search_batch(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                          int32_t n_terms_total, int32_t k, rgpu_hit* hits_out, int64_t* total_hits_out);
/* Same with device-resident outputs (for the RCCL all-gather of per-shard top-k). Enqueue-only for TERM / AND
 * batches: the call stages its inputs, launches on hip_stream (NULL = the context's own stream) and returns; the
 * outputs are complete when the stream reaches that point (hipStreamSynchronize, or rgpu_synchronize for the
 * context's stream). Back-to-back calls therefore overlap the host-side planning of batch i+1 with the kernels of
 * batch i. `queries` / `terms` are copied before the call returns.
 * What blocks: an OR group waits for its kernels once (the fixed-point kernels' floor / hand-back flags come back to the
 * host), and so does a MUST + SHOULD group under the exact ReqOptScorer rule. k > 128 runs ceil(k / 128) passes of the
 * whole search, each enqueue-only under the same rules (validation and grouping are repeated per pass: host time grows
 * with the pass count). Arithmetic of deep pages: the fixed-point kernels of the >= 10-clause disjunctions rank by exact
 * totals and round once, which has no ceiling key across passes — for k > 128 such a disjunction is summed in f32 in
 * clause order instead (still inside the reference's 1e-5; a score may differ in its last bits between k = 128 and k = 129). */
int32_t rgpu_search_batch_device(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries,
                                 const rgpu_query_term* terms, int32_t n_terms_total, int32_t k, void* hits_dev,
                                 void* total_hits_dev, void* hip_stream);

/* TopDocsCollector::finish_parallel (collector/top_docs.rs:157-172): merge n_lists per-leaf / per-shard
 * top-k lists (layout [list][query][k], device memory) into [query][k] under the canonical order and sum
 * the hit counts ([list][query] -> [query]). Enqueue-only on hip_stream, like rgpu_search_batch_device. */
int32_t rgpu_merge_topk_device(rgpu_ctx* ctx, const void* hits_dev, const void* totals_dev, int32_t n_lists,
                               int32_t n_queries, int32_t k, void* hits_out_dev, void* totals_out_dev, void* hip_stream);

/* ---- segment-sharded search across GPUs (one process per GPU, RCCL over xGMI) ------------------------------------ */
/* Rucene's search_parallel sends every leaf's heap over a channel and merges them in finish_parallel
 * (search/searcher.rs:527-630, collector/top_docs.rs:157-172, 201-214). Here the leaves of an index live on different
 * GPUs (segment s -> rank s, doc_base = cumulative max_doc), the query batch is replicated, and the channel is ONE
 * ncclAllGather per batch of each rank's {k hits per query, hit count per query} record (n_queries * (k + 1) * 8
 * bytes), followed on every rank by the canonical k-way merge (k_merge_lists) and the sum of the counts. BM25
 * statistics must already be those of the whole index (the largest leaf's, searcher.rs:311-351): they travel inside
 * the query terms' weights, so no rank re-derives idf. */
typedef struct rgpu_comm rgpu_comm;
#define RGPU_COMM_ID_BYTES 128  /* = NCCL_UNIQUE_ID_BYTES */
/* ncclGetUniqueId: called on one rank; the caller hands the bytes to the other ranks over whatever channel it has. */
int32_t rgpu_comm_unique_id(uint8_t id_out[RGPU_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device; collective: every rank of the job calls it with the same id. */
int32_t rgpu_comm_init(rgpu_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[RGPU_COMM_ID_BYTES], rgpu_comm** out_comm);
void rgpu_comm_destroy(rgpu_comm* comm);
/* rgpu_search_batch_device on this rank's segment -> all-gather -> merge, all enqueued in that order on hip_stream
 * (NULL = the context's stream); enqueue-only, collective (every rank calls it with the same batch shape).
 * hits_dev / total_hits_dev (device memory, n_queries x k and n_queries) receive the merged result on every rank.
 * A rank whose LOCAL search fails (corrupt segment, out of HBM while preparing terms, ...) still takes part in the
 * collective — with empty rows and its status in the record (see rgpu_record_bytes) — and only then returns its error, so
 * its peers neither hang in the all-gather nor keep a stale record: they get the merge of the shards that answered and can
 * ask rgpu_comm_status which did not. Nothing between the buffer reservation and the collective returns early: a failed
 * wait or status-word memset is reported after the all-gather has been joined. The one exception is the allocation of the
 * gather buffer itself (nowhere to receive into — the communicator is then out of step: destroy it); rgpu_comm_reserve
 * moves that allocation to start-up. */
int32_t rgpu_search_batch_sharded(rgpu_comm* comm, rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries,
                                  const rgpu_query_term* terms, int32_t n_terms_total, int32_t k, void* hits_dev,
                                  void* total_hits_dev, void* hip_stream);
/* Start-up sizing (not a collective): allocates every slot's gather buffer for batches of up to n_queries x k now, so that
 * the data path never allocates. */
int32_t rgpu_comm_reserve(rgpu_comm* comm, int32_t n_queries, int32_t k);
/* ncclAllGather calls enqueued on this communicator so far (a communicator of one rank issues none unless
 * rgpu_config.comm_force_gather = 1); -1 for a null communicator. */
int64_t rgpu_comm_gathers_issued(rgpu_comm* comm);
/* status_out[r], r < n_ranks: rank r's rgpu_status for the most recent sharded batch on this communicator (waits for it). */
int32_t rgpu_comm_status(rgpu_comm* comm, int32_t* status_out);
/* One process that owns several GPUs (Rucene itself is one process whose search_parallel hands leaves to threads,
 * search/searcher.rs:527-630): ctxs[r] = rank r's context, each on its own device. The n ncclCommInitRank calls are issued
 * inside one ncclGroupStart / ncclGroupEnd (one after the other from one thread they would wait for each other forever). */
int32_t rgpu_comm_init_all(rgpu_ctx* const* ctxs, int32_t n, rgpu_comm** out_comms);
/* The one-process form of rgpu_search_batch_sharded: rank r searches segs[r] with terms_per_rank[r] (the same queries; term
 * states differ per shard, weights do not) — every shard's search is enqueued first, then the n all-gathers as ONE NCCL
 * group, then the n merges. hip_streams may be NULL (each context's own stream). */
int32_t rgpu_search_batch_sharded_all(rgpu_comm* const* comms, rgpu_segment* const* segs, int32_t n, const rgpu_query* queries,
                                      int32_t n_queries, const rgpu_query_term* const* terms_per_rank, int32_t n_terms_total, int32_t k,
                                      void* const* hits_dev, void* const* total_hits_dev, void* const* hip_streams);
/* The two halves of a sharded batch without the collective between them — what a host with its own transport (or a test
 * with one GPU) composes: a shard's record = [n_queries x k rgpu_hit][n_queries x int64 hit count][int64 rgpu_status],
 * rgpu_record_bytes(n_queries, k) bytes; rgpu_search_batch_record_device fills one (enqueue-only; the status word is the
 * call's own return value, rows are empty when it failed); rgpu_merge_records_device runs finish_parallel
 * (collector/top_docs.rs:157-172) over n_ranks records laid out back to back — the all-gather's receive buffer. */
int64_t rgpu_record_bytes(int32_t n_queries, int32_t k);
int32_t rgpu_search_batch_record_device(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                                        int32_t n_terms_total, int32_t k, void* record_dev, void* hip_stream);
int32_t rgpu_merge_records_device(rgpu_ctx* ctx, const void* records_dev, int32_t n_ranks, int32_t n_queries, int32_t k,
                                  void* hits_out_dev, void* totals_out_dev, void* hip_stream);

/* ---- host helpers (no GPU work) ----------------------------------------------------------------------------- */
/* BM25Similarity::compute_weight (bm25_similarity.rs:151-177) for one TermQuery (n_terms = 1) or a multi-term
 * weight: idf summed over `doc_freqs` (idf() :99-114, f64 log -> f32), avgdl = sumTTF / docCount
 * (avg_field_length :72-83; doc_count == -1 -> max_doc; sumTTF <= 0 -> 1), cache[i] = k1*((1-b) + b*NORM_TABLE[i]/avgdl),
 * weight = idf * boost. Outputs may be NULL. */
int32_t rgpu_bm25_compute_weight(float k1, float b, int64_t max_doc, int64_t doc_count, int64_t sum_total_term_freq,
                                 const int64_t* doc_freqs, int32_t n_terms, float boost, float* weight_out, float* idf_out,
                                 float* cache_out /* 256 floats */);
/* The weights of `n` single-term queries at once: weights_out[i] = idf(doc_freqs[i]) * boost, the arithmetic of
 * rgpu_bm25_compute_weight with n_terms = 1 (TermWeight::new -> BM25Similarity::compute_weight, term_query.rs:69-95),
 * for a planner that resolves a whole batch of terms before it packs rgpu_query_term[]. The norm cache of such weights
 * does not depend on the term: take it once from rgpu_bm25_compute_weight. */
int32_t rgpu_bm25_term_weights(int64_t max_doc, int64_t doc_count, const int64_t* doc_freqs, int64_t n, float boost,
                               float* weights_out);
/* BM25Similarity::encode_norm_value (bm25_similarity.rs:90-92): float_to_byte315(boost / sqrt(field_length)). */
uint8_t rgpu_bm25_encode_norm(float boost, int32_t field_length);
/* Lucene53NormsProducer (codec/norms/norms_producer.rs:40-189; format constants codec/norms/norms.rs:23-28): the
 * 1-byte-per-doc norms array rgpu_segment_upload takes, read from a segment's ".nvm" (metadata) and ".nvd" (data)
 * files: norms_out[doc] = norms(field_number).get(doc) & 0xFF (what BM25 reads, bm25_similarity.rs:205) for
 * doc in [0, max_doc). Validates both index headers (same segment id / suffix / version), the metadata checksum
 * (check_footer) and the data footer (retrieve_checksum). */
int32_t rgpu_norms_from_lucene53(const uint8_t* nvm, size_t nvm_len, const uint8_t* nvd, size_t nvd_len, int32_t field_number,
                                 int32_t max_doc, uint8_t* norms_out);
/* Lucene50LiveDocsFormat::read_live_docs (codec/live_docs.rs:81-121): a segment's ".liv" file -> the FixedBitSet words
 * rgpu_segment_upload takes (ceil(max_doc / 64) words; bit doc&63 of word doc>>6 set = live). Validates the index
 * header, the checksum, clear ghost bits and, when del_count >= 0, max_doc - cardinality == del_count. */
int32_t rgpu_live_docs_from_lucene50(const uint8_t* liv, size_t liv_len, int32_t max_doc, int32_t del_count, uint64_t* words_out);

/* ---- term dictionary (host; no GPU work) ---------------------------------------------------------------------- */
/* A segment's block-tree term dictionary: ".tim" (BlockTreeTermsDict, term blocks) + ".tip" (BlockTreeTermsIndex, one
 * FST per field). Replaces BlockTreeTermsReader::new (codec/postings/blocktree/blocktree_reader.rs:132-304),
 * FieldReader's Terms statistics (:390-548) and, for exact lookups, SegmentTermIterator::seek_exact + term_state
 * (:1364-1550, :1805; blocktree/term_iter_frame.rs:176-402; posting_reader.rs:264-306 lucene50_decode_term).
 * rgpu_terms_open enumerates every term block once and hashes term bytes -> rgpu_term_state, so a lookup costs one
 * hash probe instead of an FST walk + block scan. Format versions 0 and 3 (what Rucene writes);
 * auto-prefix versions 1-2 -> RGPU_ERR_UNSUPPORTED. The handle is immutable: lookups may run from any thread. */
typedef struct rgpu_terms rgpu_terms;
typedef struct rgpu_field_info {   /* the slice of FieldInfo (codec/field_infos/mod.rs) the dictionary needs */
  int32_t number;                  /* FieldInfo::number */
  int32_t index_options;           /* doc::IndexOptions ordinal: 1 Docs, 2 DocsAndFreqs, 3 +Positions, 4 +Offsets */
  int32_t has_payloads;            /* FieldInfo::has_store_payloads */
  int32_t flags;                   /* filled by rgpu_field_infos_from_lucene60, ignored by rgpu_terms_open:
                                      bit 0 omit_norms, bit 1 has_store_term_vector, bits 8..15 DocValuesType ordinal */
} rgpu_field_info;
typedef struct rgpu_field_stats {  /* Terms::{size, sum_total_term_freq, sum_doc_freq, doc_count} (blocktree_reader.rs:502-516) */
  int64_t num_terms;
  int64_t sum_total_term_freq;     /* -1 for IndexOptions::Docs */
  int64_t sum_doc_freq;
  int32_t doc_count;
  int32_t longs_size;
} rgpu_field_stats;
/* Lucene60FieldInfosFormat::read (codec/field_infos/field_infos_format.rs:55-128, 186-212): a segment's ".fnm" file ->
 * one rgpu_field_info per field (ascending as stored) and the field names as consecutive NUL-terminated UTF-8 strings
 * in names_out. Returns the number of fields in the file (>= 0) or a negative status; at most `cap` infos and
 * `names_cap` name bytes are written (call with cap = 0 to size). Verifies the index header, FieldInfo consistency,
 * duplicate numbers / names and the checksum. */
int32_t rgpu_field_infos_from_lucene60(const uint8_t* fnm, size_t fnm_len, rgpu_field_info* infos_out, int32_t cap, char* names_out,
                                       size_t names_cap, size_t* names_len_out);
/* Lucene62SegmentInfoFormat::read (codec/segment_infos/segment_infos_format.rs:43-247): a segment's ".si" file.
 * expected_id16: the id the commit point holds for the segment (check_index_header_id), or NULL. */
typedef struct rgpu_segment_info {
  int32_t max_doc;
  int32_t is_compound_file;  /* 1: the segment's files live inside .cfs / .cfe (rgpu_compound_entries_from_lucene50) */
  int32_t version[3];        /* major, minor, bugfix of the writer */
  int32_t n_files;
  int32_t n_sort_fields;     /* > 0: the segment is index-sorted */
  int32_t reserved;
  uint8_t id[16];
} rgpu_segment_info;
int32_t rgpu_segment_info_from_lucene62(const uint8_t* si, size_t si_len, const uint8_t* expected_id16_or_null, rgpu_segment_info* out);
/* SegmentInfos::read_commit (codec/segment_infos/segment_infos.rs:443-569): the commit point "segments_N". generation = N
 * (the header suffix must be its base-36 form; < 0 skips that check). Returns the number of segments in the commit
 * (>= 0) or a negative status and fills at most `cap` records. The reference also bounds del_count by the segment's
 * max_doc while it reads; here that is the caller's job once it has the ".si" (rgpu_live_docs_from_lucene50 re-checks). */
typedef struct rgpu_commit_segment {
  char name[48];             /* "_0", "_1", ...: files are <name>.si, <name>.fnm, <name>_Lucene50_0.doc ..., <name>_<delgen36>.liv */
  char codec[16];            /* "Lucene62" */
  uint8_t id[16];
  int64_t del_gen;           /* -1: no deletions file */
  int64_t field_infos_gen;
  int64_t dv_gen;
  int32_t del_count;
  int32_t reserved;
} rgpu_commit_segment;
int32_t rgpu_commit_from_segments_file(const uint8_t* data, size_t len, int64_t generation, rgpu_commit_segment* out, int32_t cap);
/* Lucene50CompoundReader (codec/compound.rs:116-195): a compound segment's ".cfe" entry table -> where each of the segment's
 * files lies inside ".cfs" (each is copied there whole: header, body, footer). Entry ids are file names without the segment
 * name (".fnm", "_Lucene50_0.doc", ...: strip_segment_name, codec/segment_infos/mod.rs:64-79). cfs may be NULL; when given its
 * header, footer and total length are checked as the reference does. Returns the number of entries (>= 0) or a negative
 * status; fills at most `cap` records. */
typedef struct rgpu_compound_entry {
  char id[112];
  int64_t offset;   /* from the start of the .cfs file */
  int64_t length;
} rgpu_compound_entry;
int32_t rgpu_compound_entries_from_lucene50(const uint8_t* cfe, size_t cfe_len, const uint8_t* cfs_or_null, size_t cfs_len,
                                            const uint8_t* expected_id16_or_null, rgpu_compound_entry* out, int32_t cap);
int32_t rgpu_terms_open(const uint8_t* tim, size_t tim_len, const uint8_t* tip, size_t tip_len, const rgpu_field_info* infos,
                        int32_t n_infos, int32_t max_doc, rgpu_terms** out_terms);
void rgpu_terms_close(rgpu_terms* terms);
/* RGPU_ERR_ILLEGAL_ARGUMENT when the segment has no such indexed field (Fields::terms -> None). */
int32_t rgpu_terms_field_stats(const rgpu_terms* terms, int32_t field_number, rgpu_field_stats* out);
/* seek_exact + term_state for n_terms terms of one field; term i = term_bytes[term_offsets[i] .. term_offsets[i+1]).
 * found_out[i] = 1/0; an absent term yields the "absent" state (doc_freq 0, skip_offset -1, singleton_doc_id -1),
 * which rgpu_search_batch treats as TermWeight::create_scorer -> None. found_out may be NULL. */
int32_t rgpu_terms_lookup(const rgpu_terms* terms, int32_t field_number, const uint8_t* term_bytes, const int64_t* term_offsets,
                          int32_t n_terms, rgpu_term_state* states_out, uint8_t* found_out);
/* The rest of BlockTermState for a field indexed with positions (blocktree/mod.rs:33-59; lucene50_decode_term): where the
 * term's positions start in ".pos", its payloads / offsets in ".pay", and the offset of its last (vint) position block.
 * Same lookup as rgpu_terms_lookup plus positions_out[i] (zeros / -1 for an absent term or a field without positions).
 * rgpu_search_phrase_batch takes them beside each term's state. */
typedef struct rgpu_term_positions {
  int64_t pos_start_fp;
  int64_t pay_start_fp;           /* 0 unless the field stores payloads or offsets */
  int64_t last_pos_block_offset;  /* -1 unless total_term_freq > 128 */
} rgpu_term_positions;
int32_t rgpu_terms_lookup_positions(const rgpu_terms* terms, int32_t field_number, const uint8_t* term_bytes, const int64_t* term_offsets,
                                    int32_t n_terms, rgpu_term_state* states_out, rgpu_term_positions* positions_out, uint8_t* found_out);

/* ---- batch planner (host; no GPU work besides one sim-table upload) ------------------------------------------------------------ */
/* A whole batch of term / boolean queries, handed over as ARRAYS, -> the rgpu_query[] / rgpu_query_term[] of rgpu_search_batch*.
 * Replaces, per query per leaf, TermQuery::create_weight + IndexSearcher::term_statistics + BM25Similarity::compute_weight
 * (search/query/term_query.rs:58-95, search/searcher.rs:732-767, search/similarity/bm25_similarity.rs:99-177) and
 * TermWeight::create_scorer's seek_exact (term_query.rs:145-163): resolve every clause's term in the searched leaf (state)
 * and in the STATISTICS leaf (doc_freq: the first leaf with the largest max_doc, searcher.rs:306-363), weight = idf * boost
 * (f64 log -> f32, memoised by doc_freq), one sim table per planner. One planner per (searcher, leaf); thread-safe. */
typedef struct rgpu_planner rgpu_planner;
typedef struct rgpu_plan_stats {   /* CollectionStatistics the weights are computed against + the BM25 parameters */
  int64_t max_doc;                 /* of the whole reader */
  int64_t doc_count;               /* Terms::doc_count of the statistics leaf (-1: max_doc) */
  int64_t sum_total_term_freq;     /* of the statistics leaf */
  float k1, b;
} rgpu_plan_stats;
/* terms named by bytes: leaf_terms = the searched leaf's dictionary, stats_terms = the statistics leaf's (NULL: the same) */
int32_t rgpu_planner_create(rgpu_ctx* ctx, const rgpu_plan_stats* stats, const rgpu_terms* leaf_terms, const rgpu_terms* stats_terms_or_null,
                            int32_t field_number, rgpu_planner** out);
/* terms named by ids into flat tables of term states (synthetic indexes, or a host that keeps its own dictionary);
 * doc_freq 0 = absent. The tables are copied. */
int32_t rgpu_planner_create_flat(rgpu_ctx* ctx, const rgpu_plan_stats* stats, const rgpu_term_state* leaf_states, int64_t n_leaf,
                                 const rgpu_term_state* stats_states_or_null, int64_t n_stats, rgpu_planner** out);
void rgpu_planner_destroy(rgpu_planner* planner);
int32_t rgpu_planner_sim_table(const rgpu_planner* planner);   /* the handle its clauses carry */
/* ctx may be NULL in the two create calls: nothing is uploaded and clauses carry sim_table -1 until the host names the table
 * it uploaded itself (rgpu_sim_table_upload of BM25SimWeight::cache for this field's statistics). */
int32_t rgpu_planner_set_sim_table(rgpu_planner* planner, int32_t sim_table);
/* Every query the same shape: `op` = RGPU_OP_TERM (n_clauses 1), RGPU_OP_AND (all MUST) or RGPU_OP_OR / RGPU_OP_OR_MSM(m)
 * (all SHOULD), boost 1; term q * n_clauses + c = clause c of query q. A one-clause AND / OR is rewritten to that clause
 * (BooleanQuery::build, query/boolean_query.rs:66-75). Outputs: n_queries queries, n_queries * n_clauses terms. */
int32_t rgpu_plan_uniform_ids(rgpu_planner* planner, int32_t op, int32_t n_queries, int32_t n_clauses, const int64_t* term_ids,
                              rgpu_query* queries_out, rgpu_query_term* terms_out);
int32_t rgpu_plan_uniform_bytes(rgpu_planner* planner, int32_t op, int32_t n_queries, int32_t n_clauses, const uint8_t* term_bytes,
                                const int64_t* term_offsets, rgpu_query* queries_out, rgpu_query_term* terms_out);
/* Plan + search in ONE call: rgpu_plan_uniform_ids followed by rgpu_search_batch_device on `seg` (the leaf the planner's
 * flat table belongs to), without the rgpu_query[] / rgpu_query_term[] arrays in between — per query what IndexSearcher::search
 * does from the top (search/searcher.rs:487-525: create_normalized_weight -> TermQuery::create_weight, term_query.rs:58-95;
 * then per leaf TermWeight::create_scorer, :145-163, and the collector loop). Same rows as the two calls, bit for bit; enqueue-
 * only like rgpu_search_batch_device (rgpu_config.or_deferred applies). Single-term batches whose terms are all prepared take
 * a one-pass path (planner entry -> device descriptor, two enqueues: the host's share of a 1024-query batch drops from ~63
 * to ~13 us); every other batch is planned and searched as the two calls would. Flat-table planners only. The planner keeps a
 * memo of finished descriptors for that path (65536 records, 5 MB of host memory, allocated by the first such call; emptied
 * whenever the segment's prepared terms change). */
int32_t rgpu_planner_search_uniform_ids_device(rgpu_planner* planner, rgpu_segment* seg, int32_t op, int32_t n_queries, int32_t n_clauses,
                                               const int64_t* term_ids, int32_t k, void* hits_dev, void* total_hits_dev, void* hip_stream);
/* ... and as rgpu_search_batch_sharded: this rank's plan + search -> the all-gather of the shards' records -> the merge. */
int32_t rgpu_planner_search_uniform_ids_sharded(rgpu_comm* comm, rgpu_planner* planner, rgpu_segment* seg, int32_t op, int32_t n_queries,
                                                int32_t n_clauses, const int64_t* term_ids, int32_t k, void* hits_dev, void* total_hits_dev,
                                                void* hip_stream);
/* Any mix: ops[q] as rgpu_query.op (incl. RGPU_OP_OR_MSM / RGPU_OP_WITH_SHOULD), n_terms[q] / n_must_not[q] as in rgpu_query;
 * a query's terms follow each other in the order scored, optional SHOULD, MUST_NOT; boosts (NULL: all 1) one per term.
 * terms_cap = room in terms_out. */
int32_t rgpu_plan_batch_ids(rgpu_planner* planner, int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not_or_null,
                            const int64_t* term_ids, const float* boosts_or_null, rgpu_query* queries_out, rgpu_query_term* terms_out,
                            int64_t terms_cap);
int32_t rgpu_plan_batch_bytes(rgpu_planner* planner, int32_t n_queries, const int32_t* ops, const int32_t* n_terms, const int32_t* n_must_not_or_null,
                              const uint8_t* term_bytes, const int64_t* term_offsets, const float* boosts_or_null, rgpu_query* queries_out,
                              rgpu_query_term* terms_out, int64_t terms_cap);

/* ---- exact phrases (positions fields) ----------------------------------------------------------------------------- */
/* The ".pos" file of a segment uploaded with index_options = 3 (Lucene50PostingsReader::open, posting_reader.rs:112-158:
 * header "Lucene50PostingsWriterPos", the .doc file's version, segment id and suffix; footer). Needed before
 * rgpu_search_phrase_batch; TERM / AND / OR search of a positions field works without it. */
int32_t rgpu_segment_attach_positions(rgpu_segment* seg, const uint8_t* pos_file, size_t pos_len);
/* The ".pay" file of a segment whose field stores payloads or offsets (posting_reader.rs:131-156: header
 * "Lucene50PostingsWriterPay", the .doc file's version, segment id and suffix; footer): checked as open() checks it. No
 * search needs its bytes (see rgpu_segment_upload_field), so none are kept in HBM. */
int32_t rgpu_segment_attach_payloads(rgpu_segment* seg, const uint8_t* pay_file, size_t pay_len);
/* BlockPostingIterator::{next, next_position} over whole terms (posting_reader.rs:1285-1324 refill_positions, :1357-1380
 * next_position, :1400-1437 next; for a field with payloads / offsets the same iterator walks past them, :1291-1312): every
 * position of every doc of every given term into positions_out — terms concatenated in order (total_term_freq entries each), doc
 * after doc in doc order (rgpu_decode_terms gives the docs and the freqs that delimit them), a doc's positions ascending.
 * The parity and micro-benchmark surface of the ".pos" stream, as rgpu_decode_terms is of ".doc". At most 2^32 positions per
 * call. The _device variant leaves them in device memory (hip_stream NULL = the context's stream); both return synchronised. */
int32_t rgpu_decode_positions(rgpu_segment* seg, const rgpu_term_state* terms, const rgpu_term_positions* positions, int64_t n_terms,
                              int32_t* positions_out);
int32_t rgpu_decode_positions_device(rgpu_segment* seg, const rgpu_term_state* terms, const rgpu_term_positions* positions, int64_t n_terms,
                                     void* positions_dev, void* hip_stream);
/* One term of a phrase: its postings (BlockTermState), its position-stream pointers (rgpu_terms_lookup_positions) and
 * its position inside the phrase (PhraseQuery::build numbers them 0, 1, 2, ...; query/phrase_query.rs:60-110). */
typedef struct rgpu_phrase_term {
  rgpu_term_state state;
  rgpu_term_positions positions;
  int32_t position;
  int32_t reserved;
} rgpu_phrase_term;
/* PhraseQuery. `weight` = idf summed over the phrase's terms x boost (PhraseQuery::create_weight,
 * phrase_query.rs:136-186 -> rgpu_bm25_compute_weight with every term's doc_freq), `sim_table` as for rgpu_query_term.
 * `terms` of a query are in the QUERY's own order (PhraseQuery::new's terms / positions vectors): a sloppy phrase's scorer
 * breaks ties by that order. */
typedef struct rgpu_phrase_query {
  int32_t n_terms;     /* 2..RGPU_MAX_PHRASE_TERMS (the reference turns a one-term phrase into a TermQuery) */
  int32_t first_term;  /* index of the query's first term in `terms` */
  float weight;
  int32_t sim_table;
  int32_t slop;        /* PhraseQuery::slop: 0 = ExactPhraseScorer, > 0 = SloppyPhraseScorer (phrase_query.rs:312-331) */
  int32_t next_limit;  /* sloppy phrases only. SloppyPhraseScorer is two-phase, so BulkScorer drives it through its two-phase loop
                          (bulk_scorer.rs:97-113): every conjunction match of the phrase's terms — phrase or not, live or deleted —
                          counts, and a leaf on which more than next_limit of them went by before the first collected doc is
                          abandoned: the query then has NO hit in this segment. DefaultIndexSearcher::new(reader, next_limit):
                          0 = its default, DEFAULT_DISMATCH_NEXT_LIMIT = 500 000 (searcher.rs:47, :361); n > 0 = n; -1 = no limit;
                          RGPU_NEXT_LIMIT_ZERO = Some(0): the leaf is abandoned as soon as one approximation went by uncollected
                          (a zero-filled struct keeps meaning "the default", as it did when the field was `reserved`).
                          (ExactPhraseScorer is not two-phase in the reference — slop 0 ignores this field) */
} rgpu_phrase_query;
#define RGPU_NEXT_LIMIT_ZERO (-2)
/* One leaf of IndexSearcher::search(PhraseQuery, TopDocsCollector(k)): PhraseWeight::create_scorer (None when a term is
 * absent from the leaf) -> slop 0: ExactPhraseScorer (scorer/phrase_scorer.rs:122-294) — a doc matches when the terms occur
 * at their phrase offsets, its score is BM25(phrase frequency, norm); slop > 0: SloppyPhraseScorer (:432-1071) — the
 * reference's walk of the terms' positions through a priority queue, repeated terms and their collision handling
 * included, sloppy frequency = sum of 1 / (span + 1) over the spans within the slop, score BM25(sloppy frequency, norm).
 * Outputs as rgpu_search_batch; scores are bit-exact with the CPU scorers. A doc holding one of an exact phrase's terms more
 * than 1024 times, or a sloppy phrase's terms more than 2048 times in all -> RGPU_ERR_UNSUPPORTED. */
int32_t rgpu_search_phrase_batch(rgpu_segment* seg, const rgpu_phrase_query* queries, int32_t n_queries,
                                 const rgpu_phrase_term* terms, int32_t n_terms_total, int32_t k, rgpu_hit* hits_out,
                                 int64_t* total_hits_out);

/* ---- second-pass scoring (QueryRescorer) ------------------------------------------------------------------------------ */
/* search/scorer/rescorer.rs: QueryRescorer re-ranks the top `window_size` hits of a first pass with a second query —
 * per hit the second query's scorer is advanced to the doc and the two scores are combined (combine_score :337-352:
 * mode.combine(first * query_weight, second * rescore_weight), or first * query_weight when the second query does not
 * match), the window is sorted again (score desc, doc asc) and hits past the window take the query weight
 * (combine_docs :356-374). Rucene scores the window one doc at a time, or all at once through a Weight's BatchScorer
 * (:32-36, batch_rescore :133-229); this entry point IS the batched form: every hit of every row is scored by its
 * own wavefront. Rescore queries: TERM, all-MUST (AND) or all-SHOULD (OR) term queries. */
typedef enum rgpu_rescore_mode { RGPU_RESCORE_AVG = 0, RGPU_RESCORE_MAX = 1, RGPU_RESCORE_MIN = 2, RGPU_RESCORE_TOTAL = 3,
                                 RGPU_RESCORE_MULTIPLY = 4 } rgpu_rescore_mode;  /* RescoreMode (rescorer.rs:96-116) */
typedef struct rgpu_rescore_request {  /* RescoreRequest (rescorer.rs:67-93), one per row */
  float query_weight;
  float rescore_weight;
  int32_t mode;         /* rgpu_rescore_mode */
  int32_t window_size;  /* <= RGPU_MAX_K */
} rgpu_rescore_request;
/* hits_inout: n_queries rows of k rgpu_hit (host memory), best first, unused slots {-1, 0} — a first pass's output; on
 * return the rescored rows. Docs carry doc_base; hits of other leaves are left alone, so a multi-leaf caller runs one
 * call per leaf with finish = 0 and finish = 1 on the last: only then is the window sorted and the tail re-weighted. */
int32_t rgpu_rescore_batch(rgpu_segment* seg, const rgpu_query* queries, int32_t n_queries, const rgpu_query_term* terms,
                           int32_t n_terms_total, const rgpu_rescore_request* requests, int32_t k, rgpu_hit* hits_inout,
                           int32_t finish);

/* ---- measurement ------------------------------------------------------------------------------------------ */
typedef struct rgpu_kernel_stat {
  char name[48];
  int64_t launches;
  double total_ms;          /* HIP-event time summed over launches (needs cfg.profile_kernels = 1) */
  int64_t postings;          /* sum of doc_freq over the terms the launches covered */
  int64_t timed_launches;    /* launches whose own HIP-event duration was kept (the first 8192 per name) */
  double min_ms, median_ms, max_ms; /* over those launches: price a roofline on the MEDIAN — total_ms / launches is
                                multiplied by one stalled launch (a first-use allocation, a clock dip) */
} rgpu_kernel_stat;
int32_t rgpu_kernel_stats(rgpu_ctx* ctx, rgpu_kernel_stat* out, int32_t max_out); /* returns count */
/* Switch the HIP-event bracketing of kernels on / off (rgpu_config.profile_kernels sets the initial state): a timed
 * region runs without it, the per-kernel pass that follows with it. */
int32_t rgpu_set_profiling(rgpu_ctx* ctx, int32_t on);
/* SURVEY.md 8(d) "touched bytes" of the most recent AND launch on this context: the encoded bytes (two header bytes +
 * doc and freq payload) of every FullBlock the conjunction kernel decoded — lead blocks and the blocks of the other
 * clauses that held a pending candidate. Waits for that launch. */
int32_t rgpu_and_touched_bytes(rgpu_ctx* ctx, int64_t* bytes_out);
/* What the most recent TERM / AND / (>= 10-clause) OR launch on this context really did — as opposed to what its queries
 * cover: the TERM kernel skips every block whose (freq, norm) frontier cannot reach the top-k, the conjunction kernel
 * only visits blocks that may hold a candidate. Waits for that launch. */
typedef struct rgpu_search_counters {
  int32_t op;                /* rgpu_query_op of the launch (-1: none yet) */
  int32_t reserved;
  int64_t postings_covered;  /* sum of doc_freq over the launch's clauses: what an exhaustive scorer would iterate */
  int64_t postings_decoded;  /* 128 per FullBlock actually unpacked + prepared tails / singletons read */
  int64_t blocks_decoded;
  int64_t touched_bytes;     /* encoded bytes of those blocks; TERM: + 1 norm byte per posting of them + what the launch read of the
                                block directory (18 bytes per block of every chunk of 64 it visited, 8 per chunk frontier, 2 per sketch
                                entry), counted by the kernel. 0 for OR launches (everything is read) */
} rgpu_search_counters;
int32_t rgpu_last_search_counters(rgpu_ctx* ctx, rgpu_search_counters* out);
void rgpu_kernel_stats_reset(rgpu_ctx* ctx);
int32_t rgpu_synchronize(rgpu_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RUCENE_GPU_H */
