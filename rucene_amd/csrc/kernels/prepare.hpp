// Term preparation on the GPU: from the reference's bytes (.doc: level-0 skip entries, block framing, VInt tails) to what
// the query kernels read (block directory, 16-byte aligned block store, decoded tails) — stage A, everything a DECODE needs
// — and, only for terms that get scored, posting-order norms + block-max frontier words — stage B. GPU counterpart of
// (paths relative to /root/reference/src/core):
//   codec/postings/skip_reader.rs:460-511   load_skip_levels  (vlong length + bytes for levels L-1..1, then level 0)
//   codec/postings/skip_reader.rs:431-453   read_skip_data    (vint docDelta, vlong docFpDelta per entry)
//   codec/postings/skip_reader.rs:513-539   load_next_skip    (the running sums skip_doc / doc_pointer)
//   codec/postings/for_util.rs:196-223      block header byte (encode type, num_bits, all-equal vint)
//   codec/postings/posting_reader.rs:308-333 read_vint_block  (the tail)
// The higher skip levels are subsampled copies of level 0 with child pointers; a flat level-0 directory plus binary
// search gives the same `advance` answers, so only their byte lengths are parsed (to find level 0).
//
// Stage A is a handful of launches, each spread over the whole chip whatever the terms' sizes (round 2 walked a term's skip data
// and block headers with ONE workgroup: 3.4 ms for the 20 M-posting head term of the 100 M-doc shard, whatever else ran):
//   k_skip_dir<1,2>  one wavefront per 1 KB of level-0 bytes: parallel VInt parse; a chunk's position in the value stream
//                    and the running sums before it come from the chunks in front of it: pass 1 leaves every chunk's
//                    {count, sums}, pass 2 adds up the ones in front and writes the directory; for a term of more than 64
//                    chunks k_skip_groups runs in between: 64 chunks' aggregates become one, so that a chunk of the longest
//                    term looks back over n / 64 / 64 + 1 wavefront-wide rounds instead of n / 64
//   k_block_headers  one lane per block: header bytes -> directory header word + the block's rows in the store; every skip
//                    pointer checked against the block sizes
//   k_scan_*         exclusive prefix sum of the row counts -> each block's place in the store (one dense region per call)
//   k_prepare_blocks one wavefront per 32 blocks: the block's byte-misaligned span is staged in LDS with ALIGNED 16-byte
//                    loads (each file byte fetched once; byte-misaligned dwordx4 loads run at a third of the rate and
//                    over-fetch a KB per block) and leaves as aligned rows; every block is decoded once and validated;
//                    EF / BITSET blocks are re-packed; the term's VInt tail is decoded into 16-byte cells
// Stage B (k_prepare_norms) decodes the blocks again from the store and gathers each posting's norm byte — one cache line
// per posting for a sparse term: 17-35 x the file's size in fetches on the 100 M-doc shard, which is why it no longer
// rides on every first touch of a term: a materialising decode (rgpu_decode_terms, rgpu_advance_batch) never pays it.
#pragma once
#include "decode_terms.hpp"
#include "types.hpp"

namespace rgpu {

constexpr int PREP_THREADS = 256;
constexpr int PREP_WAVES = PREP_THREADS / 64;
constexpr int SKIP_CHUNK_BYTES = 1024;     // level-0 bytes per wavefront (16 per lane)
#ifndef RGPU_SKIP_GROUP
#define RGPU_SKIP_GROUP 64  // (a build with 1 or 2 here walks the grouped look-back on test-sized terms)
#endif
constexpr int SKIP_GROUP = RGPU_SKIP_GROUP;  // chunks per group aggregate; a term of at most this many chunks looks back directly
static_assert(SKIP_GROUP >= 1 && SKIP_GROUP <= 64, "a group is scanned by one wavefront");
constexpr int PREP_BLOCKS_PER_ITEM = 32;
constexpr int SCAN_TILE = 2048;            // row counts per workgroup of the prefix sum (8 per thread)

// err[0] = the most severe status (rgpu_status, or -101 "plan again with worst-case rows"), err[1] = the highest-numbered
// check that failed — which of this file's consistency checks it was ends up in the host's error message; err[2] = some
// block of the call is EF / BITSET encoded
__device__ __forceinline__ void flag_err(int* err, int status, int site) {
  atomicMin(err, status);
  atomicMax(err + 1, site);
}

__device__ __forceinline__ uint64_t read_vlong_serial(const uint8_t* p, int* len) {
  uint64_t v = 0;
  int i = 0;
  for (; i < 9; ++i) {
    uint64_t b = p[i];
    v |= (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) { ++i; break; }
  }
  *len = i;
  return v;
}

__device__ __forceinline__ int vint_len_serial(const uint8_t* p) {
  int i = 0;
  while (i < 4 && (p[i] & 0x80)) ++i;
  return i + 1;
}

// ---- A1: level-0 skip entries -> dir_last / dir_off (/ dir_pos) ------------------------------------------------------------
// Values per level-0 entry (skip_writer.rs:261-289), by field: 2 = {docDelta, docFpDelta}; a positions field 4: + {posFpDelta,
// posBufferUpto}; one that stores offsets but no payloads 5: + {payFpDelta}; one that stores payloads (with or without
// offsets) 6: + {payloadByteUpto, payFpDelta}. Fields 0 - 2 are deltas (running sums), field 3 is absolute; the .pay words
// (fields 4, 5) are parsed and dropped — no kernel of this path reads payload bytes or offsets.
// What a chunk tells the chunks behind it: how many VInt values END inside it and their sums by LOCAL residue (index inside
// the chunk modulo the values per entry). Which field a residue is depends on how many values precede the chunk, known once
// the chunks in front have published.
struct SkipAgg {
  uint32_t count;
  uint32_t sum[6];
  uint32_t pad;
};
static_assert(sizeof(SkipAgg) == 32, "one aggregate per 32 bytes");

// bytes a level-0 entry can take: vint docDelta <= 5, vlong docFpDelta (a block is < 16 KiB) <= 3; positions: + vlong
// posFpDelta <= 9, vint posBufferUpto (< 128) 1; payloads: + vint payloadByteUpto <= 5; payloads or offsets: + vlong payFpDelta <= 9
__host__ __device__ constexpr int skip_entry_max_bytes(int vals) { return vals == 2 ? 8 : vals == 4 ? 18 : vals == 5 ? 27 : 32; }
// A term whose level 0 fits SKIP_SMALL_BYTES (16 entries; 7 with positions: df < 2176 / 1024 — nine terms in ten of a Zipf
// vocabulary) is parsed by ONE LANE of k_skip_terms and takes no chunk: a wavefront per term for a few dozen bytes made
// k_skip_dir a launch of 330 k two-microsecond wavefronts on the 100 M-doc shard, most of its time spent starting them.
constexpr int SKIP_SMALL_BYTES = 128;
__host__ __device__ inline bool skip_is_small(int32_t n_entries, int vals) {
  return n_entries * skip_entry_max_bytes(vals) <= SKIP_SMALL_BYTES;
}
__host__ __device__ inline int64_t skip_chunks(int32_t n_entries, int vals) {
  return skip_is_small(n_entries, vals) ? 0 : ((int64_t)n_entries * skip_entry_max_bytes(vals) + SKIP_CHUNK_BYTES - 1) / SKIP_CHUNK_BYTES;
}

template <uint32_t VALS>
__device__ __forceinline__ uint32_t pick_res(const uint32_t (&v)[VALS], uint32_t i) {  // v[i], i < VALS, without a register-array index
  static_assert(VALS >= 2 && VALS <= 8, "a three-level select tree");
  const uint32_t e0 = v[0], e1 = v[1], e2 = VALS > 2 ? v[VALS > 2 ? 2 : 0] : 0u, e3 = VALS > 3 ? v[VALS > 3 ? 3 : 0] : 0u,
                 e4 = VALS > 4 ? v[VALS > 4 ? 4 : 0] : 0u, e5 = VALS > 5 ? v[VALS > 5 ? 5 : 0] : 0u, e6 = VALS > 6 ? v[VALS > 6 ? 6 : 0] : 0u,
                 e7 = VALS > 7 ? v[VALS > 7 ? 7 : 0] : 0u;
  const uint32_t a = (i & 1u) ? e1 : e0, b = (i & 1u) ? e3 : e2, c = (i & 1u) ? e5 : e4, d = (i & 1u) ? e7 : e6;
  const uint32_t lo = (i & 2u) ? b : a, hi = (i & 2u) ? d : c;
  return (i & 4u) ? hi : lo;
}
// (a - b) mod VALS for a < VALS and any b (residue arithmetic on value counts)
template <uint32_t VALS>
__device__ __forceinline__ uint32_t sub_mod(uint32_t a, uint32_t b) { return (a + VALS - b % VALS) % VALS; }

__host__ __device__ inline int64_t skip_groups(int64_t n_chunks_of_term) { return (n_chunks_of_term + SKIP_GROUP - 1) / SKIP_GROUP; }

// One lane per term: where its level 0 starts (skip_reader.rs:481-509: a vlong byte length in front of each of the levels
// L-1 .. 1; -1 when the lengths lead out of the file) — up to six dependent loads which every chunk of the term used to walk
// again in both passes of k_skip_dir —, the directory words no skip entry writes, and, for a SMALL term, its whole level 0:
// the lane copies SKIP_SMALL_BYTES into LDS (dword-interleaved across the lanes: lane L's dword i is word 64 i + L, so the
// lanes' serial walks do not collide on banks) and reads entry after entry (skip_reader.rs:431-453, 513-539).
__global__ __launch_bounds__(PREP_THREADS) void k_skip_terms(const uint8_t* __restrict__ doc, int64_t doc_len, const PrepTerm* __restrict__ terms,
                                                             int n_terms, int64_t* __restrict__ l0s, int32_t* dir_last, uint32_t* dir_off,
                                                             uint64_t* dir_pos, int vals, int* err) {
  __shared__ uint32_t slab[PREP_WAVES][(SKIP_SMALL_BYTES / 4) * 64];
  const int i = (int)(blockIdx.x * PREP_THREADS + threadIdx.x);
  if (i >= n_terms) return;
  const int lane = lane_id();
  const PrepTerm T = terms[i];
  int64_t l0 = T.n_entries > 0 ? T.skip_fp : -1;
  for (int lvl = T.n_levels - 1; lvl >= 1 && l0 >= 0; --lvl) {
    int n;
    const uint64_t len = read_vlong_serial(doc + l0, &n);
    l0 += n + (int64_t)len;
    if (l0 >= doc_len) l0 = -1;
  }
  l0s[i] = l0;
  dir_off[T.dir_base] = 0;
  if (dir_pos) dir_pos[T.dir_base] = 0ull;
  if (T.nblocks > T.n_entries && T.nblocks > 0) dir_last[T.dir_base + T.nblocks - 1] = DIR_SENTINEL_DOC;  // df % 128 == 0
  if (T.n_entries <= 0 || !skip_is_small(T.n_entries, vals)) return;
  if (l0 < 0) { flag_err(err, -4, 1); return; }
  uint32_t* const my = slab[wave_id()] + lane;
#pragma unroll
  for (int k = 0; k < SKIP_SMALL_BYTES / 16; ++k) {  // (the device copy of .doc is zero-padded for 8 KB past doc_len)
    const uint4 w = load16_unaligned(doc + l0 + 16 * k);
    my[64 * (4 * k)] = w.x; my[64 * (4 * k + 1)] = w.y; my[64 * (4 * k + 2)] = w.z; my[64 * (4 * k + 3)] = w.w;
  }
  int p = 0;
  bool over = false;
  auto vnum = [&](int max_bytes) -> uint32_t {  // VInt / the low 32 bits of a VLong (pointers of a < 4 GiB term, like k_skip_dir)
    uint64_t v = 0;
    for (int k = 0; k < max_bytes; ++k) {
      if (p >= SKIP_SMALL_BYTES) { over = true; break; }
      const uint32_t b = (my[64 * (p >> 2)] >> (8 * (p & 3))) & 0xffu;
      ++p;
      v |= (uint64_t)(b & 0x7fu) << (7 * k);
      if (!(b & 0x80u)) break;
    }
    return (uint32_t)v;
  };
  uint32_t run_doc = 0, run_fp = 0, run_pos = 0;
  for (int e = 0; e < T.n_entries; ++e) {
    run_doc += vnum(5);                                               // skip_doc += delta (skip_reader.rs:530)
    run_fp += vnum(9);                                                // doc_pointer += delta (:434)
    dir_last[T.dir_base + e] = (int32_t)run_doc;
    dir_off[T.dir_base + e + 1] = run_fp;
    if (dir_pos) {
      run_pos += vnum(9);                                             // pos_pointer += delta
      const uint32_t upto = vnum(5);                                  // posBufferUpto: absolute
      if (upto >= 128u) flag_err(err, -4, 4);
      dir_pos[T.dir_base + e + 1] = (uint64_t)run_pos | ((uint64_t)upto << 32);
      if (vals == 6) (void)vnum(5);                                   // payloadByteUpto
      if (vals >= 5) (void)vnum(9);                                   // pay_pointer += delta
    }
  }
  if (over) flag_err(err, -4, 2);  // an entry longer than any the writer produces: ran off the bytes taken
}

// A run of aggregates, front to back, folded into (values before, running sums by FIELD — the delta fields 0 .. 2): aggregate
// j holds its sums by residue relative to ITS first value, which is field (values before j) mod VALS.
template <uint32_t VALS>
__device__ __forceinline__ void fold_aggs(const SkipAgg* a, int n, int lane, uint32_t& P, uint32_t (&base)[3]) {
  constexpr uint32_t NF = VALS < 3u ? VALS : 3u;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    uint32_t cj = 0, sj[VALS];
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) sj[r] = 0u;
    if (j < n) {
      cj = a[j].count;
#pragma unroll
      for (uint32_t r = 0; r < VALS; ++r) sj[r] = a[j].sum[r];
    }
    const uint32_t inc = (uint32_t)wave_incl_scan((int)cj);
    const uint32_t Pj = P + inc - cj;  // values before aggregate j: its local residue r is field (Pj + r) mod VALS
#pragma unroll
    for (uint32_t f = 0; f < NF; ++f) base[f] += (uint32_t)wave_reduce_add((int)pick_res<VALS>(sj, sub_mod<VALS>(f, Pj)));
    P += (uint32_t)readlane((int)inc, 63);
  }
}

// Between the passes, for the terms of more than SKIP_GROUP chunks (`group_prefix` counts groups for those terms only): one
// wavefront per SKIP_GROUP consecutive chunks of a term. Each chunk's aggregate is REPLACED by what precedes it inside its group (count and sums by residue relative to the
// group's first value), and the group's own aggregate goes to `gaggs`: pass 2 then folds the groups in front of its chunk's
// group and adds one slot. Without it the last chunk of a 20 M-posting term folds 1220 aggregates in 20 dependent rounds —
// and a 2 G-posting term's would take 1900: the look-back of a term is quadratic in its length, this makes it n^2 / 64.
template <uint32_t VALS>
__global__ __launch_bounds__(PREP_THREADS) void k_skip_groups(const int64_t* __restrict__ chunk_prefix, const int64_t* __restrict__ group_prefix,
                                                              int n_terms, int64_t n_groups, SkipAgg* aggs, SkipAgg* gaggs) {
  const int lane = lane_id();
  const int64_t s = (int64_t)blockIdx.x * PREP_WAVES + wave_id();
  if (s >= n_groups) return;
  const int t = upper_slot_wave(group_prefix, n_terms, s, lane);
  const int n_mine = (int)(chunk_prefix[t + 1] - chunk_prefix[t]);
  const int j = (int)(s - group_prefix[t]) * SKIP_GROUP + lane;
  const bool have = lane < SKIP_GROUP && j < n_mine;
  SkipAgg* const a = aggs + chunk_prefix[t] + j;
  uint32_t cj = 0, sj[VALS];
#pragma unroll
  for (uint32_t r = 0; r < VALS; ++r) sj[r] = 0u;
  if (have) {
    cj = a->count;
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) sj[r] = a->sum[r];
  }
  const uint32_t inc = (uint32_t)wave_incl_scan((int)cj);
  const uint32_t Pj = inc - cj;
  uint32_t ex[VALS], tot[VALS];
#pragma unroll
  for (uint32_t f = 0; f < VALS; ++f) {
    const uint32_t r = pick_res<VALS>(sj, sub_mod<VALS>(f, Pj));  // this chunk's sum for residue f of the GROUP
    const uint32_t in = (uint32_t)wave_incl_scan((int)r);
    ex[f] = in - r;
    tot[f] = (uint32_t)readlane((int)in, 63);
  }
  if (have) {
    a->count = Pj;
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) a->sum[r] = ex[r];
  }
  if (lane == 0) {
    SkipAgg* const g = gaggs + s;
    g->count = (uint32_t)readlane((int)inc, 63);
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) g->sum[r] = tot[r];
  }
}

// PASS selects the half of the job: 1 = parse this chunk and leave its aggregate; 2 = parse it again (cheaper than keeping
// the values anywhere), take the aggregates of the chunks in front of it — complete: they were written by the launches
// before — and write the directory. (One launch with the chunks waiting for each other was tried first: the wavefronts of a
// 20 M-posting term's 1220 chunks spinning on acquire loads cost 1 - 4 ms, varying from run to run.)
template <int PASS, uint32_t VALS>
__global__ __launch_bounds__(PREP_THREADS) void k_skip_dir(const uint8_t* __restrict__ doc, int64_t doc_len, int64_t doc_cap,
                                                           const PrepTerm* __restrict__ terms, const int64_t* __restrict__ chunk_prefix,
                                                           const int64_t* __restrict__ l0s, int n_terms, int64_t n_chunks, SkipAgg* aggs,
                                                           const int64_t* __restrict__ group_prefix, const SkipAgg* __restrict__ gaggs,
                                                           int32_t* dir_last, uint32_t* dir_off, uint64_t* dir_pos, int* err) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[PREP_WAVES][16 + SKIP_CHUNK_BYTES];
  constexpr uint32_t NF = VALS < 3u ? VALS : 3u;  // the delta fields: doc, doc pointer, position pointer
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * PREP_WAVES + wave;
  if (item >= n_chunks) return;
  const int t = upper_slot_wave(chunk_prefix, n_terms, item, lane);
  const int c = (int)(item - chunk_prefix[t]);
  const int n_mine = (int)(chunk_prefix[t + 1] - chunk_prefix[t]);
  const PrepTerm T = terms[t];
  const uint32_t need = VALS * (uint32_t)T.n_entries;
  SkipAgg* const mine = aggs + item;
  // ---- where level 0 starts: k_skip_terms walked the level headers once per term (and wrote the term's first directory words)
  const int64_t l0 = l0s[t];
  if (l0 < 0) {
    if (PASS == 2 && lane == 0 && c == 0) flag_err(err, -4, 1);
    if (PASS == 1 && lane == 0) {
      mine->count = 0;
#pragma unroll
      for (uint32_t r = 0; r < VALS; ++r) mine->sum[r] = 0u;
    }
    return;
  }
  // ---- this chunk's bytes (and the 16 in front of it: a value may begin there) into LDS
  const int64_t at = l0 + (int64_t)SKIP_CHUNK_BYTES * c;
  uint8_t* const st = stage[wave];
  uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
  if (at + 16 * lane + 16 <= doc_cap) w4 = load16_unaligned(doc + at + 16 * lane);  // (the device copy is zero-padded past doc_len)
  *reinterpret_cast<uint4*>(st + 16 + 16 * lane) = w4;
  if (lane == 0) {
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);  // chunk 0: level 0 starts here — nothing in front of it continues into it
    if (c > 0) pre = load16_unaligned(doc + at - 16);
    *reinterpret_cast<uint4*>(st) = pre;
  }
  wave_sync();
  const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
  uint32_t term = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) term |= (((ws[j >> 2] >> (8 * (j & 3) + 7)) & 1u) ^ 1u) << j;
  const int cnt = __popc(term);
  const int incl = wave_incl_scan(cnt);
  const uint32_t total = (uint32_t)readlane(incl, 63);
  const uint32_t li0 = (uint32_t)(incl - cnt);  // local index of this lane's first value
  // the value whose terminator is byte j of this lane's 16: look back over its continuation bytes (vlongs: the low 32 bits
  // are kept, like the reference's `as i32` / u32 pointers of a < 4 GiB term)
  auto value_at = [&](int j) -> uint32_t {
    int p = 16 + 16 * lane + j;
    uint64_t v = st[p];
    for (int back = 0; back < 9 && p > 0 && (st[p - 1] & 0x80); ++back) {
      --p;
      v = (v << 7) | (uint64_t)(st[p] & 0x7f);
    }
    return (uint32_t)v;
  };
  // ---- sums by local residue
  uint32_t sl[VALS];
#pragma unroll
  for (uint32_t r = 0; r < VALS; ++r) sl[r] = 0u;
  {
    uint32_t m = term, res = li0 % VALS;
    while (m) {
      const int j = __builtin_ctz(m);
      m &= m - 1;
      const uint32_t v = value_at(j);
#pragma unroll
      for (uint32_t r = 0; r < VALS; ++r) sl[r] += res == r ? v : 0u;
      res = res + 1u == VALS ? 0u : res + 1u;
    }
  }
  if (PASS == 1) {
    uint32_t red[VALS];
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) red[r] = (uint32_t)wave_reduce_add((int)sl[r]);
    if (lane == 0) {
      mine->count = total;
#pragma unroll
      for (uint32_t r = 0; r < VALS; ++r) mine->sum[r] = red[r];
    }
    return;
  }
  // ---- the chunks in front of this one, front to back: values before this chunk (P) and the running sums by field
  uint32_t P = 0;
  uint32_t base[3] = {0u, 0u, 0u};
  if (n_mine <= SKIP_GROUP) {
    fold_aggs<VALS>(aggs + (item - c), c, lane, P, base);
  } else {
    // the groups in front of this chunk's group, then what k_skip_groups left in this chunk's own slot: the chunks in
    // front of it inside its group, by residue relative to the group's first value = field (P + residue) mod VALS
    fold_aggs<VALS>(gaggs + group_prefix[t], c / SKIP_GROUP, lane, P, base);
    const SkipAgg* a = aggs + item;
    uint32_t in[VALS];
#pragma unroll
    for (uint32_t r = 0; r < VALS; ++r) in[r] = a->sum[r];
#pragma unroll
    for (uint32_t f = 0; f < NF; ++f) base[f] += pick_res<VALS>(in, sub_mod<VALS>(f, P));
    P += a->count;
  }
  if (c == n_mine - 1 && P + total < need && lane == 0) flag_err(err, -4, 2);  // ran off the skip data looking for entries
  // ---- every value's running sum -> the directory
  uint32_t run[3] = {0u, 0u, 0u};
  {
    // this lane's sums by FIELD, then the lanes in front of it
#pragma unroll
    for (uint32_t f = 0; f < NF; ++f) {
      const uint32_t lf = pick_res<VALS>(sl, sub_mod<VALS>(f, P));
      run[f] = base[f] + (uint32_t)wave_incl_scan((int)lf) - lf;
    }
  }
  uint32_t m = term, g = P + li0, fld = g % VALS;
  while (m) {
    const int j = __builtin_ctz(m);
    m &= m - 1;
    const uint32_t v = value_at(j), f = fld;
    run[0] += f == 0u ? v : 0u; run[1] += f == 1u ? v : 0u; run[2] += f == 2u ? v : 0u;
    if (g < need) {
      const uint32_t e = g / VALS;
      if (f == 0u) dir_last[T.dir_base + e] = (int32_t)run[0];             // skip_doc += delta (skip_reader.rs:530)
      else if (f == 1u) dir_off[T.dir_base + e + 1] = run[1];             // doc_pointer += delta (:434)
      else if (f == 2u) reinterpret_cast<uint32_t*>(dir_pos + T.dir_base + e + 1)[0] = run[2];  // pos_pointer += delta
      else if (f == 3u) { reinterpret_cast<uint32_t*>(dir_pos + T.dir_base + e + 1)[1] = v; if (v >= 128u) flag_err(err, -4, 4); }  // posBufferUpto: absolute
      // (fields 4, 5 — payloadByteUpto, the .pay pointer — are read past: see the top of this section)
    }
    ++g;
    fld = fld + 1u == VALS ? 0u : fld + 1u;
  }
}

// ---- A2: block headers (for_util.rs:196-223) + consistency of every skip pointer with the block sizes ------------------------
// One LANE per directory slot of the call (a term's slots are consecutive: its FullBlocks, then the slot of its tail), whatever
// the terms' sizes: a wavefront per (term, 32 blocks) was 200 k wavefronts for the 1.9 M blocks of the 100 M-doc shard, nine in
// ten of them with a handful of busy lanes (0.145 ms; slot-major: 0.079 — the header bytes of 355-byte blocks are a cache line
// each, 0.43 GB fetched for 4 MB read). Leaves the block's ROW COUNT in dir_row
// (k_scan_* turn counts into positions).
// largest t in [0, n) with terms[t].dir_base <= slot, searched by the whole wavefront (upper_slot_wave over PrepTerm::dir_base)
__device__ __forceinline__ int upper_term_wave(const PrepTerm* __restrict__ terms, int n, uint32_t slot, int lane) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const bool ok = idx < hi && terms[idx].dir_base <= slot;  // true on a prefix of the lanes, lane 0 included
    const int cnt = __popcll(__ballot(ok));
    lo += (cnt - 1) * step;
    hi = min(hi, lo + step);
  }
  return lo;
}
template <bool LEGACY>
__global__ __launch_bounds__(PREP_THREADS) void k_block_headers(const uint8_t* __restrict__ doc, int64_t doc_len,
                                                                const PrepTerm* __restrict__ terms, int n_terms, uint32_t slot0,
                                                                int64_t n_slots, const int32_t* __restrict__ dir_last,
                                                                const uint32_t* __restrict__ dir_off, uint32_t* dir_row, uint16_t* dir_hdr,
                                                                int has_freqs, int* err) {
  const int lane = lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * PREP_WAVES + wave_id()) * 64;
  if (w0 >= n_slots) return;
  const uint32_t first = slot0 + (uint32_t)w0, slot = first + (uint32_t)lane;
  // the term of the wavefront's first slot, then this lane's: a term has at least one slot, so the terms that begin inside
  // these 64 slots are among the 63 behind the first one
  const int t0 = upper_term_wave(terms, n_terms, first, lane);
  const uint32_t begins = t0 + lane < n_terms ? terms[t0 + lane].dir_base : 0xffffffffu;
  int ti = t0;
#pragma unroll
  for (int j = 1; j < 64; ++j) ti += (uint32_t)readlane((int)begins, j) <= slot ? 1 : 0;
  if (w0 + lane >= n_slots) return;
  struct { uint64_t start_fp; uint32_t dir_base; int32_t nblocks, n_entries, df; } t;
  t.start_fp = terms[ti].start_fp; t.dir_base = terms[ti].dir_base; t.nblocks = terms[ti].nblocks; t.n_entries = terms[ti].n_entries;
  t.df = terms[ti].df;
  const int i = (int)(slot - t.dir_base);
  if (i >= t.nblocks) {  // where the term's decoded tail goes
    dir_row[slot] = (t.df > 1 && t.df % 128 != 0) ? (uint32_t)TAIL_STORE_ROWS : 0u;
    return;
  }
  const uint32_t off = dir_off[t.dir_base + i];
  // offsets are running sums of deltas nobody has checked yet: a block (<= 2 + 2 * 512 bytes) must start inside the file
  if ((uint64_t)t.start_fp + (uint64_t)off + 1030u > (uint64_t)doc_len + 4096u) {
    flag_err(err, -4, 5);
    dir_hdr[t.dir_base + i] = 0;
    dir_row[t.dir_base + i] = 2u;
    return;
  }
  const uint8_t* p = doc + t.start_fp + off;
  const uint32_t h = p[0];
  int bd = (int)(h & 63);
  int vlen = 0;
  const int etype = (int)(h >> 6);
  int doc_sz = 16 * bd;
  uint32_t flag = 0;
  if (etype != 0) {  // EF / BITSET doc block (decode.hpp): sized here, decoded and re-packed by k_prepare_blocks
    if (etype == 3 || LEGACY) flag_err(err, -5, 6);  // FULL is unimplemented in the reference; EF + the legacy layout: not served
    doc_sz = etype == 3 ? 0 : nonpf_doc_bytes(p, etype);
    if (doc_sz < 0) { flag_err(err, -4, 7); doc_sz = 0; }
    bd = 32;       // doc rows reserved in the block store
    vlen = etype;  // the vint-length field is free in a flagged word
    flag = HDR_NONPF;
    atomicMax(err + 2, 1);
  } else {
    if (bd > 32) { flag_err(err, -4, 8); bd = 32; doc_sz = 512; }
    if (bd == 0) { vlen = vint_len_serial(p + 1); doc_sz = vlen; }
  }
  // the freq block (absent for IndexOptions::Docs: posting_writer.rs:334-351 writes it only when the field has freqs;
  // the directory then says "all-equal freq stream" and the block store supplies the value 1)
  int bf = 0;
  uint32_t end = off + 1u + (uint32_t)doc_sz;
  if (has_freqs) {
    const uint32_t h2 = p[1 + doc_sz];
    bf = (int)(h2 & 63);
    if (bf > 32) { flag_err(err, -4, 9); bf = 32; }
    const int freq_sz = bf ? 16 * bf : vint_len_serial(p + 1 + doc_sz + 1);
    end += 1u + (uint32_t)freq_sz;
  }
  if (i < t.n_entries && end != dir_off[t.dir_base + i + 1]) flag_err(err, -4, 10);
  if (i > 0 && i < t.n_entries && dir_last[t.dir_base + i] <= dir_last[t.dir_base + i - 1]) flag_err(err, -4, 11);
  const uint32_t hdr = (uint32_t)bd | ((uint32_t)vlen << 6) | ((uint32_t)bf << 9) | flag;
  dir_hdr[t.dir_base + i] = (uint16_t)hdr;
  dir_row[t.dir_base + i] = (uint32_t)(store_doc_rows(hdr) + store_freq_rows(hdr));
}

// ---- A3: exclusive prefix sum of the row counts of one call's directory slots -------------------------------------------------
// (reduce per tile, scan the tile sums in one workgroup, scan inside the tiles) — rows are positions in ONE dense region
// of the block store per call, relative to its first byte (PrepTerm::bs_base is that byte for every term of the call)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_sums, uint32_t& total) {
  const int lane = lane_id();
  const int wave = wave_id();
  const uint32_t incl = (uint32_t)wave_incl_scan((int)v);
  __syncthreads();  // protect wave_sums reuse
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PREP_WAVES; ++w) {
    const uint32_t s = wave_sums[w];
    if (w < wave) off += s;
    tot += s;
  }
  total = tot;
  return off + incl - v;
}
__global__ __launch_bounds__(PREP_THREADS) void k_scan_reduce(const uint32_t* __restrict__ v, int64_t n, unsigned long long* tile_sums) {
  __shared__ uint32_t s_ws[PREP_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + 8 * (int64_t)threadIdx.x;
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) mine += base + j < n ? v[base + j] : 0u;
  uint32_t tot;
  (void)block_excl_scan(mine, s_ws, tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
// tile sums -> exclusive; out2[0] = the call's total rows. cap_rows = rows the host reserved: more than that is corrupt
// framing — or EF / BITSET blocks whose re-packed deltas outgrow their file bytes: -101 asks for a worst-case plan
__global__ __launch_bounds__(PREP_THREADS) void k_scan_tiles(unsigned long long* tile_sums, int64_t n_tiles, unsigned long long cap_rows,
                                                             unsigned long long* out2, int* err) {
  __shared__ uint32_t s_ws[PREP_WAVES];
  unsigned long long carry = 0;
  for (int64_t i0 = 0; i0 < n_tiles; i0 += PREP_THREADS) {
    const int64_t i = i0 + (int64_t)threadIdx.x;
    const unsigned long long mine = i < n_tiles ? tile_sums[i] : 0ull;  // (< 2^32 per tile: at most 2048 x 64 rows)
    uint32_t tot;
    const uint32_t ex = block_excl_scan((uint32_t)mine, s_ws, tot);
    if (i < n_tiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    out2[0] = carry;
    if (carry > cap_rows || carry > 0xfffffff0ull) flag_err(err, (err[2] != 0 && carry <= 0xfffffff0ull) ? -101 : -4, 12);
  }
}
__global__ __launch_bounds__(PREP_THREADS) void k_scan_down(uint32_t* v, int64_t n, const unsigned long long* __restrict__ tile_sums) {
  __shared__ uint32_t s_ws[PREP_WAVES];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + 8 * (int64_t)threadIdx.x;
  uint32_t x[8], mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = base + j < n ? v[base + j] : 0u; mine += x[j]; }
  uint32_t tot;
  uint32_t at = block_excl_scan(mine, s_ws, tot) + (uint32_t)tile_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j) { if (base + j < n) v[base + j] = at; at += x[j]; }
}

// ---- A4: payload rows -> block store, validation, tails -----------------------------------------------------------------------
// this lane's 16-byte row of its stream out of the block's bytes staged at `slab` (the block starts at byte `mis`): a
// misaligned 16 bytes = five aligned dwords shifted into place
__device__ __forceinline__ uint4 staged_file_row(const uint8_t* slab, uint32_t mis, uint32_t hdr, int lane) {
  const int bd = hdr_bdoc(hdr);
  const uint32_t doc_sz = bd ? 16u * (uint32_t)bd : (uint32_t)hdr_vlen(hdr);
  const uint32_t o = mis + 1u + 16u * (uint32_t)(lane & 31) + __umul24((uint32_t)(lane >> 5), doc_sz + 1u);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(slab + (o & ~3u));
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
  const uint32_t sh = o & 3u;
  return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh),
                    __builtin_amdgcn_alignbyte(w4, w3, sh));
}
constexpr int PREP_SLAB_BYTES = SLAB_BYTES;  // >= 15 + 1026 + 20 staged block bytes; the tail decoder's scratch
static_assert(PREP_SLAB_BYTES >= 1104, "a staged block: 66 aligned rows + the over-read of the last misaligned row");

template <bool LEGACY>
__global__ __launch_bounds__(PREP_THREADS) void k_prepare_blocks(const uint8_t* __restrict__ doc, const PrepTerm* __restrict__ terms,
                                                                  const int64_t* __restrict__ item_prefix, int n_terms,
                                                                  int64_t n_items, int32_t* dir_last,
                                                                  const uint32_t* __restrict__ dir_off,
                                                                  const uint32_t* __restrict__ dir_row,
                                                                  uint16_t* dir_hdr, uint8_t* bstore,
                                                                  uint64_t* __restrict__ dir_bmax, int has_freqs,
                                                                  int32_t max_doc, int* err, int32_t* __restrict__ docs_out,
                                                                  int32_t* __restrict__ freqs_out) {
  // docs_out / freqs_out (nullable): the materialising decode that triggered this preparation (PrepTerm::out_base) — a
  // first-touch decode then costs no second pass over the block store (k_decode_terms serves terms prepared earlier)
  __shared__ __attribute__((aligned(16))) uint8_t slabs[PREP_WAVES][PREP_SLAB_BYTES];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * PREP_WAVES + wave;
  if (item >= n_items || *err != 0) return;  // a term whose framing did not check out must not be walked
  const int ti = upper_slot_wave(item_prefix, n_terms, item, lane);
  const PrepTerm t = terms[ti];
  const int b0 = (int)(item - item_prefix[ti]) * PREP_BLOCKS_PER_ITEM;
  const int b1 = min(t.nblocks, b0 + PREP_BLOCKS_PER_ITEM);
  uint8_t* term_rows = bstore + t.bs_base;
  uint8_t* slab = slabs[wave];
  // the term's last item also takes its VInt tail (posting_reader.rs:308-333): decoded here once, checked like the blocks
  // (doc ids strictly increasing from the last FullBlock's last doc, inside the segment) and stored as 16-byte cells (tail_load)
  const int tail_n = t.df > 1 ? t.df % 128 : 0;
  if (tail_n > 0 && b0 + PREP_BLOCKS_PER_ITEM >= t.nblocks) {
    const uint32_t toff = t.nblocks ? dir_off[t.dir_base + t.nblocks] : 0u;
    const int32_t tbase = t.nblocks ? dir_last[t.dir_base + t.nblocks - 1] : 0;
    int32_t d0, d1;
    uint32_t f0, f1;
    decode_tail(doc + t.start_fp + toff, tail_n, tbase, slab, lane, d0, d1, f0, f1, has_freqs != 0);
    const bool v0 = 2 * lane < tail_n, v1 = 2 * lane + 1 < tail_n;
    const int32_t prev = __builtin_amdgcn_update_dpp(tbase, d1, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 <- the base doc
    const bool first_ok = (t.nblocks == 0 && lane == 0) ? d0 >= 0 : d0 > prev;  // a term's very first doc may be doc 0
    const bool bad = (v0 && (!first_ok || d0 >= max_doc)) || (v1 && (d1 <= d0 || d1 >= max_doc));
    if (__ballot(bad)) { if (lane == 0) flag_err(err, -4, 13); return; }
    uint8_t* tp = term_rows + 16 * (size_t)dir_row[t.dir_base + t.nblocks];
    *reinterpret_cast<uint4*>(tp + 16 * lane) = make_uint4(v0 ? (uint32_t)d0 : 0x7fffffffu, v1 ? (uint32_t)d1 : 0x7fffffffu, v0 ? f0 : 0u, v1 ? f1 : 0u);
    // the tail's directory slot gets its last doc, like a FullBlock's: the wide OR kernel walks tails as one more block
    const int32_t tail_last = readlane(((tail_n - 1) & 1) ? d1 : d0, (tail_n - 1) >> 1);
    if (lane == 0) { dir_last[t.dir_base + t.nblocks] = tail_last; dir_bmax[t.dir_base + t.nblocks] = 15ull; }
    if (docs_out != nullptr && t.out_base >= 0) {
      const int64_t o = t.out_base + 128 * (int64_t)t.nblocks + 2 * lane;
      if (v0) { docs_out[o] = d0; freqs_out[o] = (int32_t)f0; }
      if (v1) { docs_out[o + 1] = d1; freqs_out[o + 1] = (int32_t)f1; }
    }
  }
  if (b1 <= b0) return;
  // lane j: block b0 + j's directory words (one coalesced look instead of dependent loads per block). (Asking for them and
  // for the first block's bytes BEFORE the tail above is decoded — nine items in ten are a small term's only one — changed
  // nothing: 1.09 ms against 1.02 - 1.10 on the 100 M-doc shard. The kernel moves 3.5 GB, most of it written, in that time.)
  const int nb = b1 - b0;
  const bool mine = lane < nb;
  const uint32_t my_off = mine ? dir_off[t.dir_base + b0 + lane] : 0u;
  const uint32_t my_hdr = mine ? (uint32_t)dir_hdr[t.dir_base + b0 + lane] : 0u;
  const uint32_t my_row = mine ? dir_row[t.dir_base + b0 + lane] : 0u;
  const int32_t my_last = mine ? dir_last[t.dir_base + b0 + lane] : 0;  // (a dependent load per block would be ~1.5 us of nothing else to do)
  int32_t base = b0 == 0 ? 0 : dir_last[t.dir_base + b0 - 1];
  // a block's aligned rows: [a, a + 16 * n16) covers its bytes; lane l takes row l (rows 64, 65 — a 1026-byte block that
  // starts late in its first row — ride on lanes 0, 1)
  struct Staged { uint4 r0, r1; };
  auto request = [&](int j) -> Staged {
    const uint32_t hdr = (uint32_t)readlane((int)my_hdr, j);
    const uint64_t p = t.start_fp + (uint64_t)(uint32_t)readlane((int)my_off, j);
    const uint32_t mis = (uint32_t)(p & 15u);
    const uint32_t bytes = hdr_nonpf(hdr) ? 0u : mis + encoded_block_bytes(hdr) + 3u;  // (+3: the last row's fifth dword)
    const uint8_t* a = doc + (p - mis);
    Staged s;
    s.r0 = make_uint4(0u, 0u, 0u, 0u);
    s.r1 = s.r0;
    if (16u * (uint32_t)lane < bytes) s.r0 = *reinterpret_cast<const uint4*>(a + 16 * lane);
    if (1024u + 16u * (uint32_t)lane < bytes) s.r1 = *reinterpret_cast<const uint4*>(a + 1024 + 16 * lane);
    return s;
  };
  Staged cur = request(0);
  for (int j = 0; j < nb; ++j) {
    const int blk = b0 + j;
    uint32_t hdr = (uint32_t)readlane((int)my_hdr, j);
    const uint32_t row0 = (uint32_t)readlane((int)my_row, j);
    const uint32_t off = (uint32_t)readlane((int)my_off, j);
    // the next block's bytes are in flight while this one is re-laid (two blocks ahead: 78 VGPRs, 6 wavefronts per SIMD
    // instead of 7, and 8 % slower on both the 10 M- and the 50 M-doc shard)
    const Staged nxt = request(min(j + 1, nb - 1));
    uint4 rows;
    if (!hdr_nonpf(hdr)) {
      const uint32_t mis = (uint32_t)((t.start_fp + off) & 15u);
      *reinterpret_cast<uint4*>(slab + 16 * lane) = cur.r0;
      if (lane < 5) *reinterpret_cast<uint4*>(slab + 1024 + 16 * lane) = cur.r1;
      wave_sync();
      rows = store_rows_from_file(staged_file_row(slab, mis, hdr, lane), hdr, lane, has_freqs != 0);
      wave_sync();
    } else {
      // ---- an EF / BITSET doc block: 128 doc ids -> deltas -> BP128 rows (decode.hpp; for_util.rs:337-372,
      // posting_reader.rs:622-637, elias_fano_decoder.rs:95-169, util/bit_set.rs:351-376)
      const uint8_t* p = doc + t.start_fp + off;
      const int etype = hdr_vlen(hdr);
      int32_t* ids = reinterpret_cast<int32_t*>(slab);            // 128 doc ids
      uint32_t* pack = reinterpret_cast<uint32_t*>(slab + 512);   // 132 dwords of BP128 rows
      const int32_t pf_base = base;  // the last doc of the block before this one (0 for a term's first block)
      int doc_sz, total;
      if (etype == 2) {
        int v = 1;
        while (v < 5 && (p[v] & 0x80)) ++v;
        int vl;
        const int32_t min_doc = (int32_t)read_vint_uniform(p + 1, &vl);
        const int nw = p[1 + v];
        doc_sz = v + 1 + 8 * nw;
        uint64_t word = 0;
        if (lane < nw) { const uint8_t* wp = p + 2 + v + 8 * lane; word = (uint64_t)load4_unaligned(wp) | ((uint64_t)load4_unaligned(wp + 4) << 32); }
        const int cnt = __popcll(word);
        const int incl = wave_incl_scan(cnt);
        total = readlane(incl, 63);
        int at = incl - cnt;
        while (word) {
          if (at < 128) ids[at] = min_doc + 64 * lane + (int)__builtin_ctzll(word);
          ++at;
          word &= word - 1;
        }
      } else {
        const EfShape e = ef_shape(p + 1);
        doc_sz = e.vlen + 8 * (e.upper_longs + e.lower_longs + e.index_longs);
        const uint8_t* up = p + 1 + e.vlen;
        const uint8_t* lp = up + 8 * e.upper_longs;
        const int32_t ef_base = blk == 0 ? -1 : pf_base;  // refill_docs: ef_base_doc = accum when accum > 0, else -1
        uint64_t word = 0;
        if (lane < e.upper_longs) word = (uint64_t)load4_unaligned(up + 8 * lane) | ((uint64_t)load4_unaligned(up + 8 * lane + 4) << 32);
        const int cnt = __popcll(word);
        const int incl = wave_incl_scan(cnt);
        total = readlane(incl, 63);
        int i = incl - cnt;  // index of this lane's first value
        const uint64_t lmask = e.low_bits ? (~0ull >> (64 - e.low_bits)) : 0ull;
        while (word) {
          const int64_t high = 64 * lane + (int)__builtin_ctzll(word) - i;  // set bit position - index (current_high_value)
          uint64_t low = 0;
          if (e.low_bits) {  // unpack_value
            const int bit_pos = i * e.low_bits, at = bit_pos & 63;
            const uint8_t* wp = lp + 8 * (bit_pos >> 6);
            low = ((uint64_t)load4_unaligned(wp) | ((uint64_t)load4_unaligned(wp + 4) << 32)) >> at;
            if (at + e.low_bits > 64) low |= ((uint64_t)load4_unaligned(wp + 8) | ((uint64_t)load4_unaligned(wp + 12) << 32)) << (64 - at);
            low &= lmask;
          }
          if (i < 128) ids[i] = (int32_t)(((uint64_t)high << e.low_bits) | low) + 1 + ef_base;
          ++i;
          word &= word - 1;
        }
      }
      pack[lane] = 0u; pack[64 + lane] = 0u;
      if (lane < 4) pack[128 + lane] = 0u;
      wave_sync();
      if (total != 128) { if (lane == 0) flag_err(err, -4, 14); return; }
      const int32_t a = ids[2 * lane], b = ids[2 * lane + 1];
      const int32_t before = lane == 0 ? pf_base : ids[2 * lane - 1];
      const uint32_t d0 = (uint32_t)(a - before), d1 = (uint32_t)(b - a);
      const uint32_t top = wave_reduce_max_u32(d0 | d1);  // bits_required looks at the OR: the same most significant bit as the maximum
      const int bits = top == 0u ? 1 : 32 - __builtin_clz(top);
      auto put = [&](int i, uint32_t v) {  // SIMD128Packer::pack for one value (packed_simd.rs:81-108)
        const int bit = (i >> 2) * bits, w = bit >> 5, sft = bit & 31, l = i & 3;
        atomicOr(&pack[4 * w + l], v << sft);
        if (sft + bits > 32) atomicOr(&pack[4 * (w + 1) + l], v >> (32 - sft));
      };
      put(2 * lane, d0);
      put(2 * lane + 1, d1);
      wave_sync();
      hdr = (uint32_t)bits | (hdr & (63u << 9));  // from here on an ordinary packed-delta block
      if (lane == 0) dir_hdr[t.dir_base + blk] = (uint16_t)hdr;
      if (lane < 32) rows = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(pack) + 16 * lane);
      else rows = load16_unaligned(p + 1 + doc_sz + 1 + 16 * (lane & 31));
      rows = store_rows_from_file(rows, hdr, lane, has_freqs != 0);
      wave_sync();
    }
    cur = nxt;
    const int half = lane >> 5, row = lane & 31;
    const int rd = store_doc_rows(hdr);
    if (row < (half ? store_freq_rows(hdr) : rd))
      *reinterpret_cast<uint4*>(term_rows + 16 * (size_t)(row0 + (uint32_t)(half ? rd : 0) + (uint32_t)row)) = rows;
    // Every FullBlock is decoded and validated here, once: doc ids inside the segment and strictly increasing (a zero
    // delta or a 32-bit wrap shows up as d1 <= d0 or d0 <= the previous lane's d1), and the block ending on the doc its
    // skip entry names. The query kernels then gather norms / live bits and index windows with these docs unchecked.
    const BlockPair bp = block_rows_decode<LEGACY>(rows, hdr, slab, lane);
    int32_t d0, d1;
    deltas_to_docs(bp.d0, bp.d1, base, d0, d1);
    // df % 128 == 0: no skip entry names the final block's last doc (k_skip_dir left a sentinel there); the
    // decode does — windowed consumers (the OR kernel) can then tell that the term has ended
    if (blk == t.nblocks - 1 && t.nblocks > t.n_entries && lane == 63) dir_last[t.dir_base + blk] = d1;
    const int32_t prev = __builtin_amdgcn_update_dpp(base, d1, 0x138, 0xf, 0xf, false);  // wave_shr:1; lane 0 <- the block's base doc
    // (a term's very first delta is relative to doc 0 and may be 0: posting_writer.rs:298)
    const bool bad_first = (blk == 0 && lane == 0) ? d0 < 0 : d0 <= prev;
    const bool bad = bad_first || d1 <= d0 || d1 >= max_doc;
    const bool bad_last = blk < t.n_entries && lane == 63 && d1 != readlane(my_last, j);
    if (__ballot(bad || bad_last)) { if (lane == 0) flag_err(err, -4, 15); return; }
    base = readlane(d1, 63);
    if (lane == 0) dir_bmax[t.dir_base + blk] = 15ull;  // "no bound" until (unless) stage B learns the block's norms
    // (k_decode_terms' store bursts — four blocks' postings held in registers and stored together — do not pay here: 1.20 ms
    // against 1.15 on the 100 M-doc shard, 80 VGPRs / 6 wavefronts per SIMD instead of 70 / 7; the row stores sit between
    // the bursts anyway)
    if (docs_out != nullptr && t.out_base >= 0) {  // wave-uniform
      const int64_t o = t.out_base + 128 * (int64_t)blk + 2 * lane;
      __builtin_nontemporal_store(d0, docs_out + o);
      __builtin_nontemporal_store(d1, docs_out + o + 1);
      __builtin_nontemporal_store((int32_t)bp.f0, freqs_out + o);
      __builtin_nontemporal_store((int32_t)bp.f1, freqs_out + o + 1);
    }
  }
}

// ---- B: posting-order norms + the (freq, norm rank) frontier of every block, for terms that get scored ------------------------
// items as in k_prepare_blocks; `pn_base` of a term = where its posting-order norms go (PrepTerm::pn_base)
template <bool LEGACY>
__global__ __launch_bounds__(PREP_THREADS) void k_prepare_norms(SegView seg, const PrepTerm* __restrict__ terms,
                                                                const int64_t* __restrict__ item_prefix, int n_terms, int64_t n_items,
                                                                uint8_t* pnorm, uint64_t* __restrict__ dir_bmax, int ranked) {
  __shared__ __attribute__((aligned(16))) uint8_t slabs[PREP_WAVES][2 * SLAB_STREAM];
  const int lane = lane_id();
  const int wave = wave_id();
  const int64_t item = (int64_t)blockIdx.x * PREP_WAVES + wave;
  if (item >= n_items) return;
  const int ti = upper_slot_wave(item_prefix, n_terms, item, lane);
  const PrepTerm t = terms[ti];
  const int b0 = (int)(item - item_prefix[ti]) * PREP_BLOCKS_PER_ITEM;
  const int b1 = min(t.nblocks, b0 + PREP_BLOCKS_PER_ITEM);
  const uint8_t* term_rows = seg.bstore + t.bs_base;
  uint8_t* pn = pnorm + t.pn_base;
  const int tail_n = t.df > 1 ? t.df % 128 : 0;
  if (tail_n > 0 && b0 + PREP_BLOCKS_PER_ITEM >= t.nblocks) {  // the tail's norms continue the FullBlocks'
    int32_t d0, d1;
    uint32_t f0, f1;
    tail_load(term_rows, seg.dir_row[t.dir_base + t.nblocks], lane, d0, d1, f0, f1);
    const uint32_t n0 = 2 * lane < tail_n ? seg.norms[d0] : 0u, n1 = 2 * lane + 1 < tail_n ? seg.norms[d1] : 0u;
    *reinterpret_cast<uint16_t*>(pn + 128 * (uint64_t)t.nblocks + 2 * lane) = (uint16_t)(n0 | (n1 << 8));
  }
  int32_t base = b0 == 0 ? 0 : seg.dir_last[t.dir_base + b0 - 1];
  stream_blocks<LEGACY, false>(term_rows, seg.dir_row, seg.dir_hdr, t.dir_base, nullptr, b0, b1, slabs[wave], lane, base,
                               [&](int blk, int32_t d0, int32_t d1, uint32_t f0, uint32_t f1, uint32_t, uint32_t) {
    const uint32_t n0 = seg.norms[d0], n1 = seg.norms[d1];
    *reinterpret_cast<uint16_t*>(pn + 128 * (uint64_t)blk + 2 * lane) = (uint16_t)(n0 | (n1 << 8));
    // the block's (freq, norm rank) frontier word (SegView::dir_bmax)
    uint64_t w = 15ull;
    const uint32_t fmax = wave_reduce_max_u32(f0 > f1 ? f0 : f1);
    if (ranked && fmax <= 10u) {
      w = fmax;
#pragma unroll
      for (uint32_t f = 1; f <= 10; ++f) {
        const uint32_t r0 = f0 == f ? n0 : 0u, r1 = f1 == f ? n1 : 0u;
        w |= (uint64_t)(wave_reduce_max_u32(r0 > r1 ? r0 : r1) & 63u) << (4 + 6 * (f - 1));
      }
    }
    if (lane == 0) dir_bmax[t.dir_base + blk] = w;
  });
}

// ---- C: the frontier of every whole chunk of 64 blocks (SegView::dir_sum), from the words stage B has just written ------------
// items as in k_prepare_norms (PREP_BLOCKS_PER_ITEM = 32 blocks each): the even items of a term take the chunk that starts with them
static_assert(PREP_BLOCKS_PER_ITEM == 32, "k_chunk_frontiers pairs the items of k_prepare_norms into chunks of 64 blocks");
__global__ __launch_bounds__(PREP_THREADS) void k_chunk_frontiers(const PrepTerm* __restrict__ terms, const int64_t* __restrict__ item_prefix,
                                                                  int n_terms, int64_t n_items, const uint64_t* __restrict__ dir_bmax,
                                                                  uint64_t* __restrict__ dir_sum) {
  const int lane = lane_id();
  const int64_t item = (int64_t)blockIdx.x * PREP_WAVES + wave_id();
  if (item >= n_items) return;
  const int ti = upper_slot_wave(item_prefix, n_terms, item, lane);
  const PrepTerm t = terms[ti];
  const int it = (int)(item - item_prefix[ti]);
  if (it & 1) return;
  const int cj = it >> 1;
  if (64 * cj + 64 > t.nblocks) return;  // the partial chunk at the end has no word
  const uint64_t w = dir_bmax[t.dir_base + 64u * (uint32_t)cj + (uint32_t)lane];
  const uint32_t f = (uint32_t)w & 15u;
  uint64_t sum = wave_reduce_max_u32(f);  // (15 — no bound — wins, as it must)
#pragma unroll
  for (int i = 0; i < 10; ++i) sum |= (uint64_t)wave_reduce_max_u32((uint32_t)(w >> (4 + 6 * i)) & 63u) << (4 + 6 * i);
  if (((uint32_t)sum & 15u) > 10u) sum = 15ull;
  if (lane == 0) dir_sum[((t.dir_base + 63u) >> 6) + (uint32_t)cj] = sum;
}

}  // namespace rgpu
