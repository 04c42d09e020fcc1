// The table of a segment's prepared terms: doc_start_fp -> TermInfo (where the term's directory, block-store rows and posting-order
// norms are). Host-only; tests/cpp/prepared_map_test.cpp drives it against a std::map.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "flat_fp_map.hpp"
#include "host_threads.hpp"

namespace rucene {

struct TermInfo {
  uint32_t dir_base; int32_t nblocks; int32_t df; uint64_t pn_base; uint64_t bs_base; bool norms;
  uint32_t sketch = 0;  // 1 + index of the term's block-max sketch (kernels/search_term.hpp), 0: none
};

// doc_start_fp -> TermInfo of the prepared terms (host/flat_fp_map.hpp: two look-ups per clause per batch).
// A bulk first touch (every term of a segment's dictionary: 152 k at 100 M docs) used to end with 152 k insertions into a table
// that is grown and first touched right there — 4 to 8 ms of page faults and cache misses for 1.3 ms of kernels. Such a call's
// terms arrive in file order: they are kept as the sorted array the planning loop built anyway (`bulk`), looked up by binary
// search, and a term moves into the hash table the first time a query names it.
// One term of a bulk call, 16 bytes: what differs from term to term. The rest of its TermInfo is the same for every term of the
// call (block-store base of the call's region, "has no norms to prepare") or follows (nblocks = df / 128, pn_base = 0 until the
// norms are prepared — by then the term has moved into the table). 152 k of them are 2.4 MB, first touched by the planner's threads.
struct PreparedEntry { int64_t first; uint32_t dir_base; int32_t df; };
using PreparedBulk = std::vector<PreparedEntry, NoInitAlloc<PreparedEntry>>;
struct PreparedMap {
  FlatFpMap<TermInfo> map;
  PreparedBulk bulk;  // ascending keys
  uint64_t bulk_bs_base = 0;
  bool bulk_no_norms = false;
  std::vector<uint8_t> moved;                      // bulk[i] lives in `map` now
  size_t bulk_live = 0;
  // bumped by everything that adds, replaces or removes a term's TermInfo (a term moving from `bulk` into `map` keeps its value and
  // does not count): what a memo of finished descriptors checks (rgpu_api.hip term_batch_fast)
  uint64_t epoch = 1;
  static TermInfo expand(const PreparedEntry& e, uint64_t bs_base, bool no_norms) {
    return TermInfo{e.dir_base, e.df / 128, e.df, 0, bs_base, no_norms};
  }
  long bulk_at(int64_t key) const {
    size_t lo = 0, hi = bulk.size();
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (bulk[mid].first < key) lo = mid + 1; else hi = mid; }
    return (lo < bulk.size() && bulk[lo].first == key && !moved[lo]) ? (long)lo : -1;
  }
  const TermInfo* find(int64_t key) {
    if (const TermInfo* p = map.find(key)) return p;
    if (bulk_live == 0) return nullptr;
    const long i = bulk_at(key);
    if (i < 0) return nullptr;
    map.put(key, expand(bulk[(size_t)i], bulk_bs_base, bulk_no_norms));
    moved[(size_t)i] = 1;
    if (--bulk_live == 0) { bulk.clear(); moved.clear(); }  // (the array keeps its memory: the next bulk call of the segment fills it again)
    return map.find(key);
  }
  void put(int64_t key, const TermInfo& v) {
    ++epoch;
    if (bulk_live != 0) { const long i = bulk_at(key); if (i >= 0) { moved[(size_t)i] = 1; --bulk_live; } }
    map.put(key, v);
  }
  void prefetch(int64_t key) const { map.prefetch(key); }
  void reserve_more(size_t n) { map.reserve_more(n); }
  size_t size() const { return map.size() + bulk_live; }
  void clear() { ++epoch; map.clear(); bulk.clear(); moved.clear(); bulk_live = 0; }
  // An empty array for a bulk call to fill (with whatever memory the last one left), then adopt_sorted(): ascending keys none of
  // which is in the table. An earlier bulk that is still (partly) pending moves into the table first.
  PreparedBulk take_array() {
    ++epoch;
    if (bulk_live != 0) {
      map.reserve_more(bulk_live);
      for (size_t i = 0; i < bulk.size(); ++i) if (!moved[i]) map.put(bulk[i].first, expand(bulk[i], bulk_bs_base, bulk_no_norms));
      bulk_live = 0;
    }
    PreparedBulk out = std::move(bulk);
    bulk = PreparedBulk();
    out.clear();
    moved.clear();
    return out;
  }
  void adopt_sorted(PreparedBulk&& sorted, uint64_t bs_base, bool no_norms) {
    ++epoch;
    if (bulk_live != 0) (void)take_array();
    bulk = std::move(sorted);
    bulk_bs_base = bs_base;
    bulk_no_norms = no_norms;
    moved.assign(bulk.size(), 0);
    bulk_live = bulk.size();
  }
  void drop_bulk() { ++epoch; bulk.clear(); moved.clear(); bulk_live = 0; }
  void remove_keys(const int64_t* keys, size_t n) { ++epoch; map.remove_keys(keys, n); }
};

}  // namespace rucene
