// csrc/host/batch_planner.hpp for_each_flat_memo (term id -> the caller's finished record, rgpu_api.hip term_batch_fast) against
// for_each_flat: the same (state, idf) reach `make` as for_each_flat hands out, once per id per key; ids that share a slot evict each
// other and stay right; absent ids and ids outside the table arrive as null; a refusal keeps nothing and stops at once; another key
// empties the table. And csrc/host/prepared_map.hpp: `epoch` moves with every change of a term's TermInfo, not with look-ups.
#include "../../rucene_amd/csrc/host/batch_planner.hpp"
#include "../../rucene_amd/csrc/host/prepared_map.hpp"

#include <cstdio>
#include <map>
#include <random>

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)

struct Rec { int64_t fp; int32_t df; float idf; uint64_t stamp; };

int main() {
  std::mt19937_64 rng(11);
  const int64_t n_leaf = 200000;
  std::vector<rgpu_term_state> states((size_t)n_leaf);
  for (int64_t i = 0; i < n_leaf; ++i) {
    rgpu_term_state s{};
    s.doc_freq = (i % 17 == 3) ? 0 : (int32_t)(1 + rng() % 500000);  // every 17th term is absent from the leaf
    s.doc_start_fp = 1000 + 13 * i;
    s.total_term_freq = s.doc_freq * 2;
    s.skip_offset = -1;
    s.singleton_doc_id = s.doc_freq == 1 ? (int32_t)(rng() % 1000) : -1;
    states[(size_t)i] = s;
  }
  rgpu_plan_stats ps{};
  ps.max_doc = 10000000;
  ps.doc_count = 10000000;
  ps.sum_total_term_freq = 1000000000;
  ps.k1 = 1.2f;
  ps.b = 0.75f;
  rucene::BatchPlanner P(ps, 0, states.data(), n_leaf, nullptr, 0);

  // ids: a hot set that is asked again and again, pairs that share a memo slot, absent terms, ids outside the table
  std::vector<int64_t> by_slot_first(65536, -1);
  std::vector<int64_t> ids;
  for (int64_t id = 0; id < n_leaf && ids.size() < 400; ++id) {
    const size_t slot = (size_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> 48);
    if (by_slot_first[slot] < 0) by_slot_first[slot] = id;
    else { ids.push_back(by_slot_first[slot]); ids.push_back(id); ids.push_back(by_slot_first[slot]); }
  }
  for (int i = 0; i < 3000; ++i) ids.push_back((int64_t)(rng() % 5000));
  ids.push_back(-1); ids.push_back(n_leaf); ids.push_back(n_leaf + 5); ids.push_back(3); ids.push_back(20);  // 3, 20: absent terms

  // what for_each_flat says about every id
  std::vector<Rec> want(ids.size());
  P.for_each_flat(ids.data(), (int64_t)ids.size(), [&](int64_t i, const rgpu_term_state& s, float idf) { want[(size_t)i] = Rec{s.doc_start_fp, s.doc_freq, idf, 0}; });

  uint64_t stamp = 1, min_stamp = 0;   // every record handed out was made under the current key: its stamp is at least min_stamp
  std::map<int64_t, int> made;  // fp -> how often `make` ran for it under the current key
  auto run = [&](const rucene::BatchPlanner::MemoKey& key, int64_t refuse_fp) {
    size_t seen = 0;
    const bool ok = P.for_each_flat_memo<Rec>(ids.data(), (int64_t)ids.size(), key, [&](const rgpu_term_state& s, float idf, Rec* out) -> int32_t {
      if (s.doc_start_fp == refuse_fp) return -1;
      ++made[s.doc_start_fp];
      *out = Rec{s.doc_start_fp, s.doc_freq, idf, stamp};
      return 1;
    }, [&](int64_t i, const Rec* r) {
      ++seen;
      const Rec& w = want[(size_t)i];
      if (w.df <= 0) { CHECK(r == nullptr); return; }
      CHECK(r != nullptr);
      if (r) CHECK(r->fp == w.fp && r->df == w.df && r->idf == w.idf && r->stamp <= stamp && r->stamp >= min_stamp);
    });
    if (ok) CHECK(seen == ids.size());
    return ok;
  };
  const rucene::BatchPlanner::MemoKey k1{{7, 1, 0, 0}}, k2{{7, 2, 0, 0}};
  CHECK(run(k1, -1));
  size_t distinct = made.size(), calls = 0;
  for (const auto& kv : made) calls += (size_t)kv.second;
  CHECK(distinct > 2000 && calls >= distinct);       // every held term was made at least once ...
  const size_t calls_first = calls;
  ++stamp;
  CHECK(run(k1, -1));                                 // ... and on the second pass only the evicted ones again
  calls = 0;
  for (const auto& kv : made) calls += (size_t)kv.second;
  CHECK(calls - calls_first < calls_first / 2 && calls > calls_first);   // (the slot-sharing pairs: made again; the hot set: not)
  // another key: everything is made again, with the new stamp
  made.clear();
  ++stamp;
  const uint64_t stamp2 = stamp;
  min_stamp = stamp;
  CHECK(run(k2, -1));
  CHECK(made.size() == distinct);
  P.for_each_flat_memo<Rec>(ids.data(), 1, k2, [&](const rgpu_term_state&, float, Rec*) -> int32_t { CHECK(false); return 1; },
                            [&](int64_t, const Rec* r) { CHECK(r && r->stamp == stamp2); });
  // keys that change in front of every call (other batches keep preparing terms): a record of an earlier key never comes back
  for (uint64_t i = 0; i < 60; ++i) {
    ++stamp;
    min_stamp = stamp;
    CHECK(run(rucene::BatchPlanner::MemoKey{{9, i & 1, 0, 0}}, -1));
  }
  // a refusal: false at once, and the refused term is not kept (the next pass makes it)
  const int64_t refuse = want[ids.size() - 6].fp;   // (the last hot-set id: held)
  CHECK(want[ids.size() - 6].df > 0);
  made.clear();
  const rucene::BatchPlanner::MemoKey k3{{8, 1, 0, 0}};
  ++stamp;
  min_stamp = stamp;
  CHECK(!run(k3, refuse));
  CHECK(made.count(refuse) == 0);
  CHECK(run(k3, -1));
  CHECK(made.count(refuse) == 1);

  // PreparedMap::epoch
  rucene::PreparedMap pm;
  uint64_t e = pm.epoch;
  pm.put(5, rucene::TermInfo{1, 2, 300, 0, 0, false});
  CHECK(pm.epoch != e); e = pm.epoch;
  CHECK(pm.find(5) != nullptr && pm.find(6) == nullptr && pm.epoch == e);
  rucene::PreparedBulk arr = pm.take_array();
  CHECK(pm.epoch != e); e = pm.epoch;
  arr.push_back(rucene::PreparedEntry{10, 4, 256});
  arr.push_back(rucene::PreparedEntry{20, 9, 512});
  pm.adopt_sorted(std::move(arr), 64, true);
  CHECK(pm.epoch != e); e = pm.epoch;
  CHECK(pm.find(10) != nullptr && pm.epoch == e);     // a term moving from the array into the table keeps its value
  const int64_t gone[1] = {5};
  pm.remove_keys(gone, 1);
  CHECK(pm.epoch != e); e = pm.epoch;
  pm.drop_bulk();
  CHECK(pm.epoch != e); e = pm.epoch;
  pm.clear();
  CHECK(pm.epoch != e);
  if (failures) return 1;
  std::printf("planner_memo OK\n");
  return 0;
}
