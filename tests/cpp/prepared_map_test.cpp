// csrc/host/prepared_map.hpp against a std::map under random sequences of what rgpu_api.hip does with it: single puts, bulk calls
// (take_array -> fill -> adopt_sorted, or drop_bulk when the call fails), look-ups that move a term from the sorted array into the
// table, puts of keys that are still in the array, remove_keys, clear.
#include "../../rucene_amd/csrc/host/prepared_map.hpp"

#include <cstdio>
#include <map>
#include <random>

using rucene::PreparedBulk;
using rucene::PreparedEntry;
using rucene::PreparedMap;
using rucene::TermInfo;

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)

static bool same(const TermInfo& a, const TermInfo& b) {
  return a.dir_base == b.dir_base && a.nblocks == b.nblocks && a.df == b.df && a.pn_base == b.pn_base && a.bs_base == b.bs_base && a.norms == b.norms;
}

int main() {
  std::mt19937_64 rng(7);
  for (int round = 0; round < 40; ++round) {
    PreparedMap pm;
    std::map<int64_t, TermInfo> ref;
    int64_t next_key = 1 + (int64_t)(rng() % 1000);
    auto fresh_key = [&]() { next_key += 1 + (int64_t)(rng() % 97); return next_key; };
    for (int step = 0; step < 400; ++step) {
      const int op = (int)(rng() % 100);
      if (op < 20) {  // a single term prepared by a query batch
        const int64_t key = fresh_key();
        const TermInfo v{(uint32_t)(rng() % 100000), (int32_t)(rng() % 50), (int32_t)(2 + rng() % 9000), rng() % 1000, 16 * (rng() % 1000), (rng() & 1) != 0};
        pm.put(key, v);
        ref[key] = v;
      } else if (op < 32) {  // a bulk first touch: ascending keys, one block-store base and norms flag for the call
        PreparedBulk arr = pm.bulk_live == 0 ? pm.take_array() : PreparedBulk();
        CHECK(arr.empty());
        const size_t n = 1 + (size_t)(rng() % 300);
        const uint64_t bs_base = 16 * (rng() % 100000);
        const bool no_norms = (rng() & 1) != 0;
        arr.resize(n);
        std::vector<std::pair<int64_t, TermInfo>> filed;
        for (size_t i = 0; i < n; ++i) {
          const int64_t key = fresh_key();
          const int32_t df = (int32_t)(2 + rng() % 70000);
          arr[i] = PreparedEntry{key, (uint32_t)(rng() % 1000000), df};
          filed.push_back({key, TermInfo{arr[i].dir_base, df / 128, df, 0, bs_base, no_norms}});
        }
        pm.adopt_sorted(std::move(arr), bs_base, no_norms);
        if (rng() % 5 == 0) {
          pm.drop_bulk();  // the call failed: nothing of it stays
        } else {
          for (auto& f : filed) ref[f.first] = f.second;
        }
      } else if (op < 80) {  // look-ups: known keys (those of the array move into the table), unknown keys
        for (int j = 0; j < 8; ++j) {
          int64_t key;
          if (!ref.empty() && rng() % 4 != 0) {
            auto it = ref.lower_bound((int64_t)(rng() % (uint64_t)(next_key + 1)));
            if (it == ref.end()) it = ref.begin();
            key = it->first;
          } else {
            key = (int64_t)(rng() % (uint64_t)(next_key + 50));
          }
          const TermInfo* got = pm.find(key);
          auto it = ref.find(key);
          CHECK((got != nullptr) == (it != ref.end()));
          if (got && it != ref.end()) CHECK(same(*got, it->second));
        }
      } else if (op < 88) {  // a put over a key that may still sit in the array (norms prepared: pn_base assigned)
        if (!ref.empty()) {
          auto it = ref.lower_bound((int64_t)(rng() % (uint64_t)(next_key + 1)));
          if (it == ref.end()) it = ref.begin();
          TermInfo v = it->second;
          v.pn_base = 1 + rng() % 100000;
          v.norms = true;
          pm.put(it->first, v);
          it->second = v;
        }
      } else if (op < 94) {  // a failed call of single puts takes its keys back — they are in the table, never in the array
        std::vector<int64_t> keys;
        for (int j = 0; j < 5; ++j) {
          const int64_t key = fresh_key();
          const TermInfo v{1, 2, 300, 0, 0, false};
          pm.put(key, v);
          keys.push_back(key);
        }
        pm.remove_keys(keys.data(), keys.size());
      } else if (op < 96) {
        pm.clear();
        ref.clear();
      }
      CHECK(pm.size() == ref.size());
    }
    for (auto& kv : ref) {
      const TermInfo* got = pm.find(kv.first);
      CHECK(got != nullptr && same(*got, kv.second));
    }
    CHECK(pm.size() == ref.size());
  }
  if (failures == 0) std::printf("prepared_map OK\n");
  return failures ? 1 : 0;
}
