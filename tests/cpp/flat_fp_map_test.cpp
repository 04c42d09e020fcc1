// FlatFpMap (rucene_amd/csrc/host/flat_fp_map.hpp) against std::unordered_map on random and adversarial key sets:
// file pointers that share their low bits, overwrites, growth from empty, negative keys, clear(). Prints "ok <n>".
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <vector>

#include "../../rucene_amd/csrc/host/flat_fp_map.hpp"

struct Info { uint32_t a; int32_t b; uint64_t c; };

int main() {
  std::mt19937_64 rng(0x527563656E65ull);
  size_t checks = 0;
  for (int round = 0; round < 6; ++round) {
    rucene::FlatFpMap<Info> m;
    std::unordered_map<int64_t, Info> ref;
    if (m.find(0) || m.find(-1) || m.size() != 0) return 1;
    const int n = round == 0 ? 3 : 50000 * round;
    for (int i = 0; i < n; ++i) {
      int64_t k;
      switch (round % 3) {
        case 0: k = (int64_t)(rng() >> 20); break;                       // spread
        case 1: k = (int64_t)(rng() % 4096) << 24; break;                // multiples of 16 MiB: equal low bits, many repeats
        default: k = (int64_t)i * 1027 + (int64_t)(rng() % 3); break;    // dense, near-sequential
      }
      const Info v{(uint32_t)i, (int32_t)(rng() % 1000), rng()};
      m.put(k, v);
      ref[k] = v;
      if (i % 1000 == 0) m.put(-5 - i, v);  // ignored
    }
    if (m.size() != ref.size()) { std::printf("size %zu != %zu\n", m.size(), ref.size()); return 2; }
    for (const auto& kv : ref) {
      const Info* f = m.find(kv.first);
      if (!f || f->a != kv.second.a || f->b != kv.second.b || f->c != kv.second.c) { std::printf("lost key %lld\n", (long long)kv.first); return 3; }
      ++checks;
    }
    for (int i = 0; i < 20000; ++i) {
      const int64_t k = (int64_t)(rng() >> 18);
      if ((m.find(k) != nullptr) != (ref.count(k) != 0)) { std::printf("phantom key %lld\n", (long long)k); return 4; }
      ++checks;
    }
    if (m.find(-1) || m.find(INT64_MIN)) return 5;
    {  // remove_keys (a bulk insert taken back): every other key goes, the rest keep their values, the size follows
      std::vector<int64_t> gone;
      size_t j = 0;
      for (const auto& kv : ref) if ((j++ & 1) == 0) gone.push_back(kv.first);
      gone.push_back(-7);                     // a key that was never there (and an illegal one at that)
      gone.push_back((int64_t)1 << 61);       // ... and a legal one that is absent
      m.remove_keys(gone.data(), gone.size());
      for (size_t g = 0; g + 2 < gone.size(); ++g) { if (m.find(gone[g])) return 7; ref.erase(gone[g]); }
      if (m.size() != ref.size()) return 8;
      for (const auto& kv : ref) {
        const Info* f = m.find(kv.first);
        if (!f || f->a != kv.second.a || f->c != kv.second.c) return 9;
        ++checks;
      }
      m.put(gone[0], Info{1u, 2, 3u});        // a removed key can come back
      if (!m.find(gone[0]) || m.size() != ref.size() + 1) return 10;
      ref[gone[0]] = Info{1u, 2, 3u};
    }
    m.clear();
    if (m.size() != 0 || m.find(ref.begin()->first)) return 6;
  }
  std::printf("ok %zu\n", checks);
  return 0;
}
