// A map keyed by file pointers (int64 >= 0) for the host side of a search call: open addressing over one flat array
// (load <= 1/2, linear probing, one cache line per hit as a rule) instead of a node-based map. The prepared-term table
// of a segment is looked up twice for every clause of every batch (is the term prepared? where are its structures?);
// through std::unordered_map the 10 240 clauses of a ten-term OR batch cost 2 ms of host time per batch — serialised
// with the GPU, because an OR group ends in a synchronisation. Host-only; tests/cpp/flat_fp_map_test.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rucene {

template <typename V>
class FlatFpMap {
 public:
  const V* find(int64_t key) const {
    if (slots_.empty() || key < 0) return nullptr;
    for (size_t i = hash(key) & mask_;; i = (i + 1) & mask_) {
      const Slot& s = slots_[i];
      if (s.key == key) return &s.value;
      if (s.key == EMPTY) return nullptr;
    }
  }
  // Ask for the cache line a later find(key) / put(key) starts at. A table of 150 k prepared terms is 25 MB: a bulk caller (a
  // first-touch decode of every term of a segment) that looks its keys up one after the other waits ~80 ns of DRAM latency per
  // key; asking 16 keys ahead makes that ~10 ns.
  void prefetch(int64_t key) const {
    if (!slots_.empty() && key >= 0) __builtin_prefetch(&slots_[hash(key) & mask_], 0, 1);
  }
  // key >= 0; an existing key keeps its slot and takes the new value
  void put(int64_t key, const V& value) {
    if (key < 0) return;
    if ((used_ + 1) * 2 > slots_.size()) grow();
    for (size_t i = hash(key) & mask_;; i = (i + 1) & mask_) {
      Slot& s = slots_[i];
      if (s.key == key) { s.value = value; return; }
      if (s.key == EMPTY) { s.key = key; s.value = value; ++used_; return; }
    }
  }
  void clear() { slots_.clear(); used_ = 0; mask_ = 0; }
  // room for n more keys without rehashing on the way
  void reserve_more(size_t n) {
    size_t want = slots_.empty() ? 1024 : slots_.size();
    while ((used_ + n + 1) * 2 > want) want *= 2;
    if (want != slots_.size()) regrow(want);
  }
  size_t size() const { return used_; }
  // drop the given keys (a rebuild of the table: the rare path of a bulk insert that has to be taken back)
  void remove_keys(const int64_t* keys, size_t n) {
    if (n == 0 || slots_.empty()) return;
    FlatFpMap<char> gone;
    gone.reserve_more(n);
    for (size_t i = 0; i < n; ++i) gone.put(keys[i], 1);
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(old.size(), Slot());
    mask_ = slots_.size() - 1;
    used_ = 0;
    for (const Slot& s : old) if (s.key != EMPTY && !gone.find(s.key)) put(s.key, s.value);
  }

 private:
  static constexpr int64_t EMPTY = INT64_MIN;  // (a file pointer is never negative)
  struct Slot { int64_t key = EMPTY; V value{}; };
  static size_t hash(int64_t k) {
    const uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull;
    return (size_t)(x ^ (x >> 29));
  }
  void grow() { regrow(slots_.empty() ? 1024 : slots_.size() * 2); }
  void regrow(size_t n_slots) {
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(n_slots, Slot());
    mask_ = slots_.size() - 1;
    used_ = 0;
    for (const Slot& s : old) if (s.key != EMPTY) put(s.key, s.value);
  }
  std::vector<Slot> slots_;
  size_t used_ = 0, mask_ = 0;
};

}  // namespace rucene
