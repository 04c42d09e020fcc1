// csrc/host/host_threads.hpp: parallel_run covers every part exactly once whatever the thread count, part_range tiles [0, n),
// a two-pass count / fill over ranges reproduces the sequential prefix sums, NoInitAlloc vectors behave like vectors.
#include "../../rucene_amd/csrc/host/host_threads.hpp"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <numeric>

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)

struct Entry { int64_t key; int32_t a, b; };

int main() {
  CHECK(rucene::host_threads() >= 1 && rucene::host_threads() <= 64);
  for (int parts : {1, 2, 3, 8, 17}) {
    std::vector<std::atomic<int>> seen((size_t)parts);
    for (auto& s : seen) s = 0;
    rucene::parallel_run(parts, [&](int t) { seen[(size_t)t]++; });
    for (auto& s : seen) CHECK(s == 1);
    for (size_t n : {size_t(0), size_t(1), size_t(7), size_t(1000), size_t(100003)}) {
      size_t at = 0;
      for (int t = 0; t < parts; ++t) {
        const auto r = rucene::part_range(n, parts, t);
        CHECK(r.first == at && r.second >= r.first);
        at = r.second;
      }
      CHECK(at == n);
    }
  }
  // count, then fill: what the bulk planner does with a term list
  const size_t n = 200001;
  std::vector<int32_t> w(n);
  for (size_t i = 0; i < n; ++i) w[i] = (int32_t)((i * 2654435761u) % 97);  // 0 = a skipped element
  std::vector<int64_t> want;  // exclusive prefix sums over the kept elements
  int64_t acc = 0;
  for (size_t i = 0; i < n; ++i) if (w[i]) { want.push_back(acc); acc += w[i]; }
  for (int parts : {1, 2, 5, 8}) {
    std::vector<size_t> cnt((size_t)parts);
    std::vector<int64_t> sum((size_t)parts);
    rucene::parallel_run(parts, [&](int t) {
      const auto r = rucene::part_range(n, parts, t);
      size_t c = 0; int64_t s = 0;
      for (size_t i = r.first; i < r.second; ++i) if (w[i]) { ++c; s += w[i]; }
      cnt[(size_t)t] = c; sum[(size_t)t] = s;
    });
    const size_t total = std::accumulate(cnt.begin(), cnt.end(), size_t(0));
    CHECK(total == want.size());
    std::vector<Entry, rucene::NoInitAlloc<Entry>> out;
    out.resize(total);
    rucene::parallel_run(parts, [&](int t) {
      const auto r = rucene::part_range(n, parts, t);
      size_t j = 0; int64_t s = 0;
      for (int u = 0; u < t; ++u) { j += cnt[(size_t)u]; s += sum[(size_t)u]; }
      for (size_t i = r.first; i < r.second; ++i) if (w[i]) { out[j++] = Entry{s, w[i], (int32_t)i}; s += w[i]; }
    });
    bool same = true;
    for (size_t j = 0; j < total; ++j) same = same && out[j].key == want[j];
    CHECK(same);
    out.push_back(Entry{1, 2, 3});
    CHECK(out.size() == total + 1 && out.back().b == 3);
    std::vector<Entry, rucene::NoInitAlloc<Entry>> moved = std::move(out);
    CHECK(moved.size() == total + 1 && out.empty());
  }
  // the same through two_pass_run: one start of the threads, mid() between the passes; and a mid() that says no
  for (int parts : {1, 2, 5, 8}) {
    std::vector<size_t> cnt((size_t)parts);
    std::vector<int64_t> sum((size_t)parts);
    std::vector<Entry, rucene::NoInitAlloc<Entry>> out;
    int mids = 0;
    rucene::two_pass_run(parts,
      [&](int t) {
        const auto r = rucene::part_range(n, parts, t);
        size_t c = 0; int64_t s = 0;
        for (size_t i = r.first; i < r.second; ++i) if (w[i]) { ++c; s += w[i]; }
        cnt[(size_t)t] = c; sum[(size_t)t] = s;
      },
      [&]() { ++mids; out.resize(std::accumulate(cnt.begin(), cnt.end(), size_t(0))); return true; },
      [&](int t) {
        const auto r = rucene::part_range(n, parts, t);
        size_t j = 0; int64_t s = 0;
        for (int u = 0; u < t; ++u) { j += cnt[(size_t)u]; s += sum[(size_t)u]; }
        for (size_t i = r.first; i < r.second; ++i) if (w[i]) { out[j++] = Entry{s, w[i], (int32_t)i}; s += w[i]; }
      });
    CHECK(mids == 1 && out.size() == want.size());
    bool same = out.size() == want.size();
    for (size_t j = 0; same && j < out.size(); ++j) same = out[j].key == want[j];
    CHECK(same);
    std::atomic<int> first{0}, second{0};
    rucene::two_pass_run(parts, [&](int) { first++; }, [&]() { return false; }, [&](int) { second++; });
    CHECK(first == parts && second == 0);
    // a mid() that throws: the workers are released and joined, the exception reaches the caller
    bool caught = false;
    std::atomic<int> ran{0};
    try {
      rucene::two_pass_run(parts, [&](int) { ran++; }, [&]() -> bool { throw std::bad_alloc(); }, [&](int) { second++; });
    } catch (const std::bad_alloc&) {
      caught = true;
    }
    CHECK(caught && ran == parts && second == 0);
  }
  if (failures == 0) std::printf("host_threads OK\n");
  return failures ? 1 : 0;
}
