// Host threads for the one place where the host side of a call is a bulk job: planning a first touch of a whole term dictionary
// (152 k terms of a 100 M-doc segment: every term's descriptor, directory base, row budget and three prefix arrays — ~2 ms of
// one thread in front of 1.3 ms of kernels). The planning is a counting pass and a filling pass over contiguous ranges of the
// call's terms; parallel_run gives each range a thread for the length of a pass. Host-only; tests/cpp/host_threads_test.cpp.
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <new>
#include <thread>
#include <utility>
#include <vector>

namespace rucene {

// One step of a bounded spin: a CPU pause hint while the wait is young (x86 `pause`, aarch64 `yield`), a scheduler yield once it is
// not — mid() of two_pass_run may hold an hipMalloc (0.3-20 ms): seven cores must not spin through that.
inline void spin_wait_step(int& spins) {
  if (++spins < 2048) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
  } else {
    std::this_thread::yield();
  }
}

// How many threads a bulk pass may use: the CPUs this process may run on (affinity mask), the cgroup's CPU quota when there is one
// (cgroup v2 `cpu.max`: a container with "1600000 100000" has 16 CPUs' worth of time whatever the mask says), at most 8 — the
// passes are memory-bound and short, more threads only add start-up time. RGPU_HOST_THREADS overrides (1 = everything inline).
inline int host_threads() {
  if (const char* e = std::getenv("RGPU_HOST_THREADS")) {  // (looked at on every call: tests flip it between two calls)
    const int n = std::atoi(e);
    if (n >= 1) return std::min(n, 64);
  }
  static const int v = [] {
    int n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (std::FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long period = 0;
      char word[32] = {0};
      if (std::fscanf(f, "%31s %lld", word, &period) == 2 && word[0] != 'm' && period > 0) {
        const long long quota = std::atoll(word);
        if (quota > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
      }
      std::fclose(f);
    } else {  // cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1: no quota)
      long long quota = -1, period = 0;
      if (std::FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(q, "%lld", &quota) != 1) quota = -1; std::fclose(q); }
      if (std::FILE* pf = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(pf, "%lld", &period) != 1) period = 0; std::fclose(pf); }
      if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
    }
    return std::min(n, 8);
  }();
  return v;
}

// fn(0) .. fn(n_parts - 1), each once, fn(0) on the calling thread; returns when all are done. A thread that cannot be started
// (resource limits) has its part run inline: the result never depends on how many threads there were.
template <class F>
inline void parallel_run(int n_parts, F&& fn) {
  if (n_parts <= 1) { if (n_parts == 1) fn(0); return; }
  std::vector<std::thread> threads;
  threads.reserve((size_t)n_parts - 1);
  std::vector<int> inline_parts;
  for (int t = 1; t < n_parts; ++t) {
    try {
      threads.emplace_back([&fn, t] { fn(t); });
    } catch (...) {
      inline_parts.push_back(t);
    }
  }
  fn(0);
  for (int t : inline_parts) fn(t);
  for (auto& th : threads) th.join();
}

// Two passes over the same parts with the threads kept between them: pass1(t) for every part, then mid() ONCE on the calling
// thread (all of pass 1 is visible to it), then — if mid() returned true — pass2(t) for every part (mid()'s writes are visible to
// them). The workers wait for the length of mid() — a bounded spin (2048 pause hints), then scheduler yields: mid() sizes buffers and may sit in
// an allocation for milliseconds; starting threads twice costs more than the wait, and a core that has just run pass 1 is awake for pass 2. A thread that cannot be started has both passes of
// its part run on the calling thread.
template <class F1, class Mid, class F2>
inline void two_pass_run(int n_parts, F1&& pass1, Mid&& mid, F2&& pass2) {
  if (n_parts <= 1) {
    if (n_parts == 1) { pass1(0); if (mid()) pass2(0); }
    return;
  }
  std::atomic<int> done{0}, go{0};
  std::vector<std::thread> threads;
  threads.reserve((size_t)n_parts - 1);
  std::vector<int> inline_parts;
  for (int t = 1; t < n_parts; ++t) {
    try {
      threads.emplace_back([&, t] {
        pass1(t);
        done.fetch_add(1, std::memory_order_release);
        int g, spins = 0;
        while ((g = go.load(std::memory_order_acquire)) == 0) spin_wait_step(spins);
        if (g > 0) pass2(t);
      });
    } catch (...) {
      inline_parts.push_back(t);
    }
  }
  // (whatever the calling thread's share throws — mid() sizes buffers — the workers are released and joined first)
  std::exception_ptr thrown;
  bool ok = false;
  try {
    pass1(0);
    for (int t : inline_parts) pass1(t);
  } catch (...) {
    thrown = std::current_exception();
  }
  { int spins = 0; while (done.load(std::memory_order_acquire) != (int)threads.size()) spin_wait_step(spins); }
  if (!thrown) {
    try { ok = mid(); } catch (...) { thrown = std::current_exception(); }
  }
  go.store(ok ? 1 : -1, std::memory_order_release);
  if (ok) {
    try {
      pass2(0);
      for (int t : inline_parts) pass2(t);
    } catch (...) {
      thrown = std::current_exception();
    }
  }
  for (auto& th : threads) th.join();
  if (thrown) std::rethrow_exception(thrown);
}

// [begin, end) of part t when n items are dealt to n_parts contiguous ranges
inline std::pair<size_t, size_t> part_range(size_t n, int n_parts, int t) {
  return {n * (size_t)t / (size_t)n_parts, n * ((size_t)t + 1) / (size_t)n_parts};
}

// std::vector<T, NoInitAlloc<T>>::resize(n) leaves trivially constructible elements uninitialised: an array that a filling pass
// writes in full right away is not written twice (and its pages are first touched by the threads that fill them).
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U>
  void construct(U* p) { ::new ((void*)p) U; }
  template <class U, class... A>
  void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};

}  // namespace rucene
