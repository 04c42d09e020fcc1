// LDS read-modify-write throughput on gfx950: float atomic add vs integer atomic adds vs plain read+add+write, 512-thread
// workgroups over a 48 KB window with pseudo-random or increasing addresses. hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int WS = 12288;
template <int MODE, int PATTERN>
__global__ __launch_bounds__(512) void k(int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) uint64_t mem64[WS / 2 + (MODE == 2 ? WS / 2 : 0)];
  float* accf = reinterpret_cast<float*>(mem64);
  uint32_t* accu = reinterpret_cast<uint32_t*>(mem64);
  for (int i = threadIdx.x; i < WS; i += 512) accf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = (uint32_t)(blockIdx.x * 977 + wave * 131 + 7);
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t o;
    if (PATTERN == 0) o = ((x >> 8) + (uint32_t)lane * 2654435761u) % WS;          // random per lane
    else o = (((x >> 8) % (WS - 256)) + 3u * (uint32_t)lane + ((uint32_t)lane >> 3));  // increasing with the lane (a sorted block)
    if (MODE == 0) __hip_atomic_fetch_add(accf + o, 1.0f + (float)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 1) __hip_atomic_fetch_add(accu + o, 1u + (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 2) __hip_atomic_fetch_add(mem64 + o, 1ull + (uint64_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) accf[o] = accf[o] + 1.0f + (float)lane;   // racy read+add+write
    else accf[o] = 1.0f + (float)lane;                             // plain store
  }
  __syncthreads();
  if (sink && accf[threadIdx.x] == 12345.678f) sink[0] = 1.f;
}
template <int MODE, int PATTERN>
static void run(const char* name) {
  const int iters = 2000, grid = 4096;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(grid), dim3(512), 0, 0, 10, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(grid), dim3(512), 0, 0, iters, nullptr);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveops = (double)grid * 8 * iters;
  // cycles per wave-op per CU at 2.4 GHz, 256 CUs
  printf("%-28s pattern %d: %8.3f ms  %7.1f G wave-ops/s  %6.1f cycles per wave-op per CU\n", name, PATTERN, ms, waveops / ms / 1e6,
         ms * 1e-3 * 2.4e9 * 256 / waveops);
}
int main() {
  run<0, 0>("ds_add_f32 (atomic)"); run<0, 1>("ds_add_f32 (atomic)");
  run<1, 0>("ds_add_u32 (atomic)"); run<1, 1>("ds_add_u32 (atomic)");
  run<2, 0>("ds_add_u64 (atomic)"); run<2, 1>("ds_add_u64 (atomic)");
  run<3, 0>("read + add + write"); run<3, 1>("read + add + write");
  run<4, 0>("plain store"); run<4, 1>("plain store");
  return 0;
}
