// Microbenchmark: what FETCH_SIZE (rocprofv3 --pmc) tallies per request for the access shapes this library uses —
//   k_stream   16 B per lane, coalesced (the block-store rows of the decode kernels)
//   k_gather4  one random aligned dword per lane out of a 2 GiB array (k_search_and's membership probe, k_prepare_norms)
//   k_gather1  one random byte per lane (norm gathers)
//   k_gather4_dense  dwords at a stride of 64 B (every request its own 64-byte sector, neighbouring sectors)
// Every kernel moves a KNOWN number of distinct 64-byte sectors from HBM (the array is far larger than the 256 MiB Infinity
// Cache and every index is used once), so FETCH_SIZE / that number = what one sector costs in the counter:
// MI355X_MICROARCH.md gives "x 2" for the streaming shape and calls the other widths uncalibrated.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_granule fetch_granule.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_granule     (prints the byte counts to compare with)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ a, size_t n16, uint32_t* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// index i -> a distinct pseudo-random 64-byte sector (a multiplicative permutation of the sector numbers: n_sectors is a power of two)
__device__ __forceinline__ size_t sector_of(size_t i, size_t n_sectors) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (n_sectors - 1); }
__global__ __launch_bounds__(256) void k_gather4(const uint32_t* __restrict__ a, size_t n_sectors, size_t n_probes, uint32_t* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_probes; i += (size_t)gridDim.x * 256) acc += a[sector_of(i, n_sectors) * 16 + (i & 15)];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather1(const uint8_t* __restrict__ a, size_t n_sectors, size_t n_probes, uint32_t* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_probes; i += (size_t)gridDim.x * 256) acc += a[sector_of(i, n_sectors) * 64 + (i & 63)];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather4_dense(const uint32_t* __restrict__ a, size_t n_probes, uint32_t* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_probes; i += (size_t)gridDim.x * 256) acc += a[i * 16];   // lane l reads sector base + l: 64 sectors per instruction, neighbours
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30, n_sectors = bytes / 64;
  uint8_t* buf; uint32_t* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  const size_t n_probes = n_sectors / 4;   // 8 M probes, every one its own sector: 512 MiB of sectors
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](const char* name, auto launch, double sectors) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemsetAsync(out, 0, 4, 0);   // (a different kernel in between)
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) std::printf("%-16s %8.3f ms   distinct 64-byte sectors %.0f = %.1f MiB   (%.2f TB/s of sectors)\n", name, ms, sectors, sectors * 64 / 1048576.0, sectors * 64 / (ms * 1e-3) / 1e12);
    }
  };
  timed("k_stream", [&] { k_stream<<<256 * 16, 256>>>((const uint4*)buf, bytes / 16, out); }, (double)n_sectors);
  timed("k_gather4", [&] { k_gather4<<<256 * 16, 256>>>((const uint32_t*)buf, n_sectors, n_probes, out); }, (double)n_probes);
  timed("k_gather1", [&] { k_gather1<<<256 * 16, 256>>>((const uint8_t*)buf, n_sectors, n_probes, out); }, (double)n_probes);
  timed("k_gather4_dense", [&] { k_gather4_dense<<<256 * 16, 256>>>((const uint32_t*)buf, n_probes, out); }, (double)n_probes);
  std::printf("compare: FETCH_SIZE (KiB) per dispatch x 1024 / (sectors x 64) = what the counter tallies per 64-byte sector moved\n");
  return 0;
}
