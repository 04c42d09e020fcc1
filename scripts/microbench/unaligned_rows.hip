// Microbenchmark: cost of one 16-byte-per-lane row load per "block" when the rows are byte-misaligned
// (Lucene50 FullBlock payloads start right after a 1-byte header) vs 16-byte aligned, same stride / footprint.
// Build: hipcc --offload-arch=gfx950 -O3 -o unaligned_rows unaligned_rows.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
template <int MODE>  // 0: misaligned by 1 byte, 1: aligned 16, 2: misaligned by 4 bytes (dword aligned)
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, size_t bytes_per_wave, int nblocks, int stride, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint8_t* p = buf + wave * bytes_per_wave + (MODE == 0 ? 1 : MODE == 2 ? 4 : 0);
  const int half = lane >> 5, row = lane & 31;
  uint32_t acc = 0;
  uint4 ring[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { U4 v = *reinterpret_cast<const U4*>(p + (size_t)j * stride + half * 80 + 16 * row); ring[j] = make_uint4(v.x, v.y, v.z, v.w); }
  for (int i = 0; i + 4 <= nblocks; i += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 r = ring[j];
      const int nx = min(i + j + 4, nblocks - 1);
      U4 v = *reinterpret_cast<const U4*>(p + (size_t)nx * stride + half * 80 + 16 * row);
      ring[j] = make_uint4(v.x, v.y, v.z, v.w);
      acc += r.x ^ r.y ^ r.z ^ r.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const int waves = 256 * 32 * 4, nblocks = 256, stride = 160;
  const size_t bpw = (size_t)nblocks * stride + 2048;
  uint8_t* buf; uint32_t* out;
  hipMalloc(&buf, bpw * waves + 4096); hipMalloc(&out, 4);
  hipMemset(buf, 1, bpw * waves + 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, buf, bpw, nblocks, stride, out);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, buf, bpw, nblocks, stride, out);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, buf, bpw, nblocks, stride, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep == 2) printf("mode %d (%s): %.3f ms, %.1f ns/block/CU, %.2f TB/s footprint\n", mode, mode == 0 ? "byte-misaligned" : mode == 1 ? "16B aligned" : "dword-misaligned",
                           ms, ms * 1e6 / ((double)waves * nblocks / 256), (double)waves * nblocks * stride / ms / 1e9);
    }
  }
  return 0;
}
