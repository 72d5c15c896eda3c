// global_load_dwordx4 throughput from an L2-resident buffer versus the byte alignment of the per-lane address
// (the Q8_0 / Q5_x blocks of a GGUF row are 2-byte aligned: 34 / 22 / 24-byte blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) u32x4_a2 { unsigned v[4]; };

// lane l of wave w reads 16 bytes at base + (l * pitch + mis) + it * step: pitch 34 = one Q8_0 block per lane
__global__ void __launch_bounds__(256) k(const unsigned char* buf, int iters, int mis, int pitch, int step, unsigned mask, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* p = buf + (size_t)(blockIdx.x * 4 + wave) * 65536 + lane * pitch + mis;
  unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  unsigned off = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32x4_a2 v = *(const u32x4_a2*)(p + ((off + u * step) & mask));
      a0 ^= v.v[0]; a1 ^= v.v[1]; a2 ^= v.v[2]; a3 ^= v.v[3];
    }
    off += 4 * step;
  }
  if (a0 == 0x12345) out[0] = a1 + a2 + a3;
}
int main() {
  unsigned char* buf; unsigned* out;
  const size_t bytes = (size_t)256 * 4 * 65536 + 65536;   // 64 MiB: every wave streams its own 64 KiB window (L2 / MALL resident)
  hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 64);
  struct { const char* name; int mis, pitch, step; } cases[] = {
    {"16-B aligned, contiguous (pitch 16)", 0, 16, 1024}, {"8-B aligned, contiguous", 8, 16, 1024}, {"4-B aligned, contiguous", 4, 16, 1024},
    {"2-B aligned, contiguous", 2, 16, 1024}, {"pitch 34 (+2): one Q8_0 block per lane", 2, 34, 2176}, {"pitch 36, 4-B aligned", 0, 36, 2304},
    {"pitch 32, 16-B aligned (stride-2 chunks)", 0, 32, 2048}, {"pitch 136 (+2), 2-B: 16 B of every 4th block", 2, 136, 8704}, {"pitch 144, 16-B aligned", 0, 144, 9216}};
  for (auto& c : cases) {
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, buf, 8, c.mis, c.pitch, c.step, 0x7FFFu, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, buf, iters, c.mis, c.pitch, c.step, 0x7FFFu, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)iters * 4 * 4;   // wave-level load instructions per CU (one workgroup of 4 waves per CU)
    printf("%-48s %7.1f ns per wave-load per CU  (%.0f GB/s per CU useful)\n", c.name, ms * 1e6 / loads_per_cu, 1024.0 / (ms * 1e6 / loads_per_cu));
  }
  return 0;
}
