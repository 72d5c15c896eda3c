// ds_read_b128 / ds_write_b128 throughput: 16-byte aligned vs 2-byte / 4-byte / 8-byte aligned addresses
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((packed, aligned(2))) u32x4_a2 { unsigned v[4]; };
template <bool WRITE>
__global__ void __launch_bounds__(256) k(int iters, int mis, int pitch, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned*)lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* base = lds + wave * 16384 + lane * pitch + mis;
  u32x4_a2 acc = {{0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (WRITE) {
        *(u32x4_a2*)(base + (u & 3) * 32) = acc;
        asm volatile("" ::: "memory");
        acc.v[0] += it;
      } else {
        u32x4_a2 v = *(u32x4_a2*)(base + (u & 3) * 32);
        asm volatile("" : "+v"(v.v[0]), "+v"(v.v[1]), "+v"(v.v[2]), "+v"(v.v[3]));
        acc.v[0] ^= v.v[0]; acc.v[1] ^= v.v[1]; acc.v[2] ^= v.v[2]; acc.v[3] ^= v.v[3];
      }
    }
  }
  if (acc.v[0] == 0x12345) out[0] = acc.v[1] + acc.v[2] + acc.v[3];
}
template <bool WRITE> void run(const char* name, int mis, int pitch, unsigned* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<WRITE><<<256, 256>>>(iters, mis, pitch, out);
  hipEventRecord(e0);
  k<WRITE><<<256, 256>>>(iters, mis, pitch, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-6s misalign %2d pitch %3d: %.1f ns per wave-instr per CU (%.1f cycles @2.1GHz)\n", name, mis, pitch,
         ms * 1e6 / (iters * 8.0 * 4), ms * 1e6 / (iters * 8.0 * 4) * 2.1);
}
int main() {
  unsigned* out; hipMalloc(&out, 4);
  for (int pitch : {16, 144, 136}) for (int mis : {0, 2, 4, 8}) { run<false>("read", mis, pitch, out); run<true>("write", mis, pitch, out); }
  return 0;
}
