// L2-resident read bandwidth: every wave streams the same small buffer (all CUs hit the same lines),
// 16 B per lane per load, UNROLL loads in flight.  Also variant with per-wave rotated start.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int UNROLL, bool ROT>
__global__ void __launch_bounds__(256) k(const v4i* __restrict__ buf, int n_chunks /*1 KB chunks*/, int iters, int* out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  v4i acc = {0, 0, 0, 0};
  int c = ROT ? (wave * 7) % n_chunks : 0;
  for (int it = 0; it < iters; ++it) {
    v4i v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u] = buf[(size_t)c * 64 + lane];
      c = c + 1 == n_chunks ? 0 : c + 1;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if (acc[0] == 0x12345678) out[0] = acc[1] + acc[2] + acc[3];
}
template <int UNROLL, bool ROT>
void run(const v4i* buf, int n_chunks, int grid, int* out, const char* name) {
  const int iters = 256 / UNROLL * 4;  // 1024 loads of 1 KB per wave = 1 MB per wave
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<UNROLL, ROT><<<grid, 256>>>(buf, n_chunks, iters, out);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<UNROLL, ROT><<<grid, 256>>>(buf, n_chunks, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)grid * 4 * iters * UNROLL * 1024;
  printf("%-28s buf %5d KB grid %5d: %.1f us  %.2f TB/s (%.1f B/clk/CU @2.4GHz)\n", name, n_chunks, grid, ms * 1e3,
         bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3);
}
int main() {
  v4i* buf; int* out;
  hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&out, 4);
  for (int kb : {16, 590, 4096, 32768}) {
    run<4, false>(buf, kb, 768, out, "unroll4 lockstep");
    run<4, true>(buf, kb, 768, out, "unroll4 rotated");
    run<16, false>(buf, kb, 768, out, "unroll16 lockstep");
    run<16, true>(buf, kb, 768, out, "unroll16 rotated");
    run<16, true>(buf, kb, 1024, out, "unroll16 rotated 4w/SIMD");
  }
  return 0;
}
