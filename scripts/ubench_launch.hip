// per-kernel cost of back-to-back launches replayed from a hipGraph (the way bench.py times a kernel): an empty kernel,
// and one whose 256 x 1024 threads each load one dword and hit one barrier (the fixed part of the fused GEMV)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void __launch_bounds__(1024) k_load_barrier(const unsigned* x, unsigned* out) {
  __shared__ unsigned s[1024];
  s[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  if (s[(threadIdx.x + 1) & 1023] == 0x12345u) out[0] = 1;
}
template <typename F> float graph_us(F launch, int n) {
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(st);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / (5 * n);
}
int main() {
  unsigned *x, *out; hipMalloc(&x, 4096); hipMemset(x, 0, 4096); hipMalloc(&out, 64);
  printf("empty kernel, grid 1 x 64:          %.2f us per launch\n", graph_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, 200));
  printf("empty kernel, grid 256 x 1024:      %.2f us per launch\n", graph_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, s); }, 200));
  printf("load + barrier, grid 256 x 1024:    %.2f us per launch\n", graph_us([&](hipStream_t s) { hipLaunchKernelGGL(k_load_barrier, dim3(256), dim3(1024), 0, s, x, out); }, 200));
  return 0;
}
