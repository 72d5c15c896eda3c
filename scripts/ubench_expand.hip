// What can a kernel with dequantize's traffic (0.5625 B in, 2 B out per element; Q4_K 11008 x 4096: 25 MB in, 90 MB out)
// reach when everything comes from / goes to HBM?  "expand": each lane reads 4 bytes (+ a shared 16-byte header per 32
// lanes) and writes 16 bytes, no arithmetic to speak of.  Buffers are cycled (16 inputs, 4 outputs: > 256 MB + 32 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(256) expand(const unsigned* __restrict__ in, v4u* __restrict__ out, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const unsigned q = in[i + (i >> 5) * 4 + 4];            // 4 payload bytes; 144-byte blocks: 16-byte header + 32 x 4
  const unsigned hd = in[(i >> 5) * 36];                  // the block header word (one L1 line per 32 lanes)
  const v4u v = {q & 0x0F0F0F0F, (q >> 4) & 0x0F0F0F0F, hd, q ^ hd};
  if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
}
__global__ void __launch_bounds__(256) fill(v4u* __restrict__ out, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = v4u{1, 2, 3, (unsigned)i};
}
int main() {
  const size_t n16 = (size_t)11008 * 4096 / 8;             // 16-byte outputs
  const size_t in_bytes = n16 / 32 * 144 + 4096, out_bytes = n16 * 16;
  std::vector<unsigned*> ins(16); std::vector<v4u*> outs(4);
  for (auto& p : ins) { hipMalloc(&p, in_bytes); hipMemset(p, 0x5A, in_bytes); }
  for (auto& p : outs) hipMalloc(&p, out_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 8; ++i) launch(i);
    hipEventRecord(e0);
    const int n = 64;
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %6.2f us  %5.2f TB/s\n", name, ms * 1000 / n, bytes / (ms / n * 1e-3) / 1e12);
  };
  const unsigned grid = (unsigned)((n16 + 255) / 256);
  run("expand (25 MB in, 90 MB out), cold", [&](int i) { hipLaunchKernelGGL(expand<0>, dim3(grid), dim3(256), 0, 0, ins[i % 16], outs[i % 4], n16); }, in_bytes + out_bytes);
  run("expand, nontemporal stores, cold", [&](int i) { hipLaunchKernelGGL(expand<1>, dim3(grid), dim3(256), 0, 0, ins[i % 16], outs[i % 4], n16); }, in_bytes + out_bytes);
  run("expand, warm (one input, one output)", [&](int i) { hipLaunchKernelGGL(expand<0>, dim3(grid), dim3(256), 0, 0, ins[0], outs[0], n16); }, in_bytes + out_bytes);
  run("fill 90 MB, cold (4 outputs)", [&](int i) { hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, outs[i % 4], n16); }, out_bytes);
  run("fill 90 MB, warm", [&](int i) { hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, outs[0], n16); }, out_bytes);
  return 0;
}
