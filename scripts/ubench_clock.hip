// measures the shader clock under a VALU load: s_memtime (shader cycles) vs s_memrealtime (100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fmaf(a0, 1.0001f, 1.0f); a1 = __builtin_fmaf(a1, 1.0001f, 1.0f);
    a2 = __builtin_fmaf(a2, 1.0001f, 1.0f); a3 = __builtin_fmaf(a3, 1.0001f, 1.0f);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; rt[blockIdx.x] = r1 - r0; }
}
int main() {
  const int nb = 2048;
  float* out; unsigned long long *cyc, *rt;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8); hipMalloc(&rt, nb * 8);
  for (int iters : {2000, 20000, 200000}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, out, cyc, rt, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(nb), r(nb);
    hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rt, nb * 8, hipMemcpyDeviceToHost);
    std::vector<double> f(nb);
    for (int i = 0; i < nb; ++i) f[i] = (double)c[i] / (double)r[i] * 100.0;  // MHz
    std::sort(f.begin(), f.end());
    printf("iters %d: shader clock median %.0f MHz (min %.0f, max %.0f); block time median %.1f us; cycles per fma-iter(4 fma) %.2f\n",
           iters, f[nb / 2], f[0], f[nb - 1], (double)r[nb / 2] / 100.0, (double)c[nb / 2] / iters);
  }
  return 0;
}
