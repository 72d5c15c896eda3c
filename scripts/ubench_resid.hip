// probe: are two 256-thread workgroups of a 176-VGPR kernel co-resident on a CU (2 waves per SIMD)?
// every workgroup records (HW_ID, XCC_ID, start, end) of its wave 0 in s_memrealtime ticks (100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v32f __attribute__((ext_vector_type(32)));
template <int NV>
__global__ void __launch_bounds__(256, 2) k(float* out, unsigned long long* rec, int iters, float s) {
  extern __shared__ char lds[];
  v32f a[NV];
  for (int j = 0; j < NV; ++j) for (int i = 0; i < 32; ++i) a[j][i] = i + j + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[j][i]) : "v"(s));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { rec[blockIdx.x * 4] = hwid; rec[blockIdx.x * 4 + 1] = xcc; rec[blockIdx.x * 4 + 2] = t0; rec[blockIdx.x * 4 + 3] = t1; }
  float r = 0;
  for (int j = 0; j < NV; ++j) for (int i = 0; i < 32; ++i) r += a[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NV> void run(const char* nm, int grid) {
  float* out; unsigned long long* rec; hipMalloc(&out, grid * 256 * 4); hipMalloc(&rec, grid * 32);
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 16384, 0, out, rec, 20000, 1.0001f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4); hipMemcpy(h.data(), rec, grid * 32, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0; for (int b = 0; b < grid; ++b) { tmin = std::min(tmin, h[b * 4 + 2]); tmax = std::max(tmax, h[b * 4 + 3]); }
  int late = 0; double dur = 0; for (int b = 0; b < grid; ++b) { if (h[b * 4 + 2] - tmin > 1000) ++late; dur += (double)(h[b * 4 + 3] - h[b * 4 + 2]); }
  printf("%s grid %d: kernel span %.1f us, mean block duration %.1f us, blocks starting > 10 us after the first: %d\n", nm, grid, (tmax - tmin) / 100.0, dur / grid / 100.0, late);
  hipFree(out); hipFree(rec);
}
int main() {
  run<1>("NV=1 (~40 VGPR)", 256); run<1>("NV=1 (~40 VGPR)", 512);
  run<5>("NV=5 (168 VGPR)", 256); run<5>("NV=5 (168 VGPR)", 512); run<5>("NV=5 (168 VGPR)", 768); run<6>("NV=6 (~200 VGPR)", 256); run<6>("NV=6 (~200 VGPR)", 512); run<6>("NV=6 (~200 VGPR)", 768);
  return 0;
}
