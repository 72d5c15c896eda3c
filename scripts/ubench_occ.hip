// micro-benchmark: VALU issue rate versus waves per SIMD (1..8), and the MMQ apply step
// (int8 MFMA + 2 FMAs per accumulator register, operands in registers) versus waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float as_f32(int v) { return __builtin_bit_cast(float, v); }

// MODE 0: 16 independent v_fma_f32; MODE 1: 8 v_pk_fma_f32; MODE 2: 16 v_fmac with an SGPR operand
template <int MODE>
__global__ void __launch_bounds__(256, 8) kv(float* out, int iters, float s) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f ps = {s, s};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        v2f p = {a[i], a[i + 1]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(ps));
        a[i] = p[0]; a[i + 1] = p[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// one 32x32x32 int8 MFMA (magic accumulator) + 32 FMAs (apply) per step; TB token blocks share nothing.
template <int NACC>
__global__ void __launch_bounds__(256, (NACC == 1 ? 6 : NACC == 2 ? 4 : 3)) km(float* out, int iters, float s) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
  v16f acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  v16i magic; for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
  asm volatile("" : "+v"(magic));
  float sa[16]; for (int i = 0; i < 16; ++i) sa[i] = s + i;
  float bs = s * 0.5f, nmbs = -12582912.0f * bs;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      v16i c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, magic, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = __builtin_fmaf(__builtin_fmaf(as_f32(c0[i]), bs, nmbs), sa[i], acc[j][i]);
      a[0] += 1;
    }
  }
  float r = 0;
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K> float timeit(K kern, int grid, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 50, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000;
  for (int w = 1; w <= 8; ++w) {
    const float m0 = timeit(kv<0>, 256 * w, out, iters), m1 = timeit(kv<1>, 256 * w, out, iters), m2 = timeit(kv<2>, 256 * w, out, iters);
    printf("waves/SIMD %d: fma %.3f ns/wave-instr/SIMD | pk_fma %.3f ns per pk (=%.3f per fma-equiv) | fmac sgpr %.3f\n", w,
           m0 * 1e6 / ((double)iters * 16 * w), m1 * 1e6 / ((double)iters * 8 * w), m1 * 1e6 / ((double)iters * 16 * w),
           m2 * 1e6 / ((double)iters * 16 * w));
  }
  const int it2 = 4000;
  for (int w = 1; w <= 6; ++w) printf("mfma+apply NACC=1 waves/SIMD %d: %.1f ns per tile-group per SIMD\n", w, timeit(km<1>, 256 * w, out, it2) * 1e6 / ((double)it2 * w));
  for (int w = 1; w <= 4; ++w) printf("mfma+apply NACC=2 waves/SIMD %d: %.1f ns per tile-group per SIMD\n", w, timeit(km<2>, 256 * w, out, it2) * 1e6 / ((double)it2 * 2 * w));
  for (int w = 1; w <= 3; ++w) printf("mfma+apply NACC=4 waves/SIMD %d: %.1f ns per tile-group per SIMD\n", w, timeit(km<4>, 256 * w, out, it2) * 1e6 / ((double)it2 * 4 * w));
  return 0;
}
