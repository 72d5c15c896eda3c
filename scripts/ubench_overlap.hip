// micro-benchmark: do matrix-pipe (MFMA) and vector-ALU instructions of one SIMD overlap?
// Each wave loops over { one MFMA on its own accumulators ; NV independent VALU ops }.  If the two pipes overlap the time
// per iteration is max(t_mfma, NV * t_valu); if the SIMD serialises them it is the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MF: 0 none, 1 i32_32x32x32_i8, 2 f32_32x32x2_f32 ; PK: 0 v_fma_f32, 1 v_pk_fma_f32 ; NV: VALU ops per iteration
template <int MF, int PK, int NV>
__global__ void __launch_bounds__(256, 4) ko(float* out, int iters, float s) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
  v16i c0, c1;
  v16f f0, f1;
  for (int i = 0; i < 16; ++i) { c0[i] = i; c1[i] = -i; f0[i] = i; f1[i] = -i; }
  v2f p[16];
  for (int i = 0; i < 16; ++i) p[i] = v2f{(float)threadIdx.x + i, 1.0f * i};
  v2f ps = {s, s};
  float fa = s, fb = s * 0.5f;
  for (int it = 0; it < iters; it += 2) {
    if (MF == 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    if (MF == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(fa), "v"(fb));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i % 16]) : "v"(ps));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(p[i % 16][0]) : "v"(s));
    }
    if (MF == 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
    if (MF == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(f1) : "v"(fa), "v"(fb));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(i + 8) % 16]) : "v"(ps));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(p[(i + 8) % 16][0]) : "v"(s));
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += p[i][0] + p[i][1] + c0[i] + c1[i] + f0[i] + f1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K> float timeit(K kern, int grid, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 64, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
template <int MF, int PK, int NV> void row(float* out, const char* name) {
  const int iters = 8000;
  printf("%-28s NV=%2d:", name, NV);
  for (int w = 1; w <= 4; ++w) printf("  w%d %6.1f", w, timeit(ko<MF, PK, NV>, 256 * w, out, iters) * 1e6 / ((double)iters * w));
  printf("   ns per (MFMA + NV ops) per SIMD\n");
}
template <int MF, int PK> void table(float* out, const char* name) {
  row<MF, PK, 0>(out, name); row<MF, PK, 4>(out, name); row<MF, PK, 8>(out, name); row<MF, PK, 12>(out, name);
  row<MF, PK, 16>(out, name); row<MF, PK, 24>(out, name); row<MF, PK, 32>(out, name);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  table<0, 0>(out, "no MFMA + v_fma_f32");
  table<0, 1>(out, "no MFMA + v_pk_fma_f32");
  table<1, 0>(out, "i8 32x32x32 + v_fma_f32");
  table<1, 1>(out, "i8 32x32x32 + v_pk_fma_f32");
  table<2, 0>(out, "f32 32x32x2 + v_fma_f32");
  table<2, 1>(out, "f32 32x32x2 + v_pk_fma_f32");
  return 0;
}
