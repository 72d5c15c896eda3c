// MFMA issue cost (cycles per instruction per SIMD) for the instructions the MMQ kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(256) k(int iters, float* out) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
  v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
  v16f f0 = {}, f1 = {}, f2 = {}, f3 = {};
  v4f g0 = {}, g1 = {}, g2 = {}, g3 = {};
  v4i h0 = {}, h1 = {}, h2 = {}, h3 = {};
  float fa = threadIdx.x, fb = 2.0f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    } else if (KIND == 1) {
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f0, 0, 0, 0);
      f1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f2, 0, 0, 0);
      f3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f3, 0, 0, 0);
    } else if (KIND == 2) {
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, g1, 0, 0, 0);
      g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, g2, 0, 0, 0);
      g3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, g3, 0, 0, 0);
    } else if (KIND == 3) {
      h0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, h0, 0, 0, 0);
      h1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, h1, 0, 0, 0);
      h2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, h2, 0, 0, 0);
      h3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, h3, 0, 0, 0);
    } else if (KIND == 4) {
      long la = threadIdx.x, lb = 77;
      c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(la, lb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(la, lb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(la, lb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(la, lb, c3, 0, 0, 0);
    } else if (KIND == 5) {   // f32 32x32x1 (2 blocks): 64x32... skip; use xf32-free: 32x32x2 dependent chain
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, f0, 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + f0[i] + f1[i] + f2[i] + f3[i];
  for (int i = 0; i < 4; ++i) s += g0[i] + g1[i] + g2[i] + g3[i] + h0[i] + h1[i] + h2[i] + h3[i];
  if (s == 1.2345f) out[0] = s;
}
template <int KIND> void run(const char* name, int wgs_per_cu, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256 * wgs_per_cu, 256>>>(iters, out);
  hipEventRecord(e0);
  k<KIND><<<256 * wgs_per_cu, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * 4 * wgs_per_cu);  // ns per MFMA per SIMD
  printf("%-34s waves/SIMD %d: %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", name, wgs_per_cu, per, per * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 4);
  for (int w : {1, 2}) {
    run<0>("i32_32x32x32_i8", w, out);
    run<4>("i32_32x32x16_i8", w, out);
    run<3>("i32_16x16x64_i8", w, out);
    run<1>("f32_32x32x2_f32", w, out);
    run<2>("f32_16x16x4_f32", w, out);
    run<5>("f32_32x32x2_f32 dependent chain", w, out);
  }
  return 0;
}
