// micro-benchmark: VALU issue rates on gfx950 (cvt / fma / pk_fma / imul24), 1-2-4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, ps = {s, s};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent fma
      a0 = __builtin_fmaf(a0, s, 1.0f); a1 = __builtin_fmaf(a1, s, 1.0f); a2 = __builtin_fmaf(a2, s, 1.0f); a3 = __builtin_fmaf(a3, s, 1.0f);
      a4 = __builtin_fmaf(a4, s, 1.0f); a5 = __builtin_fmaf(a5, s, 1.0f); a6 = __builtin_fmaf(a6, s, 1.0f); a7 = __builtin_fmaf(a7, s, 1.0f);
    } else if (MODE == 1) {  // 4 pk_fma (8 flops-pairs)
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ps));
    } else if (MODE == 2) {  // 4 cvt + 4 fma
      a0 = __builtin_fmaf((float)i0, s, a0); a1 = __builtin_fmaf((float)i1, s, a1); a2 = __builtin_fmaf((float)i2, s, a2); a3 = __builtin_fmaf((float)i3, s, a3);
      i0 += it; i1 ^= it; i2 -= it; i3 += 3;
    } else if (MODE == 3) {  // 8 cvt
      asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n"
                   "v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));
    } else if (MODE == 4) {  // 8 v_mul_i32_i24
      asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4\n"
                   "v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4\n"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));
    } else if (MODE == 5) {  // 8 v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                   "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
}
template <int MODE> void run(const char* name, int ops_per_iter) {
  float* out; hipMalloc(&out, 256 * 1024 * 4 * 16);
  for (int wpb : {256, 512, 1024}) {  // threads per block, 1 block per CU: 1, 2, 4 waves per SIMD
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, 100, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_simd = (double)iters * ops_per_iter * (wpb / 64) / 4.0;
    printf("%-10s waves/SIMD=%d : %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wpb / 256,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  }
  hipFree(out);
}
int main() {
  run<0>("fma", 8); run<1>("pk_fma", 4); run<2>("cvt+fma", 12); run<3>("cvt", 8); run<4>("mul_i24", 8); run<5>("mul_f32", 8);
  return 0;
}
