// inner-loop roofline of the MMQ consumer: LDS fragment/scale reads + int8 MFMA + 2 FMAs per triple,
// operands resident in LDS, no global traffic, no barriers.  Reports ns per 32x32x32 tile-group per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float as_f32(int v) { return __builtin_bit_cast(float, v); }
constexpr int WROW = 272;
// MODE 0: full; 1: no apply; 2: no LDS reads in loop (hoisted); 3: apply with plain fma on cvt (3 ops)
template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (int i = threadIdx.x; i < 60000 / 4; i += NT) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const uint8_t* wt = lds;                 // 32 x 272
  const float* sc = (const float*)(lds + 8704);   // [2][8][32]
  const uint8_t* at = lds + 8704 + 2048;   // [2][64][144]
  v16f acc; for (int i = 0; i < 16; ++i) acc[i] = 0;
  v16i magic; for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
  asm volatile("" : "+v"(magic));
  const int tl = (wave & 1) * 32 + r;
  v4i a_h, b_h; v4f sa_h[4]; float bs_h;
  {
    const int g = 0; const uint8_t* ablk = at + ((g >> 2) * 64 + tl) * 144;
    a_h = *(const v4i*)(wt + r * WROW + 32 * g + 16 * h); b_h = *(const v4i*)(ablk + 16 + 32 * (g & 3) + 16 * h);
    for (int qd = 0; qd < 4; ++qd) sa_h[qd] = *(const v4f*)(sc + g * 32 + 8 * qd + 4 * h);
    bs_h = as_f32(*(const int*)(ablk)) * 1e-30f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = ((wave >> 1) * 2 + j) & 7;
      const uint8_t* ablk = at + ((g >> 2) * 64 + tl) * 144;
      v4i a, b; v4f sa[4]; float bs;
      if (MODE == 2) { a = a_h; b = b_h; for (int qd = 0; qd < 4; ++qd) sa[qd] = sa_h[qd]; bs = bs_h; }
      else {
        const uint32_t dsw = *(const uint32_t*)(ablk + 4 * (g & 3));
        bs = (float)__builtin_bit_cast(_Float16, (uint16_t)dsw);
        a = *(const v4i*)(wt + r * WROW + 32 * g + 16 * h);
        b = *(const v4i*)(ablk + 16 + 32 * (g & 3) + 16 * h);
        for (int qd = 0; qd < 4; ++qd) sa[qd] = *(const v4f*)(sc + g * 32 + 8 * qd + 4 * h);
      }
      const float nmbs = -(12582912.0f * bs);
      v16i c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, magic, 0, 0, 0);
      if (MODE == 1) { acc[0] += as_f32(c0[0]); acc[5] += as_f32(c0[15]) + sa[3][3] + bs; }
      else if (MODE == 3) {
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf((float)(c0[i]) * bs, sa[i >> 2][i & 3], acc[i]);
      } else {
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(__builtin_fmaf(as_f32(c0[i]), bs, nmbs), sa[i >> 2][i & 3], acc[i]);
      }
    }
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MODE, int NT> void run(const char* name, int wg_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  int iters = 4000;
  auto kern = k<MODE, NT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(NT), 60000, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(NT), 60000, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double tg_per_simd = (double)iters * 2 * (NT / 64) * wg_per_cu / 4.0;
  printf("%-28s NT=%4d wg/cu=%d (waves/SIMD %.1f): %.1f ns per tile-group per SIMD  -> full problem (172/SIMD) %.1f us\n", name, NT, wg_per_cu,
         NT / 256.0 * wg_per_cu, ms * 1e6 / tg_per_simd, ms * 1e6 / tg_per_simd * 172 / 1e3);
  hipFree(out);
}
int main() {
  run<0, 512>("full", 1); run<0, 512>("full", 2); run<0, 1024>("full", 1);
  run<1, 512>("mfma+lds only", 1); run<1, 512>("mfma+lds only", 2);
  run<2, 512>("mfma+apply (no lds)", 1); run<2, 512>("mfma+apply (no lds)", 2);
  run<3, 512>("cvt+mul+fma apply", 2);
  return 0;
}
