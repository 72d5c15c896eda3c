// checks for the fp16 min-term idea: (1) operand layout and exactness of v_mfma_f32_32x32x16_f16 with fp16 SUBNORMAL
// inputs (are they flushed?), (2) its issue cost next to the int8 and f32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_check(const _Float16* A /*[32][16]*/, const _Float16* B /*[16][32]*/, float* C /*[32][32]*/) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  v8h a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  v16f c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  // C layout of the 32x32 MFMAs: register i of lane (r = column, h) holds row 8 (i / 4) + 4 h + (i % 4)
  for (int i = 0; i < 16; ++i) C[(8 * (i / 4) + 4 * h + (i % 4)) * 32 + r] = c[i];
}

template <int MF>
__global__ void __launch_bounds__(256, 4) k_time(float* out, int iters) {
  v8h a, b; for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x + j); b[j] = (_Float16)j; }
  v16f c0 = {}, c1 = {};
  for (int it = 0; it < iters; it += 2) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
  }
  float r = 0; for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  std::vector<_Float16> A(32 * 16), B(16 * 32);
  srand(1);
  auto rnd_h = [&](int kind) -> _Float16 {
    unsigned short bits;
    if (kind == 0) bits = rand() & 0x03FF;                       // subnormal
    else if (kind == 1) bits = (rand() & 0x7FFF) % 0x7C00;       // any finite positive
    else bits = 0x2000 + (rand() & 0x1FFF);                      // moderate
    if (rand() & 1) bits |= 0x8000;
    _Float16 v; __builtin_memcpy(&v, &bits, 2); return v;
  };
  for (int pass = 0; pass < 3; ++pass) {
    for (auto& v : A) v = rnd_h(pass == 0 ? 0 : pass == 1 ? 2 : (rand() % 3));
    for (auto& v : B) v = rnd_h(pass == 0 ? 2 : pass == 1 ? 0 : (rand() % 3 == 1 ? 2 : rand() % 3));
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(32 * 32); hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0, mag = 0; for (int k = 0; k < 16; ++k) { ref += (double)A[i * 16 + k] * (double)B[k * 32 + j]; mag += fabs((double)A[i * 16 + k] * (double)B[k * 32 + j]); }
      const double err = fabs(C[i * 32 + j] - ref) / (mag + 1e-300);
      if (err > worst) worst = err;
      if (err > 2e-6) ++bad;
    }
    printf("pass %d (%s): worst |err| / sum|terms| = %.3g, outliers %d\n", pass, pass == 0 ? "A subnormal" : pass == 1 ? "B subnormal" : "mixed", worst, bad);
  }
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 3; ++w) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k_time<0>, dim3(256 * w), dim3(256), 0, 0, out, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time<0>, dim3(256 * w), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("f32_32x32x16_f16 waves/SIMD %d: %.2f ns per MFMA per SIMD\n", w, ms * 1e6 / ((double)iters * w));
  }
  return 0;
}
