// ubench_lds_unaligned.hip — does a ds_read_b128 / b64 / b32 at a 1- / 2-byte aligned LDS address return the right bytes on gfx950?
// build: hipcc -O2 --offload-arch=gfx950 -o scripts/_build/ubench_lds_unaligned scripts/ubench_lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned v2u_a1 __attribute__((ext_vector_type(2), aligned(1)));
struct __attribute__((packed)) u32_a1 { unsigned v; };
__global__ void k(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)(i * 7 + 3);
  __syncthreads();
  const int off = threadIdx.x;   // 0 .. 63
  const v4u_a1 a = *(const v4u_a1*)(lds + off);
  const v2u_a1 b = *(const v2u_a1*)(lds + 128 + off);
  const unsigned c = ((const u32_a1*)(lds + 256 + off))->v;
  out[threadIdx.x * 8 + 0] = a[0]; out[threadIdx.x * 8 + 1] = a[1]; out[threadIdx.x * 8 + 2] = a[2]; out[threadIdx.x * 8 + 3] = a[3];
  out[threadIdx.x * 8 + 4] = b[0]; out[threadIdx.x * 8 + 5] = b[1]; out[threadIdx.x * 8 + 6] = c;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 4);
  k<<<1, 64>>>(d); hipDeviceSynchronize();
  std::vector<unsigned> h(64 * 8); hipMemcpy(h.data(), d, 64 * 8 * 4, hipMemcpyDeviceToHost);
  auto byte = [](int i) { return (unsigned)(unsigned char)(i * 7 + 3); };
  auto word = [&](int i) { return byte(i) | byte(i + 1) << 8 | byte(i + 2) << 16 | byte(i + 3) << 24; };
  int bad128[4] = {0, 0, 0, 0}, bad64[4] = {0, 0, 0, 0}, bad32[4] = {0, 0, 0, 0};
  for (int t = 0; t < 64; ++t) {
    bool ok = true; for (int w = 0; w < 4; ++w) ok &= h[t * 8 + w] == word(t + 4 * w);
    bad128[t & 3] += !ok;
    bad64[t & 3] += !(h[t * 8 + 4] == word(128 + t) && h[t * 8 + 5] == word(132 + t));
    bad32[t & 3] += !(h[t * 8 + 6] == word(256 + t));
  }
  for (int a = 0; a < 4; ++a) printf("address mod 4 = %d: wrong results out of 16 offsets: b128 %d  b64 %d  b32 %d\n", a, bad128[a], bad64[a], bad32[a]);
  return 0;
}
