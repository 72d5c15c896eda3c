// ggq_cpu.cpp — host CPU twin of the dequantise op (product code, not the oracle).
//
// Replaces the reference's ggml-cpu op: ggml_dequantize (ggml-cpu/custom_ops.cpp:11-36) and
// dequantize_row_q{4_0,4_1,5_0,5_1,8_0} (ggml-cpu/ggml-quants.hpp:4-112).  fp32 output,
// arithmetic identical to the reference (int * float, + float for the _1 formats; build with
// -ffp-contract=off so x*d + m stays two roundings as in the reference's x86 build).
// Unlike the reference it is row-partitioned over threads and rejects unknown types.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../../include/ggq.h"

namespace {

inline float h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) { u = sign; }
    else {  // subnormal: renormalise
      int s = 0; uint32_t mm = m;
      while (!(mm & 0x400u)) { mm <<= 1; ++s; }
      u = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (e == 31) { u = sign | 0x7f800000u | (m << 13); }
  else { u = sign | ((e + 112u) << 23) | (m << 13); }
  float f; std::memcpy(&f, &u, 4); return f;
}
inline uint16_t rd16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

template <int OFFSET, bool HAS_M, bool HAS_QH, int BS>
void deq_nibble_blocks(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  constexpr int QS = 2 + (HAS_M ? 2 : 0) + (HAS_QH ? 4 : 0);
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * BS;
    const float d = h2f(rd16(b));
    const float m = HAS_M ? h2f(rd16(b + 2)) : 0.0f;
    const uint32_t qh = HAS_QH ? rd32(b + (HAS_M ? 4 : 2)) : 0u;
    float* o = y + i * 32;
    for (int j = 0; j < 16; ++j) {
      int x0 = b[QS + j] & 0x0F, x1 = b[QS + j] >> 4;
      if (HAS_QH) { x0 |= ((qh >> j) & 1) << 4; x1 |= ((qh >> (j + 16)) & 1) << 4; }
      x0 -= OFFSET; x1 -= OFFSET;
      if (HAS_M) { o[j] = x0 * d + m; o[j + 16] = x1 * d + m; }
      else { o[j] = x0 * d; o[j + 16] = x1 * d; }
    }
  }
}

void deq_q8_0(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * 34;
    const float d = h2f(rd16(b));
    for (int j = 0; j < 32; ++j) y[i * 32 + j] = (int8_t)b[2 + j] * d;
  }
}

typedef void (*range_fn)(const uint8_t*, float*, int64_t, int64_t);

}  // namespace

extern "C" int ggq_cpu_dequantize_f32(const void* w, float* out, int type, int64_t m, int64_t n,
                                      int nthreads) {
  if (m < 0 || n < 0) return GGQ_ERR_ARG;
  range_fn fn = nullptr;
  switch (type) {  // the five formats of ggml-cpu/custom_ops.cpp:16-34
    case GGQ_TYPE_Q4_0: fn = deq_nibble_blocks<8, false, false, 18>; break;
    case GGQ_TYPE_Q4_1: fn = deq_nibble_blocks<0, true, false, 20>; break;
    case GGQ_TYPE_Q5_0: fn = deq_nibble_blocks<16, false, true, 22>; break;
    case GGQ_TYPE_Q5_1: fn = deq_nibble_blocks<0, true, true, 24>; break;
    case GGQ_TYPE_Q8_0: fn = deq_q8_0; break;
    default: return GGQ_ERR_TYPE;
  }
  const int64_t k = m * n;
  if (k % 32) return GGQ_ERR_SHAPE;
  if (k == 0) return GGQ_OK;
  if (!w || !out) return GGQ_ERR_ARG;
  const int64_t nb = k / 32;
  int nt = nthreads < 1 ? 1 : nthreads;
  if ((int64_t)nt > nb) nt = (int)nb;
  if (nt == 1) { fn((const uint8_t*)w, out, 0, nb); return GGQ_OK; }
  std::vector<std::thread> th;
  const int64_t per = (nb + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
    if (b0 >= b1) break;
    th.emplace_back(fn, (const uint8_t*)w, out, b0, b1);
  }
  for (auto& x : th) x.join();
  return GGQ_OK;
}
