// ggq_cpu.cpp — host CPU twin of the dequantise op (product code, not the oracle).
//
// Replaces the reference's ggml-cpu op: ggml_dequantize (ggml-cpu/custom_ops.cpp:11-36) and
// dequantize_row_q{4_0,4_1,5_0,5_1,8_0} (ggml-cpu/ggml-quants.hpp:4-112).  fp32 output,
// arithmetic identical to the reference (int * float, + float for the _1 formats; build with
// -ffp-contract=off so x*d + m stays two roundings as in the reference's x86 build).
// Unlike the reference it is row-partitioned over threads, rejects unknown types, and decodes eight elements per
// instruction with AVX2 where the host has it (the reference's loops are scalar).  The vector
// path performs the same IEEE operations per element — int -> float conversion, one fp32 multiply, one fp32 add
// for the _1 formats (no FMA: the functions are compiled for "avx2" only) — so it is bit-identical to the scalar
// path and to the reference (tests/test_cpu_op.py).
#include <cstdint>
#include <immintrin.h>
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

// ---- AVX2: 8 outputs per instruction ----
#define GGQ_AVX2 __attribute__((target("avx2")))
// bytes 0..15 of v (values 0..255 or signed) -> two float vectors times d (+ m), stored to o[0..15]
template <bool SIGNED, bool HAS_M>
GGQ_AVX2 inline void store16(__m128i v, int offset, __m256 d, __m256 m, float* o) {
  const __m256i off = _mm256_set1_epi32(offset);
  const __m128i hi8 = _mm_srli_si128(v, 8);
  __m256i i0 = SIGNED ? _mm256_cvtepi8_epi32(v) : _mm256_cvtepu8_epi32(v);
  __m256i i1 = SIGNED ? _mm256_cvtepi8_epi32(hi8) : _mm256_cvtepu8_epi32(hi8);
  i0 = _mm256_sub_epi32(i0, off); i1 = _mm256_sub_epi32(i1, off);
  __m256 f0 = _mm256_mul_ps(_mm256_cvtepi32_ps(i0), d), f1 = _mm256_mul_ps(_mm256_cvtepi32_ps(i1), d);
  if (HAS_M) { f0 = _mm256_add_ps(f0, m); f1 = _mm256_add_ps(f1, m); }
  _mm256_storeu_ps(o, f0); _mm256_storeu_ps(o + 8, f1);
}
// bit e of qh -> 0x10 in byte e (e = 0..31)
GGQ_AVX2 inline __m256i spread_qh(uint32_t qh) {
  const __m256i idx = _mm256_setr_epi8(0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3);
  const __m256i bit = _mm256_setr_epi8(1, 2, 4, 8, 16, 32, 64, (char)128, 1, 2, 4, 8, 16, 32, 64, (char)128, 1, 2, 4, 8, 16, 32, 64,
                                       (char)128, 1, 2, 4, 8, 16, 32, 64, (char)128);
  const __m256i b = _mm256_shuffle_epi8(_mm256_set1_epi32((int)qh), idx);
  return _mm256_and_si256(_mm256_cmpeq_epi8(_mm256_and_si256(b, bit), bit), _mm256_set1_epi8(0x10));
}
template <int OFFSET, bool HAS_M, bool HAS_QH, int BS>
GGQ_AVX2 void deq_nibble_blocks_avx2(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  constexpr int QS = 2 + (HAS_M ? 2 : 0) + (HAS_QH ? 4 : 0);
  const __m128i low4 = _mm_set1_epi8(0x0F);
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * BS;
    const __m256 d = _mm256_set1_ps(h2f(rd16(b)));
    const __m256 m = _mm256_set1_ps(HAS_M ? h2f(rd16(b + 2)) : 0.0f);
    const __m128i q = _mm_loadu_si128((const __m128i*)(b + QS));
    __m128i lo = _mm_and_si128(q, low4), hi = _mm_and_si128(_mm_srli_epi16(q, 4), low4);
    if (HAS_QH) {
      const __m256i h = spread_qh(rd32(b + (HAS_M ? 4 : 2)));
      lo = _mm_or_si128(lo, _mm256_castsi256_si128(h));
      hi = _mm_or_si128(hi, _mm256_extracti128_si256(h, 1));
    }
    store16<false, HAS_M>(lo, OFFSET, d, m, y + i * 32);
    store16<false, HAS_M>(hi, OFFSET, d, m, y + i * 32 + 16);
  }
}
GGQ_AVX2 void deq_q8_0_avx2(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * 34;
    const __m256 d = _mm256_set1_ps(h2f(rd16(b))), z = _mm256_setzero_ps();
    store16<true, false>(_mm_loadu_si128((const __m128i*)(b + 2)), 0, d, z, y + i * 32);
    store16<true, false>(_mm_loadu_si128((const __m128i*)(b + 18)), 0, d, z, y + i * 32 + 16);
  }
}

// ---- AVX-512 (F + BW + VL): 16 outputs per instruction, the same operations per element (integer subtract, exact
//      int -> float conversion, one multiply, one add: no fused multiply-add), so the results stay bit-identical ----
#define GGQ_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl")))
template <bool SIGNED, bool HAS_M>
GGQ_AVX512 inline void store16_512(__m128i v, int offset, __m512 d, __m512 m, float* o) {
  __m512i i = SIGNED ? _mm512_cvtepi8_epi32(v) : _mm512_cvtepu8_epi32(v);
  i = _mm512_sub_epi32(i, _mm512_set1_epi32(offset));
  __m512 f = _mm512_mul_ps(_mm512_cvtepi32_ps(i), d);
  if (HAS_M) f = _mm512_add_ps(f, m);
  _mm512_storeu_ps(o, f);
}
template <int OFFSET, bool HAS_M, bool HAS_QH, int BS>
GGQ_AVX512 void deq_nibble_blocks_avx512(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  constexpr int QS = 2 + (HAS_M ? 2 : 0) + (HAS_QH ? 4 : 0);
  const __m128i low4 = _mm_set1_epi8(0x0F);
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * BS;
    const __m512 d = _mm512_set1_ps(h2f(rd16(b)));
    const __m512 m = _mm512_set1_ps(HAS_M ? h2f(rd16(b + 2)) : 0.0f);
    const __m128i q = _mm_loadu_si128((const __m128i*)(b + QS));
    __m128i lo = _mm_and_si128(q, low4), hi = _mm_and_si128(_mm_srli_epi16(q, 4), low4);
    if (HAS_QH) {   // bit e of qh -> 0x10 in byte e: one masked broadcast
      const __m256i h = _mm256_maskz_set1_epi8((__mmask32)rd32(b + (HAS_M ? 4 : 2)), 0x10);
      lo = _mm_or_si128(lo, _mm256_castsi256_si128(h));
      hi = _mm_or_si128(hi, _mm256_extracti128_si256(h, 1));
    }
    store16_512<false, HAS_M>(lo, OFFSET, d, m, y + i * 32);
    store16_512<false, HAS_M>(hi, OFFSET, d, m, y + i * 32 + 16);
  }
}
GGQ_AVX512 void deq_q8_0_avx512(const uint8_t* w, float* y, int64_t b0, int64_t b1) {
  for (int64_t i = b0; i < b1; ++i) {
    const uint8_t* b = w + i * 34;
    const __m512 d = _mm512_set1_ps(h2f(rd16(b))), z = _mm512_setzero_ps();
    store16_512<true, false>(_mm_loadu_si128((const __m128i*)(b + 2)), 0, d, z, y + i * 32);
    store16_512<true, false>(_mm_loadu_si128((const __m128i*)(b + 18)), 0, d, z, y + i * 32 + 16);
  }
}

bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
bool have_avx512() {
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
  return v;
}

}  // namespace

extern "C" const char* ggq_cpu_simd_name(void) { return have_avx512() ? "avx512" : have_avx2() ? "avx2" : "scalar"; }

extern "C" int ggq_cpu_dequantize_f32(const void* w, float* out, int type, int64_t m, int64_t n, int nthreads) {
  return ggq_cpu_dequantize_f32_ex(w, out, type, m, n, nthreads, 1);
}

extern "C" int ggq_cpu_dequantize_f32_ex(const void* w, float* out, int type, int64_t m, int64_t n,
                                         int nthreads, int simd) {
  if (m < 0 || n < 0) return GGQ_ERR_ARG;
  range_fn fn = nullptr;
  // simd: 0 scalar, 2 at most AVX2, any other value the widest unit of the host
  const bool v5 = simd != 0 && simd != 2 && have_avx512();
  const bool v = simd != 0 && have_avx2();
#define GGQ_PICK(A512, A2, SC) (v5 ? (range_fn)A512 : v ? (range_fn)A2 : (range_fn)SC)
  switch (type) {  // the five formats of ggml-cpu/custom_ops.cpp:16-34
    case GGQ_TYPE_Q4_0: fn = GGQ_PICK((deq_nibble_blocks_avx512<8, false, false, 18>), (deq_nibble_blocks_avx2<8, false, false, 18>), (deq_nibble_blocks<8, false, false, 18>)); break;
    case GGQ_TYPE_Q4_1: fn = GGQ_PICK((deq_nibble_blocks_avx512<0, true, false, 20>), (deq_nibble_blocks_avx2<0, true, false, 20>), (deq_nibble_blocks<0, true, false, 20>)); break;
    case GGQ_TYPE_Q5_0: fn = GGQ_PICK((deq_nibble_blocks_avx512<16, false, true, 22>), (deq_nibble_blocks_avx2<16, false, true, 22>), (deq_nibble_blocks<16, false, true, 22>)); break;
    case GGQ_TYPE_Q5_1: fn = GGQ_PICK((deq_nibble_blocks_avx512<0, true, true, 24>), (deq_nibble_blocks_avx2<0, true, true, 24>), (deq_nibble_blocks<0, true, true, 24>)); break;
    case GGQ_TYPE_Q8_0: fn = GGQ_PICK(deq_q8_0_avx512, deq_q8_0_avx2, deq_q8_0); break;
    default: return GGQ_ERR_TYPE;
  }
#undef GGQ_PICK
  const int64_t k = m * n;
  if (k % 32) return GGQ_ERR_SHAPE;
  if (k == 0) return GGQ_OK;
  if (!w || !out) return GGQ_ERR_ARG;
  const int64_t nb = k / 32;
  int nt = nthreads < 1 ? 1 : nthreads;
  // threads are created per call (no pool: the library keeps no state), so a thread is only worth starting for at least
  // 4 MiB of output (32768 blocks) — with 64 threads on 64 MiB the call measured thread creation, not dequantisation
  const int64_t max_useful = (nb + 32767) / 32768;
  if ((int64_t)nt > max_useful) nt = (int)max_useful;
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
