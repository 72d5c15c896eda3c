// ggq_cpu_mmq.cpp — host CPU twin of the quantised GEMM (product code, not the oracle): ggq_cpu_quantize_q8_1_mmq +
// ggq_cpu_mul_mat_q for Q4_K, Q4_0 and Q8_0 — the host baseline of the headline metric (the reference itself has no CPU
// matmul: ggml-cpu/custom_ops.cpp:16-34 only dequantises).
//
// Arithmetic: the MMQ canon of the GPU path restated per (row, token) — quantize_mmq_q8_1 (HK/ggml/mmq.cu:109-154:
// d = amax / 127, q = roundf(x / d), sum = the 32-lane xor-butterfly as a pairing tree) and the float sequences of the
// tensor-core bodies (Q4_K: mmq.cuh:1274-1363, per 64 elements tmpd += float(C * sc) * d8, tmpm += m * s8,
// sum += dall * tmpd - dmin * tmpm; Q4_0: :330-394, d4 * d8 * C with the nibbles minus 8; Q8_0: :913-974, C * d * d8),
// evaluated in exactly that order with -ffp-contract=off.  Only the exact integer contractions C are vectorised:
// 32 unsigned x signed bytes per instruction with AVX2 (vpmaddubsw + vpmaddwd) or AVX-512 VNNI (vpdpbusd); the eight
// group sums of a super-block come out of one horizontal-add tree.  Integer sums are exact in any order, so the scalar,
// AVX2 and VNNI paths give bit-identical results (tests/test_cpu_op.py checks them against the oracle).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <immintrin.h>
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
    else { int s = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++s; } u = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3ffu) << 13); }
  } else if (e == 31) { u = sign | 0x7f800000u | (m << 13); }
  else { u = sign | ((e + 112u) << 23) | (m << 13); }
  float f; std::memcpy(&f, &u, 4); return f;
}
inline uint16_t f2h(float f) {   // round to nearest even, overflow -> inf, NaN stays NaN
  uint32_t x; std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax < 0x38800000u) {
    if (ax < 0x33000000u) return (uint16_t)sign;
    const int e = (int)(ax >> 23); const uint32_t m = (ax & 0x7fffffu) | 0x800000u; const int shift = 126 - e;
    uint32_t q = m >> shift; const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
  }
  const uint32_t e = (ax >> 23) - 112u, m = ax & 0x7fffffu;
  uint32_t q = (e << 10) | (m >> 13); const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
  return (uint16_t)(sign | q);
}
inline uint16_t rd16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
bool have_vnni() {
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vnni");
  return v;
}

// ---- the eight 32-element integer dots of 256 weights (unsigned bytes wq[256]) against a token's int8 aq[8 groups][32] ----
void dots8_scalar(const uint8_t* wq, const int8_t* const* aq, int32_t C[8]) {
  for (int g = 0; g < 8; ++g) {
    int s = 0;
    for (int j = 0; j < 32; ++j) s += (int)wq[32 * g + j] * aq[g][j];
    C[g] = s;
  }
}
__attribute__((target("avx2"))) inline __m256i hsum8(__m256i v[8]) {
  const __m256i t0 = _mm256_hadd_epi32(v[0], v[1]), t1 = _mm256_hadd_epi32(v[2], v[3]);
  const __m256i t2 = _mm256_hadd_epi32(v[4], v[5]), t3 = _mm256_hadd_epi32(v[6], v[7]);
  const __m256i u0 = _mm256_hadd_epi32(t0, t1), u1 = _mm256_hadd_epi32(t2, t3);
  return _mm256_add_epi32(_mm256_permute2x128_si256(u0, u1, 0x20), _mm256_permute2x128_si256(u0, u1, 0x31));
}
__attribute__((target("avx2"))) void dots8_avx2(const uint8_t* wq, const int8_t* const* aq, int32_t C[8]) {
  const __m256i ones = _mm256_set1_epi16(1);
  __m256i v[8];
  for (int g = 0; g < 8; ++g) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(wq + 32 * g)), a = _mm256_loadu_si256((const __m256i*)aq[g]);
    v[g] = _mm256_madd_epi16(_mm256_maddubs_epi16(w, a), ones);   // |pair sums| <= 2 * 255 * 128: no int16 saturation for w <= 127
  }
  _mm256_storeu_si256((__m256i*)C, hsum8(v));
}
__attribute__((target("avx2,avx512f,avx512vl,avx512vnni"))) void dots8_vnni(const uint8_t* wq, const int8_t* const* aq, int32_t C[8]) {
  __m256i v[8];
  for (int g = 0; g < 8; ++g) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(wq + 32 * g)), a = _mm256_loadu_si256((const __m256i*)aq[g]);
    v[g] = _mm256_dpbusd_epi32(_mm256_setzero_si256(), w, a);
  }
  _mm256_storeu_si256((__m256i*)C, hsum8(v));
}
typedef void (*dots8_fn)(const uint8_t*, const int8_t* const*, int32_t*);
dots8_fn pick_dots(int simd) {
  if (simd == 0) return dots8_scalar;
  if (simd != 2 && have_vnni()) return dots8_vnni;
  return have_avx2() ? dots8_avx2 : dots8_scalar;
}

// block_q8_1_mmq accessors (HK/ggml/mmq.cuh:176-181): index (g32 / 4) * batch + token, 144 bytes
inline const uint8_t* qblk(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) { return q8 + ((g32 / 4) * batch + t) * 144; }
inline const int8_t* qqs(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) { return (const int8_t*)(qblk(q8, batch, t, g32) + 16 + 32 * (g32 % 4)); }

void get_scale_min_k4(int j, const uint8_t* q, int& d, int& m) {   // HK/ggml/dequantize.cuh:154-161
  if (j < 4) { d = q[j] & 63; m = q[j + 4] & 63; }
  else { d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// rows [r0, r1) of Y = X * W^T, all tokens
void rows_q4_k(const uint8_t* w, const uint8_t* q8, float* y, int64_t batch, int64_t k, int64_t n_rows, int64_t r0, int64_t r1, dots8_fn dots,
               const float* ad, const float* as /* [batch][k/32]: the groups' d and sum as floats (fp16 -> fp32 is exact) */) {
  const int64_t nsb = k / 256, nb = k / 32;
  alignas(32) uint8_t wq[256];
  for (int64_t row = r0; row < r1; ++row) {
    for (int64_t t = 0; t < batch; ++t) y[t * n_rows + row] = 0.0f;
    for (int64_t ib = 0; ib < nsb; ++ib) {
      const uint8_t* b = w + (row * nsb + ib) * 144;
      for (int p = 0; p < 4; ++p)
        for (int j = 0; j < 32; ++j) { wq[64 * p + j] = b[16 + 32 * p + j] & 0xF; wq[64 * p + 32 + j] = b[16 + 32 * p + j] >> 4; }
      int sc[8], mn[8];
      for (int g = 0; g < 8; ++g) get_scale_min_k4(g, b + 4, sc[g], mn[g]);
      const float dall = h2f(rd16(b)), dmin = h2f(rd16(b + 2));
      for (int64_t t = 0; t < batch; ++t) {
        const int8_t* aq[8];
        for (int g = 0; g < 8; ++g) aq[g] = qqs(q8, batch, t, ib * 8 + g);
        alignas(32) int32_t C[8];
        dots(wq, aq, C);
        float sum = y[t * n_rows + row];
        for (int p = 0; p < 4; ++p) {   // one k0 step of the reference = 64 elements = two Q8_1 groups (mmq.cuh:1352-1359)
          float tmpd = 0.0f, tmpm = 0.0f;
          for (int h = 0; h < 2; ++h) {
            const int g = 2 * p + h;
            tmpd += (float)(C[g] * sc[g]) * ad[t * nb + ib * 8 + g];
            tmpm += (float)mn[g] * as[t * nb + ib * 8 + g];
          }
          sum += dall * tmpd - dmin * tmpm;
        }
        y[t * n_rows + row] = sum;
      }
    }
  }
}

template <bool Q8>
void rows_legacy(const uint8_t* w, const uint8_t* q8, float* y, int64_t batch, int64_t k, int64_t n_rows, int64_t r0, int64_t r1, dots8_fn dots,
                 const int32_t* asum /* [batch][k/32] sum of the int8 of a group: the -8 offset of Q4_0 */, const float* ad) {
  const int64_t nb = k / 32;
  constexpr int BS = Q8 ? 34 : 18;
  alignas(32) uint8_t wq[256];
  alignas(32) int8_t aflip[256];
  for (int64_t row = r0; row < r1; ++row) {
    for (int64_t t = 0; t < batch; ++t) y[t * n_rows + row] = 0.0f;
    for (int64_t g0 = 0; g0 < nb; g0 += 8) {
      const int ng = (int)(nb - g0 < 8 ? nb - g0 : 8);
      float d[8];
      std::memset(wq, 0, sizeof(wq));
      for (int g = 0; g < ng; ++g) {
        const uint8_t* b = w + (row * nb + g0 + g) * BS;
        d[g] = h2f(rd16(b));
        if (Q8) for (int j = 0; j < 32; ++j) wq[32 * g + j] = b[2 + j];                      // signed bytes: handled by the sign flip below
        else for (int j = 0; j < 16; ++j) { wq[32 * g + j] = b[2 + j] & 0xF; wq[32 * g + 16 + j] = b[2 + j] >> 4; }
      }
      uint8_t wabs[256];
      if (Q8) for (int i = 0; i < 256; ++i) { const int8_t v = (int8_t)wq[i]; wabs[i] = (uint8_t)(v < 0 ? -v : v); }   // |w| <= 128
      for (int64_t t = 0; t < batch; ++t) {
        const int8_t* aq[8];
        alignas(32) int32_t C[8];
        static const int8_t zeros[32] = {0};
        if (Q8) {   // w * a = |w| * (sign(w) a): unsigned x signed for the byte-dot instructions (|a| <= 127: no overflow)
          for (int g = 0; g < ng; ++g) {
            const int8_t* a = qqs(q8, batch, t, g0 + g);
            for (int j = 0; j < 32; ++j) { const int8_t v = (int8_t)wq[32 * g + j]; aflip[32 * g + j] = (int8_t)(v < 0 ? -a[j] : a[j]); }
            aq[g] = aflip + 32 * g;
          }
          for (int g = ng; g < 8; ++g) aq[g] = zeros;
          // (|w| = 128 would saturate vpmaddubsw's int16 pair sums only beyond 2 * 128 * 127 = 32512 < 32767: safe)
          dots(wabs, aq, C);
        } else {
          for (int g = 0; g < ng; ++g) aq[g] = qqs(q8, batch, t, g0 + g);
          for (int g = ng; g < 8; ++g) aq[g] = zeros;
          dots(wq, aq, C);
        }
        float sum = y[t * n_rows + row];
        for (int g = 0; g < ng; ++g) {
          const float d8 = ad[t * nb + g0 + g];
          if (Q8) sum += (float)C[g] * d[g] * d8;                                              // mmq.cuh:971
          else sum += d[g] * d8 * (float)(C[g] - 8 * asum[t * nb + g0 + g]);                   // :359, :391
        }
        y[t * n_rows + row] = sum;
      }
    }
  }
}

void quant_group32(const float* xi, int8_t* q, float& d_out, float& sum_out) {   // HK/ggml/mmq.cu:126-147
  float amax[32], sum[32];
  for (int l = 0; l < 32; ++l) { amax[l] = std::fabs(xi[l]); sum[l] = xi[l]; }
  for (int mask = 16; mask > 0; mask >>= 1) {
    float a2[32], s2[32];
    for (int l = 0; l < 32; ++l) { a2[l] = std::fmax(amax[l], amax[l ^ mask]); s2[l] = sum[l] + sum[l ^ mask]; }
    std::memcpy(amax, a2, sizeof(a2)); std::memcpy(sum, s2, sizeof(s2));
  }
  const float d = amax[0] / 127;
  for (int l = 0; l < 32; ++l) q[l] = (int8_t)(amax[0] == 0.0f ? 0 : std::roundf(xi[l] / d));
  d_out = d; sum_out = sum[0];
}

}  // namespace

extern "C" const char* ggq_cpu_mmq_simd_name(void) { return have_vnni() ? "avx512-vnni" : have_avx2() ? "avx2" : "scalar"; }

extern "C" int ggq_cpu_quantize_q8_1_mmq(const float* x, void* vq, int64_t batch, int64_t k, int type) {
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (batch == 0) return GGQ_OK;
  if (!x || !vq) return GGQ_ERR_ARG;
  const bool need_sum = ggq_mmq_need_sum(type) != 0;
  const int64_t padded = ggq_mmq_padded_k(k);
  uint8_t* q = (uint8_t*)vq;
  float xi[32];
  for (int64_t t = 0; t < batch; ++t)
    for (int64_t ib = 0; ib < padded / 32; ++ib) {
      for (int l = 0; l < 32; ++l) { const int64_t ix = ib * 32 + l; xi[l] = ix < k ? x[t * k + ix] : 0.0f; }
      uint8_t* blk = q + ((ib / 4) * batch + t) * 144;
      const int slot = (int)(ib % 4);
      float d, s;
      quant_group32(xi, (int8_t*)(blk + 16 + 32 * slot), d, s);
      if (need_sum) {
        const uint16_t hd = f2h(d), hs = f2h(s);
        std::memcpy(blk + 4 * slot, &hd, 2); std::memcpy(blk + 4 * slot + 2, &hs, 2);
      } else {
        std::memcpy(blk + 4 * slot, &d, 4);
      }
    }
  return GGQ_OK;
}

extern "C" int ggq_cpu_mul_mat_q(const void* w, const void* q, float* y, int type, int64_t batch, int64_t k, int64_t n_rows,
                                 int nthreads, int simd) {
  if (batch < 0 || k <= 0 || n_rows < 0) return GGQ_ERR_ARG;
  if (type != GGQ_TYPE_Q4_K && type != GGQ_TYPE_Q4_0 && type != GGQ_TYPE_Q8_0) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (batch == 0 || n_rows == 0) return GGQ_OK;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  const dots8_fn dots = pick_dots(simd);
  const uint8_t* wb = (const uint8_t*)w;
  const uint8_t* q8 = (const uint8_t*)q;
  const int64_t nb = k / 32;
  std::vector<float> ad((size_t)(batch * nb)), as((size_t)(batch * nb));
  for (int64_t t = 0; t < batch; ++t)
    for (int64_t g = 0; g < nb; ++g) {
      const uint8_t* blk = qblk(q8, batch, t, g) + 4 * (g % 4);
      if (type == GGQ_TYPE_Q8_0) { std::memcpy(&ad[(size_t)(t * nb + g)], blk, 4); }
      else { ad[(size_t)(t * nb + g)] = h2f(rd16(blk)); as[(size_t)(t * nb + g)] = h2f(rd16(blk + 2)); }
    }
  std::vector<int32_t> asum;
  if (type == GGQ_TYPE_Q4_0) {
    asum.resize((size_t)(batch * nb));
    for (int64_t t = 0; t < batch; ++t)
      for (int64_t g = 0; g < nb; ++g) {
        const int8_t* a = qqs(q8, batch, t, g);
        int s = 0;
        for (int j = 0; j < 32; ++j) s += a[j];
        asum[(size_t)(t * nb + g)] = s;
      }
  }
  auto run = [&](int64_t r0, int64_t r1) {
    if (type == GGQ_TYPE_Q4_K) rows_q4_k(wb, q8, y, batch, k, n_rows, r0, r1, dots, ad.data(), as.data());
    else if (type == GGQ_TYPE_Q8_0) rows_legacy<true>(wb, q8, y, batch, k, n_rows, r0, r1, dots, nullptr, ad.data());
    else rows_legacy<false>(wb, q8, y, batch, k, n_rows, r0, r1, dots, asum.data(), ad.data());
  };
  int nt = nthreads < 1 ? 1 : nthreads;
  if ((int64_t)nt > n_rows) nt = (int)n_rows;
  if (nt == 1) { run(0, n_rows); return GGQ_OK; }
  std::vector<std::thread> th;
  const int64_t per = (n_rows + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t r0 = t * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    if (r0 >= r1) break;
    th.emplace_back(run, r0, r1);
  }
  for (auto& x : th) x.join();
  return GGQ_OK;
}
