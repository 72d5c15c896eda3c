/*
 * ggq_oracle.c — CPU restatement of the reference algorithm (plain C).
 * TEST INFRASTRUCTURE ONLY — see ggq_oracle.h for who may call this and for the
 * pinning status of each function.  Compile with -ffp-contract=off so every
 * fp32 expression below is evaluated with exactly the roundings written.
 *
 * Reference paths are relative to the reference repo; HK/ = hf-kernels/ggml-kernels/.
 */
#include "ggq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* fp16 / bf16                                                         */
/* ------------------------------------------------------------------ */

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t oracle_f32_to_f16(float f) {
  uint32_t x = f32_bits(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
  }
  if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x38800000u) { /* below 2^-14: fp16 subnormal or zero */
    if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (2^-25 itself ties to even = 0) */
    int e = (int)(ax >> 23);                    /* biased exponent, 102..112 */
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;  /* 24-bit significand */
    int shift = 126 - e;                        /* 14..24: result = m >> shift (units of 2^-24) */
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
  }
  /* normal */
  uint32_t e = (ax >> 23) - 112u; /* fp16 biased exponent 1..30 */
  uint32_t m = ax & 0x7fffffu;
  uint32_t q = (e << 10) | (m >> 13);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++; /* may carry into exponent: correct */
  return (uint16_t)(sign | q);
}

float oracle_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    float v = (float)m * 5.9604644775390625e-08f; /* 2^-24, exact */
    return sign ? -v : v;
  }
  if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

uint16_t oracle_f32_to_bf16(float f) {
  uint32_t x = f32_bits(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7fffu + lsb;
  return (uint16_t)(x >> 16);
}
float oracle_bf16_to_f32(uint16_t h) { return bits_f32(((uint32_t)h) << 16); }

/* one IEEE fp16 operation = exact fp32 op on the widened operands + one rounding
 * (binary32 has >= 2*11+2 bits, so the double rounding is innocuous). */
typedef uint16_t h16;
static inline float H(h16 h) { return oracle_f16_to_f32(h); }
static inline h16 hmul(h16 a, h16 b) { return oracle_f32_to_f16(H(a) * H(b)); }
static inline h16 hadd(h16 a, h16 b) { return oracle_f32_to_f16(H(a) + H(b)); }
static inline h16 hsub(h16 a, h16 b) { return oracle_f32_to_f16(H(a) - H(b)); }
static inline h16 i2h(int i) { return oracle_f32_to_f16((float)i); } /* __int2half_rn */

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ------------------------------------------------------------------ */
/* Block formats: HK/ggml/ggml-common.h:17-108, ggml-cpu/ggml-common.hpp:5-42 */
/* ------------------------------------------------------------------ */

enum { T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q8_1 = 9,
       T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14,
       T_IQ2_XXS = 16, T_IQ2_XS = 17, T_IQ3_XXS = 18, T_IQ1_S = 19, T_IQ4_NL = 20, T_IQ3_S = 21, T_IQ2_S = 22,
       T_IQ4_XS = 23, T_IQ1_M = 29 /* HK/ggml/ggml-common.h:1145-1158 */ };

/* the codebook grids: constant data shared with the kernels (transcribed from HK/ggml/ggml-common.h:193-1011) */
#include "../ggml-libtorch_amd/csrc/hip/iq_tables.h"
#define IQ1_DELTA 0.125f /* IQ1S_DELTA = IQ1M_DELTA, ggml-common.h:752-753 */
/* ksigns_iq2xs[i] (ggml-common.h:1013-1022): i with bit 7 set so that the popcount is even; kmask_iq2xs[j] = 1 << j */
static inline int iq_ksigns(int i) { int p = 0; for (int b = 0; b < 7; ++b) p ^= (i >> b) & 1; return i | (p << 7); }
static inline int is_grid_iq(int type) {
  return type == T_IQ2_XXS || type == T_IQ2_XS || type == T_IQ2_S || type == T_IQ3_XXS || type == T_IQ3_S;
}
/* One 8-element run (il of sub-block ib) of a grid-codebook block: unsigned magnitudes g[8], sign bits, and the float
 * scale factor the reference multiplies half2float(d) with — dequantize.cuh:256-352 / vecdotq.cuh:607-748 */
static void iq_run(int type, const uint8_t* b, int ib, int il, uint8_t g[8], int* signs, float* mul, float* post) {
  switch (type) {
    case T_IQ2_XXS: { /* {half d; uint16 qs[32]}: per sub-block 4 index bytes + uint32 (4 x 7 sign bits, scale << 28) */
      const uint8_t* q2 = b + 2 + 8 * ib;
      const uint32_t aux32 = rd32(q2 + 4);
      const uint64_t gr = ggq_iq2xxs_grid[q2[il]];
      for (int j = 0; j < 8; ++j) g[j] = (uint8_t)(gr >> (8 * j));
      *signs = iq_ksigns((aux32 >> (7 * il)) & 127); *mul = 0.5f + (float)(aux32 >> 28); *post = 0.25f;
    } break;
    case T_IQ2_XS: { /* {half d; uint16 qs[32]; uint8 scales[8]} */
      const uint16_t q2 = rd16(b + 2 + 8 * ib + 2 * il);
      const uint64_t gr = ggq_iq2xs_grid[q2 & 511];
      for (int j = 0; j < 8; ++j) g[j] = (uint8_t)(gr >> (8 * j));
      *signs = iq_ksigns(q2 >> 9); *mul = 0.5f + (float)((b[66 + ib] >> (4 * (il / 2))) & 0xf); *post = 0.25f;
    } break;
    case T_IQ2_S: { /* {half d; uint8 qs[64]; uint8 qh[8]; uint8 scales[8]}: qs[32..63] are the sign bytes */
      const uint64_t gr = ggq_iq2s_grid[b[2 + 4 * ib + il] | ((b[66 + ib] << (8 - 2 * il)) & 0x300)];
      for (int j = 0; j < 8; ++j) g[j] = (uint8_t)(gr >> (8 * j));
      *signs = b[2 + 32 + 4 * ib + il]; *mul = 0.5f + (float)((b[74 + ib] >> (4 * (il / 2))) & 0xf); *post = 0.25f;
    } break;
    case T_IQ3_XXS: { /* {half d; uint8 qs[96]}: 64 grid indices, then 8 x uint32 (signs + scale) */
      const uint8_t* q3 = b + 2 + 8 * ib;
      const uint32_t aux32 = rd32(b + 2 + 64 + 4 * ib);
      const uint32_t g1 = ggq_iq3xxs_grid[q3[2 * il]], g2 = ggq_iq3xxs_grid[q3[2 * il + 1]];
      for (int j = 0; j < 4; ++j) { g[j] = (uint8_t)(g1 >> (8 * j)); g[4 + j] = (uint8_t)(g2 >> (8 * j)); }
      *signs = iq_ksigns((aux32 >> (7 * il)) & 127); *mul = 0.5f + (float)(aux32 >> 28); *post = 0.5f;
    } break;
    default: { /* T_IQ3_S: {half d; uint8 qs[64]; qh[8]; signs[32]; scales[4]} */
      const uint8_t* qs = b + 2 + 8 * ib;
      const int qh = b[66 + ib];
      const uint32_t g1 = ggq_iq3xs_grid[qs[2 * il] | ((qh << (8 - 2 * il)) & 256)];
      const uint32_t g2 = ggq_iq3xs_grid[qs[2 * il + 1] | ((qh << (7 - 2 * il)) & 256)];
      for (int j = 0; j < 4; ++j) { g[j] = (uint8_t)(g1 >> (8 * j)); g[4 + j] = (uint8_t)(g2 >> (8 * j)); }
      *signs = b[74 + 4 * ib + il]; *mul = 0.5f + (float)((b[106 + ib / 2] >> (4 * (ib % 2))) & 0xf); *post = 0.5f;
    } break;
  }
}
/* IQ1_S / IQ1_M run: nibble values q[8] (0..2), delta, and the sub-block scale d (dequantize.cuh:354-398) */
static float iq1m_scale(const uint8_t* b) { /* iq1m_scale_t: the fp16 super-block scale scattered over scales[] */
  const uint16_t sc0 = rd16(b + 48), sc1 = rd16(b + 50), sc2 = rd16(b + 52), sc3 = rd16(b + 54);
  return H((uint16_t)((sc0 >> 12) | ((sc1 >> 8) & 0x00f0) | ((sc2 >> 4) & 0x0f00) | (sc3 & 0xf000)));
}
static void iq1_run(int type, const uint8_t* b, int ib, int il, int q[8], float* delta, float* d) {
  uint32_t grid;
  if (type == T_IQ1_S) { /* {half d; uint8 qs[32]; uint16 qh[8]} */
    const uint16_t qh = rd16(b + 34 + 2 * ib);
    *delta = (qh & 0x8000) ? -1 - IQ1_DELTA : -1 + IQ1_DELTA;
    *d = H(rd16(b)) * (2 * ((qh >> 12) & 7) + 1);
    grid = ggq_iq1s_grid_gpu[b[2 + 4 * ib + il] | (((qh >> (3 * il)) & 7) << 8)];
  } else { /* T_IQ1_M: {uint8 qs[32]; uint8 qh[16]; uint8 scales[8]} */
    const int ib16 = 2 * ib + il / 2;
    const uint16_t sc = rd16(b + 48 + 2 * (ib16 / 4));
    const int qh = b[32 + 2 * ib + il / 2];
    *d = iq1m_scale(b) * (2 * ((sc >> (3 * (ib16 % 4))) & 0x7) + 1);
    *delta = (qh & (0x08 << (4 * (il % 2)))) ? -1 - IQ1_DELTA : -1 + IQ1_DELTA;
    grid = ggq_iq1s_grid_gpu[b[4 * ib + il] | (((qh >> (4 * (il % 2))) & 7) << 8)];
  }
  for (int j = 0; j < 4; ++j) { q[j] = (int8_t)((grid >> (8 * j)) & 0x0f); q[4 + j] = (int8_t)((grid >> (8 * j + 4)) & 0x0f); }
}

/* non-linear 4-bit codebook, HK/ggml/ggml-common.h:1060 */
static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
/* block_iq4_nl {half d; u8 qs[16]} (ggml-common.h:176-182); block_iq4_xs {half d; u16 scales_h; u8 scales_l[4];
 * u8 qs[128]} (ggml-common.h:184-191) */
#define IQ4_NL_D 0
#define IQ4_NL_QS 2
#define IQ4_XS_D 0
#define IQ4_XS_SH 2
#define IQ4_XS_SL 4
#define IQ4_XS_QS 8
static inline int iq4xs_ls(const uint8_t* b, int ib) { /* 6-bit sub-block scale, dequantize.cuh:428 / vecdotq.cuh:876 */
  const int sh = b[IQ4_XS_SH] | (b[IQ4_XS_SH + 1] << 8);
  return ((b[IQ4_XS_SL + ib / 2] >> (4 * (ib % 2))) & 0xf) | (((sh >> (2 * ib)) & 3) << 4);
}

int oracle_block_elems(int type) {
  switch (type) {
    case T_Q4_0: case T_Q4_1: case T_Q5_0: case T_Q5_1: case T_Q8_0: case T_Q8_1: case T_IQ4_NL: return 32;
    case T_Q2_K: case T_Q3_K: case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_IQ4_XS: return 256;
    case T_IQ2_XXS: case T_IQ2_XS: case T_IQ2_S: case T_IQ3_XXS: case T_IQ3_S: case T_IQ1_S: case T_IQ1_M: return 256;
    default: return 0;
  }
}
int oracle_block_bytes(int type) {
  switch (type) {
    case T_Q4_0: return 18; case T_Q4_1: return 20; case T_Q5_0: return 22; case T_Q5_1: return 24;
    case T_Q8_0: return 34; case T_Q8_1: return 36;
    case T_Q2_K: return 84; case T_Q3_K: return 110; case T_Q4_K: return 144;
    case T_Q5_K: return 176; case T_Q6_K: return 210;
    case T_IQ4_NL: return 18; case T_IQ4_XS: return 136;
    case T_IQ2_XXS: return 66; case T_IQ2_XS: return 74; case T_IQ2_S: return 82; case T_IQ3_XXS: return 98;
    case T_IQ3_S: return 110; case T_IQ1_S: return 50; case T_IQ1_M: return 56;
    default: return 0;
  }
}

/* byte offsets inside each block */
#define Q4_0_D 0
#define Q4_0_QS 2
#define Q4_1_D 0
#define Q4_1_M 2
#define Q4_1_QS 4
#define Q5_0_D 0
#define Q5_0_QH 2
#define Q5_0_QS 6
#define Q5_1_D 0
#define Q5_1_M 2
#define Q5_1_QH 4
#define Q5_1_QS 8
#define Q8_0_D 0
#define Q8_0_QS 2
#define Q8_1_D 0
#define Q8_1_S 2
#define Q8_1_QS 4
#define Q2_K_SC 0
#define Q2_K_QS 16
#define Q2_K_D 80
#define Q2_K_DMIN 82
#define Q3_K_HM 0
#define Q3_K_QS 32
#define Q3_K_SC 96
#define Q3_K_D 108
#define Q4_K_D 0
#define Q4_K_DMIN 2
#define Q4_K_SC 4
#define Q4_K_QS 16
#define Q5_K_D 0
#define Q5_K_DMIN 2
#define Q5_K_SC 4
#define Q5_K_QH 16
#define Q5_K_QS 48
#define Q6_K_QL 0
#define Q6_K_QH 128
#define Q6_K_SC 192
#define Q6_K_D 208

/* ------------------------------------------------------------------ */
/* ggml-cpu semantics: ggml-cpu/ggml-quants.hpp                        */
/* ------------------------------------------------------------------ */

int oracle_dequantize_row_f32(int type, const void* vw, float* y, int64_t k) {
  const uint8_t* w = (const uint8_t*)vw;
  const int64_t nb = k / 32;
  switch (type) {
    case T_Q4_0: /* ggml-quants.hpp:4-22 */
      for (int64_t i = 0; i < nb; i++) {
        const uint8_t* b = w + i * 18;
        const float d = H(rd16(b + Q4_0_D));
        for (int j = 0; j < 16; ++j) {
          const int x0 = (b[Q4_0_QS + j] & 0x0F) - 8;
          const int x1 = (b[Q4_0_QS + j] >> 4) - 8;
          y[i * 32 + j] = x0 * d;
          y[i * 32 + j + 16] = x1 * d;
        }
      }
      return 0;
    case T_Q4_1: /* ggml-quants.hpp:24-43 */
      for (int64_t i = 0; i < nb; i++) {
        const uint8_t* b = w + i * 20;
        const float d = H(rd16(b + Q4_1_D));
        const float m = H(rd16(b + Q4_1_M));
        for (int j = 0; j < 16; ++j) {
          const int x0 = (b[Q4_1_QS + j] & 0x0F);
          const int x1 = (b[Q4_1_QS + j] >> 4);
          y[i * 32 + j] = x0 * d + m;
          y[i * 32 + j + 16] = x1 * d + m;
        }
      }
      return 0;
    case T_Q5_0: /* ggml-quants.hpp:45-69 */
      for (int64_t i = 0; i < nb; i++) {
        const uint8_t* b = w + i * 22;
        const float d = H(rd16(b + Q5_0_D));
        const uint32_t qh = rd32(b + Q5_0_QH);
        for (int j = 0; j < 16; ++j) {
          const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
          const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
          const int32_t x0 = ((b[Q5_0_QS + j] & 0x0F) | xh_0) - 16;
          const int32_t x1 = ((b[Q5_0_QS + j] >> 4) | xh_1) - 16;
          y[i * 32 + j] = x0 * d;
          y[i * 32 + j + 16] = x1 * d;
        }
      }
      return 0;
    case T_Q5_1: /* ggml-quants.hpp:71-96 */
      for (int64_t i = 0; i < nb; i++) {
        const uint8_t* b = w + i * 24;
        const float d = H(rd16(b + Q5_1_D));
        const float m = H(rd16(b + Q5_1_M));
        const uint32_t qh = rd32(b + Q5_1_QH);
        for (int j = 0; j < 16; ++j) {
          const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
          const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
          const int x0 = (b[Q5_1_QS + j] & 0x0F) | xh_0;
          const int x1 = (b[Q5_1_QS + j] >> 4) | xh_1;
          y[i * 32 + j] = x0 * d + m;
          y[i * 32 + j + 16] = x1 * d + m;
        }
      }
      return 0;
    case T_Q8_0: /* ggml-quants.hpp:98-112 */
      for (int64_t i = 0; i < nb; i++) {
        const uint8_t* b = w + i * 34;
        const float d = H(rd16(b + Q8_0_D));
        for (int j = 0; j < 32; ++j) y[i * 32 + j] = (int8_t)b[Q8_0_QS + j] * d;
      }
      return 0;
    default:
      return -1; /* ggml-cpu/custom_ops.cpp:32-34: falls through, output undefined */
  }
}

/* ------------------------------------------------------------------ */
/* 6-bit scale unpack shared by Q4_K/Q5_K: HK/ggml/dequantize.cuh:154-161 */
/* ------------------------------------------------------------------ */
static void get_scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
  if (j < 4) {
    *d = q[j] & 63;
    *m = q[j + 4] & 63;
  } else {
    *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
    *m = (uint8_t)((q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4));
  }
}

/* Q3_K 6-bit scale, HK/ggml/dequantize.cuh:140-143 */
static int q3k_scale(const uint8_t* sc, int is) {
  int8_t us = (int8_t)(is < 4   ? (sc[is - 0] & 0xF) | (((sc[is + 8] >> 0) & 3) << 4)
                       : is < 8 ? (sc[is - 0] & 0xF) | (((sc[is + 4] >> 2) & 3) << 4)
                       : is < 12 ? (sc[is - 8] >> 4) | (((sc[is + 0] >> 4) & 3) << 4)
                                 : (sc[is - 8] >> 4) | (((sc[is - 4] >> 6) & 3) << 4));
  return us;
}

/* ------------------------------------------------------------------ */
/* GPU semantics, fp16 arithmetic: HK/ggml/dequantize.cuh             */
/* ------------------------------------------------------------------ */

int oracle_dequantize_row_f16(int type, const void* vw, uint16_t* y, int64_t k) {
  const uint8_t* w = (const uint8_t*)vw;
  const h16 h8 = oracle_f32_to_f16(8.0f), h16_ = oracle_f32_to_f16(16.0f);
  if (is_grid_iq(type)) { /* dequantize.cuh:256-352: y = __float2half(d * grid[j] * (+-1.f)), d = half2float(x.d) * (0.5f + s) * post */
    const int bs = oracle_block_bytes(type);
    for (int64_t i = 0; i < k / 256; i++)
      for (int ib = 0; ib < 8; ++ib)
        for (int il = 0; il < 4; ++il) {
          const uint8_t* b = w + i * bs;
          uint8_t g[8]; int signs; float mul, post;
          iq_run(type, b, ib, il, g, &signs, &mul, &post);
          const float d = H(rd16(b)) * mul * post;
          for (int j = 0; j < 8; ++j) y[i * 256 + 32 * ib + 8 * il + j] = oracle_f32_to_f16(d * g[j] * ((signs >> j) & 1 ? -1.f : 1.f));
        }
    return 0;
  }
  if (type == T_IQ1_S || type == T_IQ1_M) { /* dequantize.cuh:354-398: y = __float2half(d * (q[j] + delta)) */
    const int bs = oracle_block_bytes(type);
    for (int64_t i = 0; i < k / 256; i++)
      for (int ib = 0; ib < 8; ++ib)
        for (int il = 0; il < 4; ++il) {
          int q[8]; float delta, d;
          iq1_run(type, w + i * bs, ib, il, q, &delta, &d);
          for (int j = 0; j < 8; ++j) y[i * 256 + 32 * ib + 8 * il + j] = oracle_f32_to_f16(d * (q[j] + delta));
        }
    return 0;
  }
  switch (type) {
    case T_IQ4_NL: /* dequantize.cuh:399-416: fp32 product d * kvalues, one rounding to fp16 */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 18;
        const float d = H(rd16(b + IQ4_NL_D));
        for (int j = 0; j < 16; ++j) {
          y[i * 32 + j] = oracle_f32_to_f16(d * (float)kvalues_iq4nl[b[IQ4_NL_QS + j] & 0xf]);
          y[i * 32 + j + 16] = oracle_f32_to_f16(d * (float)kvalues_iq4nl[b[IQ4_NL_QS + j] >> 4]);
        }
      }
      return 0;
    case T_IQ4_XS: /* dequantize.cuh:418-433: d = half2float(x.d) * (ls - 32) in fp32, then d * kvalues */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 136;
        for (int ib = 0; ib < 8; ++ib) {
          const float d = H(rd16(b + IQ4_XS_D)) * (float)(iq4xs_ls(b, ib) - 32);
          const uint8_t* q4 = b + IQ4_XS_QS + 16 * ib;
          for (int j = 0; j < 16; ++j) {
            y[i * 256 + 32 * ib + j] = oracle_f32_to_f16(d * (float)kvalues_iq4nl[q4[j] & 0xf]);
            y[i * 256 + 32 * ib + j + 16] = oracle_f32_to_f16(d * (float)kvalues_iq4nl[q4[j] >> 4]);
          }
        }
      }
      return 0;
    case T_Q4_0: /* dequantize.cuh:3-16 + dequantize_block :80-99 */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 18;
        const h16 d = rd16(b + Q4_0_D);
        for (int iqs = 0; iqs < 16; ++iqs) {
          const int vui = b[Q4_0_QS + iqs];
          y[i * 32 + iqs] = hmul(hsub(i2h(vui & 0xF), h8), d);
          y[i * 32 + iqs + 16] = hmul(hsub(i2h(vui >> 4), h8), d);
        }
      }
      return 0;
    case T_Q4_1: /* dequantize.cuh:18-32 */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 20;
        const h16 d = rd16(b + Q4_1_D), m = rd16(b + Q4_1_M);
        for (int iqs = 0; iqs < 16; ++iqs) {
          const int vui = b[Q4_1_QS + iqs];
          y[i * 32 + iqs] = hadd(hmul(i2h(vui & 0xF), d), m);
          y[i * 32 + iqs + 16] = hadd(hmul(i2h(vui >> 4), d), m);
        }
      }
      return 0;
    case T_Q5_0: /* dequantize.cuh:34-50 */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 22;
        const h16 d = rd16(b + Q5_0_D);
        const uint32_t qh = rd32(b + Q5_0_QH);
        for (int iqs = 0; iqs < 16; ++iqs) {
          const int xh_0 = ((qh >> (iqs + 0)) << 4) & 0x10;
          const int xh_1 = ((qh >> (iqs + 12))) & 0x10;
          y[i * 32 + iqs] = hmul(hsub(i2h((b[Q5_0_QS + iqs] & 0xf) | xh_0), h16_), d);
          y[i * 32 + iqs + 16] = hmul(hsub(i2h((b[Q5_0_QS + iqs] >> 4) | xh_1), h16_), d);
        }
      }
      return 0;
    case T_Q5_1: /* dequantize.cuh:52-69 */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 24;
        const h16 d = rd16(b + Q5_1_D), m = rd16(b + Q5_1_M);
        const uint32_t qh = rd32(b + Q5_1_QH);
        for (int iqs = 0; iqs < 16; ++iqs) {
          const int xh_0 = ((qh >> (iqs + 0)) << 4) & 0x10;
          const int xh_1 = ((qh >> (iqs + 12))) & 0x10;
          y[i * 32 + iqs] = hadd(hmul(i2h((b[Q5_1_QS + iqs] & 0xf) | xh_0), d), m);
          y[i * 32 + iqs + 16] = hadd(hmul(i2h((b[Q5_1_QS + iqs] >> 4) | xh_1), d), m);
        }
      }
      return 0;
    case T_Q8_0: /* dequantize.cuh:71-78; y_offset = 1 for qr == 1 (:91) */
      for (int64_t i = 0; i < k / 32; i++) {
        const uint8_t* b = w + i * 34;
        const h16 d = rd16(b + Q8_0_D);
        for (int j = 0; j < 32; ++j) y[i * 32 + j] = hmul(i2h((int8_t)b[Q8_0_QS + j]), d);
      }
      return 0;
    case T_Q2_K: /* dequantize.cuh:101-121, 64 threads per super-block */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 84;
        const uint8_t* sc = b + Q2_K_SC;
        const h16 dall = rd16(b + Q2_K_D), dmin = rd16(b + Q2_K_DMIN);
        for (int tid = 0; tid < 64; ++tid) {
          const int n = tid / 32, l = tid - 32 * n, is = 8 * n + l / 16;
          const uint8_t q = b[Q2_K_QS + 32 * n + l];
          uint16_t* yy = y + i * 256 + 128 * n;
          for (int j = 0; j < 4; ++j) {
            const uint8_t s = sc[is + 2 * j];
            yy[l + 32 * j] = hsub(hmul(dall, i2h((s & 0xF) * ((q >> (2 * j)) & 3))), hmul(dmin, i2h(s >> 4)));
          }
        }
      }
      return 0;
    case T_Q3_K: /* dequantize.cuh:123-152 */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 110;
        const h16 d_all = rd16(b + Q3_K_D);
        for (int t = 0; t < 64; ++t) {
          const int r = t / 4, tid = r / 2, is0 = r % 2, l0 = 16 * is0 + 4 * (t % 4);
          const int n = tid / 4, j = tid - 4 * n;
          const uint8_t m = (uint8_t)(1 << (4 * n + j));
          const int is = 8 * n + 2 * j + is0, shift = 2 * j;
          const int us = q3k_scale(b + Q3_K_SC, is);
          const h16 dl = hmul(d_all, i2h(us - 32));
          uint16_t* yy = y + i * 256 + 128 * n + 32 * j;
          const uint8_t* q = b + Q3_K_QS + 32 * n;
          const uint8_t* hm = b + Q3_K_HM;
          for (int l = l0; l < l0 + 4; ++l)
            yy[l] = hmul(dl, i2h((int8_t)((q[l] >> shift) & 3) - ((hm[l] & m) ? 0 : 4)));
        }
      }
      return 0;
    case T_Q4_K: /* dequantize.cuh:163-194, 32 threads */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 144;
        const h16 dall = rd16(b + Q4_K_D), dmin = rd16(b + Q4_K_DMIN);
        for (int tid = 0; tid < 32; ++tid) {
          const int il = tid / 8, ir = tid % 8, is = 2 * il, n = 4;
          uint16_t* yy = y + i * 256 + 64 * il + n * ir;
          const uint8_t* q = b + Q4_K_QS + 32 * il + n * ir;
          uint8_t sc, m;
          get_scale_min_k4(is + 0, b + Q4_K_SC, &sc, &m);
          const h16 d1 = hmul(dall, i2h(sc)), m1 = hmul(dmin, i2h(m));
          get_scale_min_k4(is + 1, b + Q4_K_SC, &sc, &m);
          const h16 d2 = hmul(dall, i2h(sc)), m2 = hmul(dmin, i2h(m));
          for (int l = 0; l < n; ++l) {
            yy[l + 0] = hsub(hmul(d1, i2h(q[l] & 0xF)), m1);
            yy[l + 32] = hsub(hmul(d2, i2h(q[l] >> 4)), m2);
          }
        }
      }
      return 0;
    case T_Q5_K: /* dequantize.cuh:196-228, 64 threads */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 176;
        const h16 dall = rd16(b + Q5_K_D), dmin = rd16(b + Q5_K_DMIN);
        for (int tid = 0; tid < 64; ++tid) {
          const int il = tid / 16, ir = tid % 16, is = 2 * il;
          uint16_t* yy = y + i * 256 + 64 * il + 2 * ir;
          const uint8_t* ql = b + Q5_K_QS + 32 * il + 2 * ir;
          const uint8_t* qh = b + Q5_K_QH + 2 * ir;
          uint8_t sc, m;
          get_scale_min_k4(is + 0, b + Q5_K_SC, &sc, &m);
          const h16 d1 = hmul(dall, i2h(sc)), m1 = hmul(dmin, i2h(m));
          get_scale_min_k4(is + 1, b + Q5_K_SC, &sc, &m);
          const h16 d2 = hmul(dall, i2h(sc)), m2 = hmul(dmin, i2h(m));
          uint8_t hm = (uint8_t)(1 << (2 * il));
          yy[0] = hsub(hmul(d1, i2h((ql[0] & 0xF) + (qh[0] & hm ? 16 : 0))), m1);
          yy[1] = hsub(hmul(d1, i2h((ql[1] & 0xF) + (qh[1] & hm ? 16 : 0))), m1);
          hm <<= 1;
          yy[32] = hsub(hmul(d2, i2h((ql[0] >> 4) + (qh[0] & hm ? 16 : 0))), m2);
          yy[33] = hsub(hmul(d2, i2h((ql[1] >> 4) + (qh[1] & hm ? 16 : 0))), m2);
        }
      }
      return 0;
    case T_Q6_K: /* dequantize.cuh:230-254, 64 threads */
      for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* b = w + i * 210;
        const h16 d = rd16(b + Q6_K_D);
        for (int tid = 0; tid < 64; ++tid) {
          const int ip = tid / 32, il = tid - 32 * ip, is = 8 * ip + il / 16;
          uint16_t* yy = y + i * 256 + 128 * ip + il;
          const uint8_t* ql = b + Q6_K_QL + 64 * ip + il;
          const uint8_t qh = b[Q6_K_QH + 32 * ip + il];
          const int8_t* sc = (const int8_t*)(b + Q6_K_SC) + is;
          yy[0] = hmul(d, i2h(sc[0] * ((int8_t)((ql[0] & 0xF) | (((qh >> 0) & 3) << 4)) - 32)));
          yy[32] = hmul(d, i2h(sc[2] * ((int8_t)((ql[32] & 0xF) | (((qh >> 2) & 3) << 4)) - 32)));
          yy[64] = hmul(d, i2h(sc[4] * ((int8_t)((ql[0] >> 4) | (((qh >> 4) & 3) << 4)) - 32)));
          yy[96] = hmul(d, i2h(sc[6] * ((int8_t)((ql[32] >> 4) | (((qh >> 6) & 3) << 4)) - 32)));
        }
      }
      return 0;
    default:
      return -1;
  }
}

/* ------------------------------------------------------------------ */
/* Integer unpack of one block into signed "dot-product integers" + the
 * exact-arithmetic dequantisation.  qv[j] is the integer the
 * reference's dot products multiply with q8 (raw unsigned for the offset/min
 * formats: Q4_0 Q4_1 Q5_0 Q5_1 Q2_K Q4_K Q5_K; signed for Q8_0 Q3_K Q6_K).  */
/* ------------------------------------------------------------------ */
static void unpack_block(int type, const uint8_t* b, int* qv) {
  switch (type) {
    case T_Q4_0:
      for (int j = 0; j < 16; ++j) { qv[j] = b[Q4_0_QS + j] & 0xF; qv[j + 16] = b[Q4_0_QS + j] >> 4; }
      break;
    case T_Q4_1:
      for (int j = 0; j < 16; ++j) { qv[j] = b[Q4_1_QS + j] & 0xF; qv[j + 16] = b[Q4_1_QS + j] >> 4; }
      break;
    case T_Q5_0: {
      const uint32_t qh = rd32(b + Q5_0_QH);
      for (int j = 0; j < 16; ++j) {
        qv[j] = (b[Q5_0_QS + j] & 0xF) | (((qh >> j) & 1) << 4);
        qv[j + 16] = (b[Q5_0_QS + j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      }
    } break;
    case T_Q5_1: {
      const uint32_t qh = rd32(b + Q5_1_QH);
      for (int j = 0; j < 16; ++j) {
        qv[j] = (b[Q5_1_QS + j] & 0xF) | (((qh >> j) & 1) << 4);
        qv[j + 16] = (b[Q5_1_QS + j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      }
    } break;
    case T_Q8_0:
      for (int j = 0; j < 32; ++j) qv[j] = (int8_t)b[Q8_0_QS + j];
      break;
    case T_Q2_K: /* element 128n + 32j + l  <- (qs[32n+l] >> 2j) & 3 */
      for (int n = 0; n < 2; ++n) for (int j = 0; j < 4; ++j) for (int l = 0; l < 32; ++l)
        qv[128 * n + 32 * j + l] = (b[Q2_K_QS + 32 * n + l] >> (2 * j)) & 3;
      break;
    case T_Q3_K: /* low 2 bits as Q2_K; minus 4 when the hmask bit (bit 4n+j of hmask[l]) is clear */
      for (int n = 0; n < 2; ++n) for (int j = 0; j < 4; ++j) for (int l = 0; l < 32; ++l) {
        const int lo = (b[Q3_K_QS + 32 * n + l] >> (2 * j)) & 3;
        const int hb = (b[Q3_K_HM + l] >> (4 * n + j)) & 1;
        qv[128 * n + 32 * j + l] = lo - (hb ? 0 : 4);
      }
      break;
    case T_Q4_K: /* element 64il + l <- low nibble of qs[32il+l]; 64il+32+l <- high nibble */
      for (int il = 0; il < 4; ++il) for (int l = 0; l < 32; ++l) {
        qv[64 * il + l] = b[Q4_K_QS + 32 * il + l] & 0xF;
        qv[64 * il + 32 + l] = b[Q4_K_QS + 32 * il + l] >> 4;
      }
      break;
    case T_Q5_K: /* + 16 * bit (2il) / (2il+1) of qh[l] */
      for (int il = 0; il < 4; ++il) for (int l = 0; l < 32; ++l) {
        const int h = b[Q5_K_QH + l];
        qv[64 * il + l] = (b[Q5_K_QS + 32 * il + l] & 0xF) + (((h >> (2 * il)) & 1) << 4);
        qv[64 * il + 32 + l] = (b[Q5_K_QS + 32 * il + l] >> 4) + (((h >> (2 * il + 1)) & 1) << 4);
      }
      break;
    case T_Q6_K:
      for (int ip = 0; ip < 2; ++ip) for (int l = 0; l < 32; ++l) {
        const uint8_t* ql = b + Q6_K_QL + 64 * ip;
        const int h = b[Q6_K_QH + 32 * ip + l];
        qv[128 * ip + l + 0] = ((ql[l] & 0xF) | (((h >> 0) & 3) << 4)) - 32;
        qv[128 * ip + l + 32] = ((ql[l + 32] & 0xF) | (((h >> 2) & 3) << 4)) - 32;
        qv[128 * ip + l + 64] = ((ql[l] >> 4) | (((h >> 4) & 3) << 4)) - 32;
        qv[128 * ip + l + 96] = ((ql[l + 32] >> 4) | (((h >> 6) & 3) << 4)) - 32;
      }
      break;
    default: break;
  }
}

/* per-16-element scale / min integers of a K-quant super-block (sc16[16], mn16[16]) */
static void kquant_scales(int type, const uint8_t* b, int* sc16, int* mn16) {
  for (int i = 0; i < 16; ++i) { sc16[i] = 0; mn16[i] = 0; }
  switch (type) {
    case T_Q2_K:
      for (int i = 0; i < 16; ++i) { sc16[i] = b[Q2_K_SC + i] & 0xF; mn16[i] = b[Q2_K_SC + i] >> 4; }
      break;
    case T_Q3_K:
      for (int i = 0; i < 16; ++i) sc16[i] = q3k_scale(b + Q3_K_SC, i) - 32;
      break;
    case T_Q4_K: case T_Q5_K:
      for (int g = 0; g < 8; ++g) {
        uint8_t s, m;
        get_scale_min_k4(g, b + (type == T_Q4_K ? Q4_K_SC : Q5_K_SC), &s, &m);
        sc16[2 * g] = sc16[2 * g + 1] = s;
        mn16[2 * g] = mn16[2 * g + 1] = m;
      }
      break;
    case T_Q6_K:
      for (int i = 0; i < 16; ++i) sc16[i] = (int8_t)b[Q6_K_SC + i];
      break;
    default: break;
  }
}

int oracle_dequantize_row_f64(int type, const void* vw, double* y, int64_t k) {
  const uint8_t* w = (const uint8_t*)vw;
  const int qk = oracle_block_elems(type), bs = oracle_block_bytes(type);
  if (!qk || type == T_Q8_1) return -1;
  int qv[256], sc16[16], mn16[16];
  for (int64_t i = 0; i < k / qk; ++i) {
    const uint8_t* b = w + i * bs;
    if (is_grid_iq(type)) {
      for (int ib = 0; ib < 8; ++ib)
        for (int il = 0; il < 4; ++il) {
          uint8_t g[8]; int signs; float mul, post;
          iq_run(type, b, ib, il, g, &signs, &mul, &post);
          const double d = (double)H(rd16(b)) * mul * post;
          for (int j = 0; j < 8; ++j) y[i * 256 + 32 * ib + 8 * il + j] = d * g[j] * ((signs >> j) & 1 ? -1.0 : 1.0);
        }
      continue;
    }
    if (type == T_IQ1_S || type == T_IQ1_M) {
      for (int ib = 0; ib < 8; ++ib)
        for (int il = 0; il < 4; ++il) {
          int q[8]; float delta, d;
          iq1_run(type, b, ib, il, q, &delta, &d);   /* d = half * odd integer <= 15: exact in fp32 */
          for (int j = 0; j < 8; ++j) y[i * 256 + 32 * ib + 8 * il + j] = (double)d * ((double)q[j] + (double)delta);
        }
      continue;
    }
    if (type == T_IQ4_NL) {
      const double d = H(rd16(b));
      for (int j = 0; j < 16; ++j) {
        y[i * 32 + j] = d * kvalues_iq4nl[b[IQ4_NL_QS + j] & 0xf];
        y[i * 32 + j + 16] = d * kvalues_iq4nl[b[IQ4_NL_QS + j] >> 4];
      }
      continue;
    }
    if (type == T_IQ4_XS) {
      for (int ib = 0; ib < 8; ++ib) {
        const double d = (double)H(rd16(b)) * (iq4xs_ls(b, ib) - 32);
        for (int j = 0; j < 16; ++j) {
          y[i * 256 + 32 * ib + j] = d * kvalues_iq4nl[b[IQ4_XS_QS + 16 * ib + j] & 0xf];
          y[i * 256 + 32 * ib + j + 16] = d * kvalues_iq4nl[b[IQ4_XS_QS + 16 * ib + j] >> 4];
        }
      }
      continue;
    }
    unpack_block(type, b, qv);
    switch (type) {
      case T_Q4_0: { double d = H(rd16(b)); for (int j = 0; j < 32; ++j) y[i * 32 + j] = d * (qv[j] - 8); } break;
      case T_Q5_0: { double d = H(rd16(b)); for (int j = 0; j < 32; ++j) y[i * 32 + j] = d * (qv[j] - 16); } break;
      case T_Q8_0: { double d = H(rd16(b)); for (int j = 0; j < 32; ++j) y[i * 32 + j] = d * qv[j]; } break;
      case T_Q4_1: case T_Q5_1: {
        double d = H(rd16(b)), m = H(rd16(b + 2));
        for (int j = 0; j < 32; ++j) y[i * 32 + j] = d * qv[j] + m;
      } break;
      case T_Q2_K: case T_Q4_K: case T_Q5_K: {
        kquant_scales(type, b, sc16, mn16);
        const int od = type == T_Q2_K ? Q2_K_D : 0, om = type == T_Q2_K ? Q2_K_DMIN : 2;
        double d = H(rd16(b + od)), dm = H(rd16(b + om));
        for (int j = 0; j < 256; ++j) y[i * 256 + j] = d * sc16[j / 16] * qv[j] - dm * mn16[j / 16];
      } break;
      case T_Q3_K: case T_Q6_K: {
        kquant_scales(type, b, sc16, mn16);
        double d = H(rd16(b + (type == T_Q3_K ? Q3_K_D : Q6_K_D)));
        for (int j = 0; j < 256; ++j) y[i * 256 + j] = d * sc16[j / 16] * qv[j];
      } break;
      default: return -1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Activation quantisers                                               */
/* ------------------------------------------------------------------ */

/* one 32-element group: HK/ggml/ggml_kernel.cu:28-49 / mmq.cu:126-147.
 * The 32-lane xor-butterfly (masks 16,8,4,2,1) is restated as the equivalent tree. */
static void quant_group32(const float* xi, int8_t* q, float* d_out, float* sum_out) {
  float amax[32], sum[32];
  for (int l = 0; l < 32; ++l) { amax[l] = fabsf(xi[l]); sum[l] = xi[l]; }
  for (int mask = 16; mask > 0; mask >>= 1) {
    float a2[32], s2[32];
    for (int l = 0; l < 32; ++l) { a2[l] = fmaxf(amax[l], amax[l ^ mask]); s2[l] = sum[l] + sum[l ^ mask]; }
    memcpy(amax, a2, sizeof(a2)); memcpy(sum, s2, sizeof(s2));
  }
  const float d = amax[0] / 127;
  for (int l = 0; l < 32; ++l) q[l] = (int8_t)(amax[0] == 0.0f ? 0 : roundf(xi[l] / d));
  *d_out = d; *sum_out = sum[0];
}

void oracle_quantize_q8_1(const float* x, void* vq, int64_t batch, int64_t k) {
  const int64_t padded = (k + 511) / 512 * 512; /* ggml_kernel.cu:54 */
  uint8_t* q = (uint8_t*)vq;
  float xi[32];
  for (int64_t t = 0; t < batch; ++t)
    for (int64_t ib = 0; ib < padded / 32; ++ib) {
      for (int l = 0; l < 32; ++l) { int64_t ix = ib * 32 + l; xi[l] = ix < k ? x[t * k + ix] : 0.0f; }
      uint8_t* blk = q + (t * (padded / 32) + ib) * 36;
      float d, s;
      quant_group32(xi, (int8_t*)(blk + Q8_1_QS), &d, &s);
      const uint16_t hd = oracle_f32_to_f16(d), hs = oracle_f32_to_f16(s);
      blk[0] = hd & 0xff; blk[1] = hd >> 8; blk[2] = hs & 0xff; blk[3] = hs >> 8;
    }
}

void oracle_quantize_q8_1_mmq(const float* x, void* vq, int64_t batch, int64_t k, int need_sum) {
  const int64_t padded = k - k % 512 + 512; /* mmq.cu:163-164 */
  uint8_t* q = (uint8_t*)vq;
  float xi[32];
  for (int64_t t = 0; t < batch; ++t)
    for (int64_t ib = 0; ib < padded / 32; ++ib) {
      for (int l = 0; l < 32; ++l) { int64_t ix = ib * 32 + l; xi[l] = ix < k ? x[t * k + ix] : 0.0f; }
      /* block_q8_1_mmq index = (ix0/128)*kx1 + token (mmq.cu:123-124); ds slot = (ix0%128)/32 */
      uint8_t* blk = q + ((ib / 4) * batch + t) * 144;
      const int slot = (int)(ib % 4);
      float d, s;
      quant_group32(xi, (int8_t*)(blk + 16 + 32 * slot), &d, &s);
      if (need_sum) {
        const uint16_t hd = oracle_f32_to_f16(d), hs = oracle_f32_to_f16(s);
        blk[4 * slot + 0] = hd & 0xff; blk[4 * slot + 1] = hd >> 8;
        blk[4 * slot + 2] = hs & 0xff; blk[4 * slot + 3] = hs >> 8;
      } else {
        memcpy(blk + 4 * slot, &d, 4);
      }
    }
}

/* ------------------------------------------------------------------ */
/* MMVQ: HK/ggml/mmvq.cuh:2-38 + HK/ggml/vecdotq.cuh                   */
/* ------------------------------------------------------------------ */

static inline int dp4a_u(const uint8_t* a, const int8_t* b, int c) { /* unsigned bytes x signed bytes */
  return c + a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}
static inline int dp4a_s(const int8_t* a, const int8_t* b, int c) {
  return c + a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}
static inline const int8_t* q8qs(const uint8_t* q8, int blk) { return (const int8_t*)(q8 + blk * 36 + Q8_1_QS); }
static inline float q8d(const uint8_t* q8, int blk) { return H(rd16(q8 + blk * 36 + Q8_1_D)); }
static inline float q8s(const uint8_t* q8, int blk) { return H(rd16(q8 + blk * 36 + Q8_1_S)); }

/* vec_dot_<fmt>_q8_1(block, q8 blocks aligned with it, iqs) — one lane's contribution */
static float vec_dot_mmvq(int type, const uint8_t* b, const uint8_t* q8, int iqs) {
  if (is_grid_iq(type)) { /* vecdotq.cuh:607-748: iqs = 32-element sub-block */
    const int ib32 = iqs;
    const int8_t* q8v = q8qs(q8, ib32);
    int sumi[2] = {0, 0};
    float mul[2] = {0, 0}, post = 0;
    for (int l = 0; l < 4; ++l) {
      uint8_t g[8]; int signs;
      iq_run(type, b, ib32, l, g, &signs, &mul[l / 2], &post);
      for (int j = 0; j < 8; ++j) sumi[l / 2] += q8v[8 * l + j] * g[j] * ((signs >> j) & 1 ? -1 : 1);
    }
    if (type == T_IQ2_XS || type == T_IQ2_S) { /* :631-700: d * ((0.5f + ls1) * sumi1 + (0.5f + ls2) * sumi2) */
      const float d = H(rd16(b)) * q8d(q8, ib32) * post;
      return d * (mul[0] * sumi[0] + mul[1] * sumi[1]);
    }
    const float d = H(rd16(b)) * mul[0] * q8d(q8, ib32) * post; /* :607-629, 702-748 */
    return d * (sumi[0] + sumi[1]);
  }
  if (type == T_IQ1_S) { /* vecdotq.cuh:750-781 */
    const uint16_t qh = rd16(b + 34 + 2 * iqs);
    const int8_t* q8v = q8qs(q8, iqs);
    int sumi = 0;
    for (int l = 0; l < 4; ++l) {
      int q[8]; float dl, dd;
      iq1_run(type, b, iqs, l, q, &dl, &dd);
      for (int j = 0; j < 8; ++j) sumi += q8v[8 * l + j] * q[j];
    }
    const float d1q = H(rd16(b)) * (((qh >> 11) & 0x0E) + 1);
    const float delta = -1.0f + IQ1_DELTA - (qh & 0x8000) * (2.0f * IQ1_DELTA / 0x8000);
    return d1q * (q8d(q8, iqs) * sumi + q8s(q8, iqs) * delta);
  }
  if (type == T_IQ1_M) { /* vecdotq.cuh:783-826 */
    const int8_t* q8v = q8qs(q8, iqs);
    int sumi[2] = {0, 0};
    float sumf[2] = {0.0f, 0.0f};
    for (int l = 0; l < 4; ++l) {
      int q[8]; float dl, dd;
      iq1_run(type, b, iqs, l, q, &dl, &dd);
      const int qhl = b[32 + 2 * iqs + l / 2] >> (4 * (l % 2));
      const float delta = -1.0f + IQ1_DELTA - (qhl & 0x08) * (2.0f * IQ1_DELTA / 0x08);
      int sumy = 0;
      for (int j = 0; j < 8; ++j) { sumi[l / 2] += q8v[8 * l + j] * q[j]; sumy += q8v[8 * l + j]; }
      sumf[l / 2] += delta * sumy;
    }
    const float d = iq1m_scale(b) * q8d(q8, iqs);
    const int tmp = rd16(b + 48 + 2 * (iqs / 2)) >> (6 * (iqs % 2));
    const int sc0 = 2 * ((tmp >> 0) & 0x07) + 1, sc1 = 2 * ((tmp >> 3) & 0x07) + 1;
    return d * ((sumi[0] + sumf[0]) * sc0 + (sumi[1] + sumf[1]) * sc1);
  }
  switch (type) {
    case T_IQ4_NL: { /* vecdotq.cuh:842-864: codebook bytes (get_int_from_table_16, :828-840) x q8, vdr = 2 */
      int sumi1 = 0, sumi2 = 0;
      for (int l = 0; l < 2; ++l) {
        int8_t v1[4], v2[4];
        for (int c = 0; c < 4; ++c) {
          const uint8_t q = b[IQ4_NL_QS + 4 * (iqs + l) + c];
          v1[c] = kvalues_iq4nl[q & 0xf]; v2[c] = kvalues_iq4nl[q >> 4];
        }
        sumi1 = dp4a_s(v1, q8qs(q8, 0) + 4 * (iqs + l), sumi1);
        sumi2 = dp4a_s(v2, q8qs(q8, 0) + 4 * (iqs + l + 4), sumi2);
      }
      const float d = H(rd16(b + IQ4_NL_D)) * q8d(q8, 0);
      return d * (sumi1 + sumi2);
    }
    case T_IQ4_XS: { /* vecdotq.cuh:867-888: iqs = 32-element sub-block of the super-block */
      const int ib32 = iqs;
      const float d = H(rd16(b + IQ4_XS_D)) * (iq4xs_ls(b, ib32) - 32) * q8d(q8, ib32);
      int sumi1 = 0, sumi2 = 0;
      for (int j = 0; j < 4; ++j) {
        int8_t v1[4], v2[4];
        for (int c = 0; c < 4; ++c) {
          const uint8_t q = b[IQ4_XS_QS + 16 * ib32 + 4 * j + c];
          v1[c] = kvalues_iq4nl[q & 0xf]; v2[c] = kvalues_iq4nl[q >> 4];
        }
        sumi1 = dp4a_s(v1, q8qs(q8, ib32) + 4 * j, sumi1);
        sumi2 = dp4a_s(v2, q8qs(q8, ib32) + 4 * (j + 4), sumi2);
      }
      return d * (sumi1 + sumi2);
    }
    case T_Q4_0: { /* vecdotq.cuh:347-363 + :45-65, vdr = 2 */
      int sumi = 0;
      for (int i = 0; i < 2; ++i) {
        uint8_t lo[4], hi[4];
        for (int c = 0; c < 4; ++c) { uint8_t v = b[Q4_0_QS + 4 * (iqs + i) + c]; lo[c] = v & 0xF; hi[c] = v >> 4; }
        sumi = dp4a_u(lo, q8qs(q8, 0) + 4 * (iqs + i), sumi);
        sumi = dp4a_u(hi, q8qs(q8, 0) + 4 * (iqs + i + 4), sumi);
      }
      const float d4 = H(rd16(b));
      return d4 * (sumi * q8d(q8, 0) - (8 * 2 / 4) * q8s(q8, 0));
    }
    case T_Q4_1: { /* vecdotq.cuh:365-381 + :69-91 */
      int sumi = 0;
      for (int i = 0; i < 2; ++i) {
        uint8_t lo[4], hi[4];
        for (int c = 0; c < 4; ++c) { uint8_t v = b[Q4_1_QS + 4 * (iqs + i) + c]; lo[c] = v & 0xF; hi[c] = v >> 4; }
        sumi = dp4a_u(lo, q8qs(q8, 0) + 4 * (iqs + i), sumi);
        sumi = dp4a_u(hi, q8qs(q8, 0) + 4 * (iqs + i + 4), sumi);
      }
      const float d4d8 = H(hmul(rd16(b + Q4_1_D), rd16(q8 + Q8_1_D)));
      const float m4s8 = H(hmul(rd16(b + Q4_1_M), rd16(q8 + Q8_1_S)));
      return sumi * d4d8 + m4s8 / (8 / (2 * 2));
    }
    case T_Q5_0: { /* vecdotq.cuh:383-401 + :95-124 */
      int sumi = 0;
      const uint32_t qh = rd32(b + Q5_0_QH);
      for (int i = 0; i < 2; ++i) {
        uint8_t lo[4], hi[4];
        const uint32_t vh = qh >> (4 * (iqs + i));
        for (int c = 0; c < 4; ++c) {
          uint8_t v = b[Q5_0_QS + 4 * (iqs + i) + c];
          lo[c] = (uint8_t)((v & 0xF) | (((vh >> c) & 1) << 4));
          hi[c] = (uint8_t)((v >> 4) | (((vh >> (16 + c)) & 1) << 4));
        }
        sumi = dp4a_u(lo, q8qs(q8, 0) + 4 * (iqs + i), sumi);
        sumi = dp4a_u(hi, q8qs(q8, 0) + 4 * (iqs + i + 4), sumi);
      }
      const float d5 = H(rd16(b));
      return d5 * (sumi * q8d(q8, 0) - (16 * 2 / 4) * q8s(q8, 0));
    }
    case T_Q5_1: { /* vecdotq.cuh:403-421 + :128-158 */
      int sumi = 0;
      const uint32_t qh = rd32(b + Q5_1_QH);
      for (int i = 0; i < 2; ++i) {
        uint8_t lo[4], hi[4];
        const uint32_t vh = qh >> (4 * (iqs + i));
        for (int c = 0; c < 4; ++c) {
          uint8_t v = b[Q5_1_QS + 4 * (iqs + i) + c];
          lo[c] = (uint8_t)((v & 0xF) | (((vh >> c) & 1) << 4));
          hi[c] = (uint8_t)((v >> 4) | (((vh >> (16 + c)) & 1) << 4));
        }
        sumi = dp4a_u(lo, q8qs(q8, 0) + 4 * (iqs + i), sumi);
        sumi = dp4a_u(hi, q8qs(q8, 0) + 4 * (iqs + i + 4), sumi);
      }
      const float d5d8 = H(hmul(rd16(b + Q5_1_D), rd16(q8 + Q8_1_D)));
      const float m5s8 = H(hmul(rd16(b + Q5_1_M), rd16(q8 + Q8_1_S)));
      return sumi * d5d8 + m5s8 / (4 / 2);
    }
    case T_Q8_0: { /* vecdotq.cuh:423-438 + :162-174 */
      int sumi = 0;
      for (int i = 0; i < 2; ++i)
        sumi = dp4a_s((const int8_t*)(b + Q8_0_QS) + 4 * (iqs + i), q8qs(q8, 0) + 4 * (iqs + i), sumi);
      return H(rd16(b)) * q8d(q8, 0) * sumi;
    }
    case T_Q2_K: { /* vecdotq.cuh:440-462 + :195-223 */
      const int bq8_offset = 4 * (iqs / 8);
      const int scale_offset = iqs - iqs % 8 + (iqs % 8) / 4;
      const uint8_t* scales = b + Q2_K_SC + scale_offset;
      const uint8_t* v = b + Q2_K_QS + 4 * iqs;
      float sumf_d = 0.0f, sumf_m = 0.0f;
      for (int i = 0; i < 4; ++i) {
        const int sc = scales[2 * i];
        const int8_t* u = q8qs(q8, bq8_offset + i) + 4 * (iqs % 8);
        const float d8 = q8d(q8, bq8_offset + i);
        uint8_t vi[4], mm[4];
        for (int c = 0; c < 4; ++c) { vi[c] = (v[c] >> (2 * i)) & 3; mm[c] = (uint8_t)(sc >> 4); }
        sumf_d += d8 * (dp4a_u(vi, u, 0) * (sc & 0xF));
        sumf_m += d8 * dp4a_u(mm, u, 0);
      }
      return H(rd16(b + Q2_K_D)) * sumf_d - H(rd16(b + Q2_K_DMIN)) * sumf_m;
    }
    case T_Q3_K: { /* vecdotq.cuh:464-490 + :227-260 */
      const int bq8_offset = 4 * (iqs / 8);
      const int scale_offset = iqs - iqs % 8 + (iqs % 8) / 4;
      const float d = H(rd16(b + Q3_K_D));
      const uint8_t* vl = b + Q3_K_QS + 4 * iqs;
      const uint8_t* hm = b + Q3_K_HM + 4 * (iqs % 8);
      const uint8_t* scales = b + Q3_K_SC;
      float sumf = 0.0f;
      for (int i = 0; i < 4; ++i) {
        const int isc = scale_offset + 2 * i;
        const int sc_low = (scales[isc % 8] >> (4 * (isc / 8))) & 0xF;
        const int sc_high = ((scales[8 + isc % 4] >> (2 * (isc / 4))) & 3) << 4;
        const int sc = (sc_low | sc_high) - 32;
        const int8_t* u = q8qs(q8, bq8_offset + i) + 4 * (iqs % 8);
        int dot = 0;
        for (int c = 0; c < 4; ++c) {
          const int vil = (vl[c] >> (2 * i)) & 3;
          const int vih = (((~hm[c]) >> (bq8_offset + i)) & 1) << 2; /* 4 if the mask bit is clear */
          dot += (vil - vih) * u[c];
        }
        sumf += q8d(q8, bq8_offset + i) * (dot * sc);
      }
      return d * sumf;
    }
    case T_Q4_K: case T_Q5_K: { /* vecdotq.cuh:492-585 + :264-323 */
      const int is5 = type == T_Q5_K;
      const int bq8_offset = 2 * ((iqs / 2) / 4);
      const uint8_t* qs = b + (is5 ? Q5_K_QS : Q4_K_QS) + 16 * bq8_offset + 4 * ((iqs / 2) % 4);
      const uint8_t* qhp = b + Q5_K_QH + 4 * ((iqs / 2) % 4);
      uint8_t sc[2], m[2];
      get_scale_min_k4(bq8_offset + 0, b + 4, &sc[0], &m[0]); /* the aux[] shuffle of :518-526 */
      get_scale_min_k4(bq8_offset + 1, b + 4, &sc[1], &m[1]);
      float sumf_d = 0.0f, sumf_m = 0.0f;
      for (int i = 0; i < 2; ++i) {
        const int8_t* q8p = q8qs(q8, bq8_offset + i) + 4 * ((iqs / 2) % 4);
        uint8_t v0[4], v1[4], one[4] = {1, 1, 1, 1};
        for (int c = 0; c < 4; ++c) {
          v0[c] = (qs[c] >> (4 * i)) & 0xF;
          v1[c] = (qs[16 + c] >> (4 * i)) & 0xF;
          if (is5) {
            v0[c] |= (uint8_t)((((qhp[c] >> bq8_offset) >> i) & 1) << 4);
            v1[c] |= (uint8_t)((((qhp[16 + c] >> bq8_offset) >> i) & 1) << 4);
          }
        }
        const int dot1 = dp4a_u(v1, q8p + 16, dp4a_u(v0, q8p, 0));
        const int dot2 = dp4a_u(one, q8p + 16, dp4a_u(one, q8p, 0));
        const float d8 = q8d(q8, bq8_offset + i);
        sumf_d += d8 * (dot1 * sc[i]);
        sumf_m += d8 * (dot2 * m[i]);
      }
      return H(rd16(b + 0)) * sumf_d - H(rd16(b + 2)) * sumf_m;
    }
    case T_Q6_K: { /* vecdotq.cuh:587-605 + :327-345 */
      const int bq8_offset = 2 * 2 * (iqs / 16) + (iqs % 16) / 8;
      const int scale_offset = 8 * (iqs / 16) + (iqs % 16) / 4;
      const int vh_shift = 2 * ((iqs % 16) / 8);
      const uint8_t* vl = b + Q6_K_QL + 4 * iqs;
      const uint8_t* vh = b + Q6_K_QH + 4 * (8 * (iqs / 16) + iqs % 8);
      const int8_t* scales = (const int8_t*)(b + Q6_K_SC) + scale_offset;
      float sumf = 0.0f;
      for (int i = 0; i < 2; ++i) {
        const int sc = scales[4 * i];
        const int8_t* u = q8qs(q8, bq8_offset + 2 * i) + 4 * (iqs % 8);
        int dot = 0;
        for (int c = 0; c < 4; ++c) {
          const int vil = (vl[c] >> (4 * i)) & 0xF;
          const int vih = (((vh[c] >> vh_shift) >> (4 * i)) << 4) & 0x30;
          dot += ((vil | vih) - 32) * u[c];
        }
        sumf += q8d(q8, bq8_offset + 2 * i) * (dot * sc);
      }
      return H(rd16(b + Q6_K_D)) * sumf;
    }
    default: return 0.0f;
  }
}

static int mmvq_qi(int type) {
  switch (type) {
    case T_Q4_0: case T_Q4_1: case T_Q5_0: case T_Q5_1: case T_IQ4_NL: return 4;   /* QI4_NL, ggml-common.h:178 */
    case T_Q8_0: case T_IQ4_XS: return 8;                                        /* QI4_XS, ggml-common.h:185 */
    case T_IQ2_XXS: case T_IQ2_XS: case T_IQ2_S: case T_IQ3_XXS: case T_IQ3_S: case T_IQ1_S: case T_IQ1_M: return 8; /* QK_K / (4 * 8) */
    case T_Q2_K: case T_Q3_K: return 16;
    case T_Q4_K: case T_Q5_K: case T_Q6_K: return 32;
    default: return 0;
  }
}
static int mmvq_vdr(int type) {
  switch (type) {
    case T_Q2_K: case T_Q3_K: case T_Q6_K: case T_IQ4_XS: return 1;   /* mmvq.cuh:198: vdr 1 */
    case T_IQ2_XXS: case T_IQ2_XS: case T_IQ2_S: case T_IQ3_XXS: case T_IQ3_S: case T_IQ1_S: case T_IQ1_M: return 1; /* mmvq.cuh:130-209 */
    default: return 2;                                                 /* IQ4_NL: VDR_Q4_0_Q8_1_MMVQ, mmvq.cuh:189 */
  }
}

int oracle_mul_mat_vec_q(int type, const void* vw, const void* vq8, float* y, float* yabs,
                         int64_t k, int64_t n_rows) {
  const int qk = oracle_block_elems(type), bs = oracle_block_bytes(type);
  const int qi = mmvq_qi(type), vdr = mmvq_vdr(type);
  if (!qi || k % qk) return -1;
  const uint8_t* w = (const uint8_t*)vw;
  const uint8_t* q8 = (const uint8_t*)vq8;
  const int WARP = 32; /* the reference's WARP_SIZE on its CUDA target */
  const int64_t blocks_per_row = k / qk;
  const int blocks_per_warp = vdr * WARP / qi;
  for (int64_t row = 0; row < n_rows; ++row) {
    float tmp[32], ab[32];
    for (int lane = 0; lane < WARP; ++lane) {
      float t = 0.0f, a = 0.0f;
      for (int64_t i = lane / (qi / vdr); i < blocks_per_row; i += blocks_per_warp) {
        const int64_t ibx = row * blocks_per_row + i;
        const int64_t iby = i * (qk / 32);
        const int iqs = vdr * (lane % (qi / vdr));
        const float v = vec_dot_mmvq(type, w + ibx * bs, q8 + iby * 36, iqs);
        t += v; a += fabsf(v);
      }
      tmp[lane] = t; ab[lane] = a;
    }
    for (int mask = WARP / 2; mask > 0; mask >>= 1) { /* mmvq.cuh:30-33 */
      float t2[32];
      for (int l = 0; l < WARP; ++l) t2[l] = tmp[l] + tmp[l ^ mask];
      memcpy(tmp, t2, sizeof(t2));
    }
    y[row] = tmp[0];
    if (yabs) { float a = 0; for (int l = 0; l < WARP; ++l) a += ab[l]; yabs[row] = a; }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* MMQ: HK/ggml/mmq.cuh (tensor-core bodies; dp4a bodies for Q2_K/Q3_K) */
/* ------------------------------------------------------------------ */

static inline const uint8_t* mmq_blk(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) {
  return q8 + ((g32 / 4) * batch + t) * 144;
}
static inline const int8_t* mmq_qs(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) {
  return (const int8_t*)(mmq_blk(q8, batch, t, g32) + 16 + 32 * (g32 % 4));
}
static inline float mmq_d_h(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) { /* half2.x */
  return H(rd16(mmq_blk(q8, batch, t, g32) + 4 * (g32 % 4)));
}
static inline uint16_t mmq_d_hbits(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) {
  return rd16(mmq_blk(q8, batch, t, g32) + 4 * (g32 % 4));
}
static inline uint16_t mmq_s_hbits(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) {
  return rd16(mmq_blk(q8, batch, t, g32) + 4 * (g32 % 4) + 2);
}
static inline float mmq_d_f(const uint8_t* q8, int64_t batch, int64_t t, int64_t g32) { /* float d */
  float d; memcpy(&d, mmq_blk(q8, batch, t, g32) + 4 * (g32 % 4), 4); return d;
}

static int idot(const int* a, const int8_t* b, int n) {
  int s = 0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

int oracle_mul_mat_q(int type, const void* vw, const void* vq8, float* y, float* yabs,
                     int64_t batch, int64_t k, int64_t n_rows) {
  if ((type >= T_IQ2_XXS && type <= T_IQ4_XS) || type == T_IQ1_M) return -1; /* the reference's ggml_mul_mat_a8 has no IQ case (mmq.cu:222-251) */
  const int qk = oracle_block_elems(type), bs = oracle_block_bytes(type);
  if (!qk || type == T_Q8_1 || k % qk) return -1;
  const uint8_t* w = (const uint8_t*)vw;
  const uint8_t* q8 = (const uint8_t*)vq8;
  int qv[256], sc16[16], mn16[16];
  for (int64_t row = 0; row < n_rows; ++row) {
    for (int64_t t = 0; t < batch; ++t) {
      float sum = 0.0f, ab = 0.0f;
      for (int64_t ib = 0; ib < k / qk; ++ib) {
        const uint8_t* b = w + (row * (k / qk) + ib) * bs;
        unpack_block(type, b, qv);
        const int64_t g0 = ib * (qk / 32); /* first 32-element group of this block */
        float term;
        switch (type) {
          case T_Q4_0: { /* vec_dot_q4_0_q8_1_mma, mmq.cuh:330-394: A = nibble - 8 (:359) */
            int a[32]; for (int j = 0; j < 32; ++j) a[j] = qv[j] - 8;
            const int C = idot(a, mmq_qs(q8, batch, t, g0), 32);
            term = H(rd16(b)) * mmq_d_h(q8, batch, t, g0) * C; /* :391 */
            sum += term; ab += fabsf(term);
          } break;
          case T_Q4_1: case T_Q5_1: { /* mmq.cuh:466-531 / :778-844 */
            const int C = idot(qv, mmq_qs(q8, batch, t, g0), 32);
            const float lo = H(hmul(rd16(b + 0), mmq_d_hbits(q8, batch, t, g0)));
            const float hi = H(hmul(rd16(b + 2), mmq_s_hbits(q8, batch, t, g0)));
            term = lo * C + hi; /* :528 */
            sum += term; ab += fabsf(lo * C) + fabsf(hi);
          } break;
          case T_Q5_0: { /* load_tiles_q5_0 subtracts 16 (:561,:570); mma body :629-690 */
            int a[32]; for (int j = 0; j < 32; ++j) a[j] = qv[j] - 16;
            const int C = idot(a, mmq_qs(q8, batch, t, g0), 32);
            term = H(rd16(b)) * mmq_d_f(q8, batch, t, g0) * C; /* :687 */
            sum += term; ab += fabsf(term);
          } break;
          case T_Q8_0: { /* mmq.cuh:913-974 */
            const int C = idot(qv, mmq_qs(q8, batch, t, g0), 32);
            term = C * H(rd16(b)) * mmq_d_f(q8, batch, t, g0); /* :971 */
            sum += term; ab += fabsf(term);
          } break;
          case T_Q2_K: { /* vec_dot_q2_K_q8_1_mul_mat :1028-1066 + impl_mmq :18-49 */
            kquant_scales(type, b, sc16, mn16);
            const float dall = H(rd16(b + Q2_K_D)), dmin = H(rd16(b + Q2_K_DMIN));
            for (int g = 0; g < 8; ++g) {
              const int8_t* u = mmq_qs(q8, batch, t, g0 + g);
              int sumi_d = 0, sumi_m = 0;
              for (int h = 0; h < 2; ++h) {
                int su = 0;
                for (int j = 0; j < 16; ++j) su += u[16 * h + j];
                sumi_d += idot(qv + 32 * g + 16 * h, u + 16 * h, 16) * sc16[2 * g + h];
                sumi_m += su * mn16[2 * g + h];
              }
              term = mmq_d_f(q8, batch, t, g0 + g) * (dall * sumi_d - dmin * sumi_m); /* :47 */
              sum += term; ab += fabsf(term);
            }
          } break;
          case T_Q3_K: { /* vec_dot_q3_K_q8_1_mul_mat :1145-1186 + impl_mmq :51-72 */
            kquant_scales(type, b, sc16, mn16);
            const float d3 = H(rd16(b + Q3_K_D));
            for (int g = 0; g < 8; ++g) {
              const int8_t* u = mmq_qs(q8, batch, t, g0 + g);
              int sumi = 0;
              for (int h = 0; h < 2; ++h) sumi += idot(qv + 32 * g + 16 * h, u + 16 * h, 16) * sc16[2 * g + h];
              term = d3 * mmq_d_f(q8, batch, t, g0 + g) * sumi; /* :70 */
              sum += term; ab += fabsf(term);
            }
          } break;
          case T_Q4_K: case T_Q5_K: { /* vec_dot_q4_K_q8_1_mma :1274-1363 / q5_K :1463-1553 */
            kquant_scales(type, b, sc16, mn16);
            const float dall = H(rd16(b + 0)), dmin = H(rd16(b + 2));
            for (int p = 0; p < 4; ++p) { /* one k0 step = 64 elements = two q8 groups */
              float tmpd = 0.0f, tmpm = 0.0f;
              for (int h = 0; h < 2; ++h) {
                const int g = 2 * p + h;
                const int C = idot(qv + 32 * g, mmq_qs(q8, batch, t, g0 + g), 32);
                tmpd += (C * sc16[2 * g]) * mmq_d_h(q8, batch, t, g0 + g);           /* :1352 */
                tmpm += mn16[2 * g] * H(mmq_s_hbits(q8, batch, t, g0 + g));           /* :1353 */
              }
              term = dall * tmpd - dmin * tmpm; /* :1359 */
              sum += term; ab += fabsf(dall * tmpd) + fabsf(dmin * tmpm);
            }
          } break;
          case T_Q6_K: { /* vec_dot_q6_K_q8_1_mma :1657-1737 */
            kquant_scales(type, b, sc16, mn16);
            const float d6 = H(rd16(b + Q6_K_D));
            for (int p = 0; p < 4; ++p) {
              float tmp = 0.0f;
              for (int h = 0; h < 2; ++h) {
                const int g = 2 * p + h;
                const int8_t* u = mmq_qs(q8, batch, t, g0 + g);
                const int C0 = idot(qv + 32 * g, u, 16), C1 = idot(qv + 32 * g + 16, u + 16, 16);
                tmp += (C0 * sc16[2 * g] + C1 * sc16[2 * g + 1]) * mmq_d_f(q8, batch, t, g0 + g); /* :1726 */
              }
              term = tmp * d6; /* :1732 */
              sum += term; ab += fabsf(term);
            }
          } break;
          default: return -1;
        }
      }
      y[t * n_rows + row] = sum;
      if (yabs) yabs[t * n_rows + row] = ab;
    }
  }
  return 0;
}
