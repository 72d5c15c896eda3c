// mmq_unpack.h — integer operands and float scales of one 32-element weight group per format (the "MMQ canon",
// SURVEY §8a / DESIGN §2): shared by every quantised-GEMM kernel (mmq.hip, mmq_sk.hip) so that all of them contract
// exactly the same int8 values with the same scales.  Restates the tile loaders of HK/ggml/mmq.cuh:257-1618.
#pragma once
#include "ggq_common.h"

namespace ggq {

constexpr int WROW = 272;  // LDS pitch of one unpacked int8 weight row (256 + 16)
constexpr uint32_t MAGIC_I = 0x4B400000u;   // bits of 12582912.0f = 1.5 * 2^23
constexpr float MAGIC_F = 12582912.0f;

template <int T> struct MmqTraits {
  static constexpr bool need_sum = T == GGQ_TYPE_Q4_0 || T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_1 ||
                                   T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K;  // mmq.cu:84-106
  static constexpr bool fp16_prod = T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_1;  // __hmul2(dm, ds8)
  static constexpr bool mfma_min = T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K;   // min term on the matrix pipe
  static constexpr bool two_tiles = T == GGQ_TYPE_Q2_K;  // second int8 tile carries the mins
  static constexpr bool half_scales = T == GGQ_TYPE_Q6_K;  // scale per 16 elements -> K=16 MFMAs
  // float scale arrays per (group,row): 1 = sA; 2 = sA + (mA | sA1 | dmin)
  static constexpr int n_scale = (fp16_prod || mfma_min || two_tiles || half_scales) ? 2 : 1;
  // one int32 result tile per group: fits 128 VGPRs, so two workgroups share a CU (4 waves/SIMD)
  static constexpr bool light = !(fp16_prod || two_tiles || half_scales);
};

// int32 bits -> float, BY VALUE: __builtin_bit_cast applied directly to an ext-vector element
// (bit_cast(float, v[i])) is miscompiled by this clang — it reads element 0 for every i.
__device__ __forceinline__ float as_f32(int v) { return __builtin_bit_cast(float, v); }

// (x - c) per byte for x in [0, 2c): exact, no inter-byte borrow
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t c4) {
  return ((x | 0x80808080u) - c4) ^ 0x80808080u;
}
// bit k (k = 0..3) of x -> bit 4 of byte k: the four partial products of the multiply do not overlap
__device__ __forceinline__ uint32_t spread4b(uint32_t x) {
  return (((x & 0xF) * 0x00204081u) & 0x01010101u) << 4;
}

// Raw bytes of one 32-element weight group as loaded from global memory (prefetch registers);
// only the members a format touches survive dead-code elimination.
struct Raw {
  u32x4_a2 q[5];
  uint32_t s[4];
};

template <int T>
__device__ __forceinline__ void load_raw(const uint8_t* row, int G, Raw& r) {
  if constexpr (T == GGQ_TYPE_Q4_0) {
    const uint8_t* b = row + (int64_t)G * 18;
    r.q[0] = ld_u32x4(b + off::Q4_0_QS); r.s[0] = ld_u16(b);
  } else if constexpr (T == GGQ_TYPE_Q4_1) {
    const uint8_t* b = row + (int64_t)G * 20;
    r.q[0] = ld_u32x4(b + off::Q4_1_QS); r.s[0] = ld_u32(b);
  } else if constexpr (T == GGQ_TYPE_Q5_0) {
    const uint8_t* b = row + (int64_t)G * 22;
    r.q[0] = ld_u32x4(b + off::Q5_0_QS); r.s[0] = ld_u16(b); r.s[1] = ld_u32(b + off::Q5_0_QH);
  } else if constexpr (T == GGQ_TYPE_Q5_1) {
    const uint8_t* b = row + (int64_t)G * 24;
    r.q[0] = ld_u32x4(b + off::Q5_1_QS); r.s[0] = ld_u32(b); r.s[1] = ld_u32(b + off::Q5_1_QH);
  } else if constexpr (T == GGQ_TYPE_Q8_0) {
    const uint8_t* b = row + (int64_t)G * 34;
    r.q[0] = ld_u32x4(b + off::Q8_0_QS); r.q[1] = ld_u32x4(b + off::Q8_0_QS + 16); r.s[0] = ld_u16(b);
  } else if constexpr (T == GGQ_TYPE_Q2_K) {
    const int ib = G >> 3, gl = G & 7, n = gl >> 2;
    const uint8_t* b = row + (int64_t)ib * 84;
    r.q[0] = ld_u32x4(b + off::Q2_K_QS + 32 * n); r.q[1] = ld_u32x4(b + off::Q2_K_QS + 32 * n + 16);
    r.s[0] = ld_u16(b + off::Q2_K_SC + 2 * gl); r.s[1] = ld_u32(b + off::Q2_K_D);
  } else if constexpr (T == GGQ_TYPE_Q3_K) {
    const int ib = G >> 3, gl = G & 7, n = gl >> 2;
    const uint8_t* b = row + (int64_t)ib * 110;
    r.q[0] = ld_u32x4(b + off::Q3_K_QS + 32 * n); r.q[1] = ld_u32x4(b + off::Q3_K_QS + 32 * n + 16);
    r.q[2] = ld_u32x4(b + off::Q3_K_HM); r.q[3] = ld_u32x4(b + off::Q3_K_HM + 16);
    const u32x3_a2 s = ld_u32x3(b + off::Q3_K_SC);
    r.s[0] = s.v[0]; r.s[1] = s.v[1]; r.s[2] = s.v[2]; r.s[3] = ld_u16(b + off::Q3_K_D);
  } else if constexpr (T == GGQ_TYPE_Q4_K) {
    const int ib = G >> 3, il = (G & 7) >> 1;
    const uint8_t* b = row + (int64_t)ib * 144;
    r.q[0] = ld_u32x4(b + off::Q4_K_QS + 32 * il); r.q[1] = ld_u32x4(b + off::Q4_K_QS + 32 * il + 16);
    r.q[2] = ld_u32x4(b);
  } else if constexpr (T == GGQ_TYPE_Q5_K) {
    const int ib = G >> 3, il = (G & 7) >> 1;
    const uint8_t* b = row + (int64_t)ib * 176;
    r.q[0] = ld_u32x4(b + off::Q5_K_QS + 32 * il); r.q[1] = ld_u32x4(b + off::Q5_K_QS + 32 * il + 16);
    r.q[2] = ld_u32x4(b); r.q[3] = ld_u32x4(b + off::Q5_K_QH); r.q[4] = ld_u32x4(b + off::Q5_K_QH + 16);
  } else if constexpr (T == GGQ_TYPE_Q6_K) {
    const int ib = G >> 3, gl = G & 7, ip = gl >> 2, j = gl & 3;
    const uint8_t* b = row + (int64_t)ib * 210;
    const uint8_t* pl = b + off::Q6_K_QL + 64 * ip + 32 * (j & 1);
    const uint8_t* ph = b + off::Q6_K_QH + 32 * ip;
    r.q[0] = ld_u32x4(pl); r.q[1] = ld_u32x4(pl + 16); r.q[2] = ld_u32x4(ph); r.q[3] = ld_u32x4(ph + 16);
    r.s[0] = ld_u16(b + off::Q6_K_D); r.s[1] = ld_u16(b + off::Q6_K_SC + 2 * gl);
  }
}

// Raw -> 8 dwords of signed int8 (w[]), optional second tile (w2[], Q2_K mins), float scales.
template <int T>
__device__ __forceinline__ void unpack_raw(const Raw& r, int G, uint32_t w[8], uint32_t w2[8], float& s0, float& s1) {
  s0 = 0.0f; s1 = 0.0f;
  const int gl = G & 7;
  if constexpr (T == GGQ_TYPE_Q4_0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes(r.q[0].v[i] & 0x0F0F0F0F, 0x08080808u);  // mmq.cuh:359
      w[4 + i] = sub_bytes((r.q[0].v[i] >> 4) & 0x0F0F0F0F, 0x08080808u);
    }
    s0 = bits_h_f32(r.s[0]);
  } else if constexpr (T == GGQ_TYPE_Q4_1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = r.q[0].v[i] & 0x0F0F0F0F; w[4 + i] = (r.q[0].v[i] >> 4) & 0x0F0F0F0F; }
    s0 = bits_h_f32(r.s[0] & 0xFFFF); s1 = bits_h_f32(r.s[0] >> 16);
  } else if constexpr (T == GGQ_TYPE_Q5_0) {
    const uint32_t qh = r.s[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes((r.q[0].v[i] & 0x0F0F0F0F) | spread4b(qh >> (4 * i)), 0x10101010u);  // mmq.cuh:561
      w[4 + i] = sub_bytes(((r.q[0].v[i] >> 4) & 0x0F0F0F0F) | spread4b(qh >> (16 + 4 * i)), 0x10101010u);
    }
    s0 = bits_h_f32(r.s[0]);
  } else if constexpr (T == GGQ_TYPE_Q5_1) {
    const uint32_t qh = r.s[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = (r.q[0].v[i] & 0x0F0F0F0F) | spread4b(qh >> (4 * i));
      w[4 + i] = ((r.q[0].v[i] >> 4) & 0x0F0F0F0F) | spread4b(qh >> (16 + 4 * i));
    }
    s0 = bits_h_f32(r.s[0] & 0xFFFF); s1 = bits_h_f32(r.s[0] >> 16);
  } else if constexpr (T == GGQ_TYPE_Q8_0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = r.q[0].v[i]; w[4 + i] = r.q[1].v[i]; }
    s0 = bits_h_f32(r.s[0]);
  } else if constexpr (T == GGQ_TYPE_Q2_K) {
    const int j = gl & 3;
    const int sc0 = r.s[0] & 0xFF, sc1 = (r.s[0] >> 8) & 0xFF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // bytes <= 3 * 15: the dword multiply cannot carry between bytes
      w[i] = ((r.q[0].v[i] >> (2 * j)) & 0x03030303u) * (uint32_t)(sc0 & 0xF);
      w[4 + i] = ((r.q[1].v[i] >> (2 * j)) & 0x03030303u) * (uint32_t)(sc1 & 0xF);
      w2[i] = 0x01010101u * (uint32_t)(sc0 >> 4);
      w2[4 + i] = 0x01010101u * (uint32_t)(sc1 >> 4);
    }
    s0 = bits_h_f32(r.s[1] & 0xFFFF); s1 = bits_h_f32(r.s[1] >> 16);
  } else if constexpr (T == GGQ_TYPE_Q3_K) {
    const int j = gl & 3;
    const int sc0 = q3k_scale(r.s[0], r.s[1], r.s[2], 2 * gl), sc1 = q3k_scale(r.s[0], r.s[1], r.s[2], 2 * gl + 1);
    // tile holds -(q3 * sc) in [-128, 124] (q3*sc itself reaches +128); the sign goes into s0.  Four bytes at a
    // time: b = q2 + 4·h in 0..7 (q3 = b - 4), u = b·|sc| (one packed 16-bit multiply: b1·|sc|·256 + b0·|sc| < 2^16),
    // c = 4·|sc|, and -(q3·sc) = (4 - b)·sc = c - u for sc >= 0, u - c for sc < 0, as a per-byte subtraction.
    auto tile4 = [&](uint32_t q, uint32_t hm, int sc) {
      const uint32_t b4 = ((q >> (2 * j)) & 0x03030303u) + (((hm >> gl) & 0x01010101u) << 2);
      const uint32_t a = (uint32_t)(sc < 0 ? -sc : sc);
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      const us2 prod = __builtin_bit_cast(us2, b4) * us2{(unsigned short)a, (unsigned short)a};
      const uint32_t u = __builtin_bit_cast(uint32_t, prod), c = 0x04040404u * a;
      const uint32_t x = sc < 0 ? u : c, y = sc < 0 ? c : u;
      constexpr uint32_t H = 0x80808080u;
      return ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);   // x - y per byte, no borrow between bytes
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = tile4(r.q[0].v[i], r.q[2].v[i], sc0);
      w[4 + i] = tile4(r.q[1].v[i], r.q[3].v[i], sc1);
    }
    s0 = -bits_h_f32(r.s[3]);
  } else if constexpr (T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K) {
    const int nib = gl & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = (r.q[0].v[i] >> (4 * nib)) & 0x0F0F0F0F;
      w[4 + i] = (r.q[1].v[i] >> (4 * nib)) & 0x0F0F0F0F;
      if constexpr (T == GGQ_TYPE_Q5_K) {
        w[i] |= ((r.q[3].v[i] >> gl) & 0x01010101u) << 4;
        w[4 + i] |= ((r.q[4].v[i] >> gl) & 0x01010101u) << 4;
      }
    }
    int sc, mn;
    k4_scale_min(r.q[2].v[1], r.q[2].v[2], r.q[2].v[3], gl, sc, mn);
    s0 = bits_h_f32(r.q[2].v[0] & 0xFFFF) * (float)sc;
    s1 = -(bits_h_f32(r.q[2].v[0] >> 16) * (float)mn);
  } else if constexpr (T == GGQ_TYPE_Q6_K) {
    const int j = gl & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes(((r.q[0].v[i] >> (4 * (j >> 1))) & 0x0F0F0F0F) | (((r.q[2].v[i] >> (2 * j)) & 0x03030303u) << 4), 0x20202020u);
      w[4 + i] = sub_bytes(((r.q[1].v[i] >> (4 * (j >> 1))) & 0x0F0F0F0F) | (((r.q[3].v[i] >> (2 * j)) & 0x03030303u) << 4), 0x20202020u);
    }
    const float d = bits_h_f32(r.s[0]);
    s0 = d * (float)(int8_t)(r.s[1] & 0xFF);
    s1 = d * (float)(int8_t)((r.s[1] >> 8) & 0xFF);
  }
}

}  // namespace ggq
