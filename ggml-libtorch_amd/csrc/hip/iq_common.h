// iq_common.h — device helpers of the grid-codebook IQ formats (IQ2_XXS / IQ2_XS / IQ2_S / IQ3_XXS / IQ3_S / IQ1_S / IQ1_M):
// block layouts after HK/ggml/ggml-common.h:108-176, decode rules after HK/ggml/dequantize.cuh:256-398, 471-512 and
// HK/ggml/vecdotq.cuh:607-826.  The grids are constant data (iq_tables.h); everything else here is ours.
#pragma once
#include "ggq_common.h"
#define GGQ_IQ_CONST static __device__ const
#include "iq_tables.h"

namespace ggq {

namespace off {
constexpr int IQ2_XXS_D = 0, IQ2_XXS_QS = 2;                                         // {half d; uint16 qs[32]}             66 B
constexpr int IQ2_XS_D = 0, IQ2_XS_QS = 2, IQ2_XS_SC = 66;                            // {half d; uint16 qs[32]; u8 scales[8]} 74 B
constexpr int IQ2_S_D = 0, IQ2_S_QS = 2, IQ2_S_SIGNS = 34, IQ2_S_QH = 66, IQ2_S_SC = 74;   // {d; qs[64] (32 idx + 32 signs); qh[8]; scales[8]} 82 B
constexpr int IQ3_XXS_D = 0, IQ3_XXS_QS = 2, IQ3_XXS_GAS = 66;                        // {d; qs[96]: 64 grid indices + 8 x uint32 scale/signs} 98 B
constexpr int IQ3_S_D = 0, IQ3_S_QS = 2, IQ3_S_QH = 66, IQ3_S_SIGNS = 74, IQ3_S_SC = 106;  // {d; qs[64]; qh[8]; signs[32]; scales[4]} 110 B
constexpr int IQ1_S_D = 0, IQ1_S_QS = 2, IQ1_S_QH = 34;                               // {d; qs[32]; uint16 qh[8]} 50 B
constexpr int IQ1_M_QS = 0, IQ1_M_QH = 32, IQ1_M_SC = 48;                             // {qs[32]; qh[16]; scales[8]} 56 B
}  // namespace off

constexpr float IQ1_DELTA = 0.125f;   // IQ1S_DELTA = IQ1M_DELTA, ggml-common.h:752-753

// ksigns_iq2xs[i] (ggml-common.h:1013-1022) = i with bit 7 set so that the popcount is even: 8 sign bits of 8 elements
__device__ __forceinline__ uint32_t iq_signs8(uint32_t idx7) { return idx7 | ((uint32_t)(__builtin_popcount(idx7) & 1) << 7); }
// sign bits k = 0..3 of `bits` -> 0xFF in byte k (the job of __vcmpeq4(((s & 0xf) * 0x01010101) & 0x08040201, 0x08040201))
__device__ __forceinline__ uint32_t iq_sign_mask4(uint32_t bits) { return (((bits & 0xF) * 0x00204081u) & 0x01010101u) * 0xFFu; }
// per byte: m == 0xFF ? -g : g, for grid magnitudes g in 1..0x7F (never 0: g ^ 0xFF + 1 cannot carry into the next byte)
__device__ __forceinline__ uint32_t iq_apply_signs4(uint32_t g4, uint32_t m4) { return (g4 ^ m4) + (m4 & 0x01010101u); }

// which codebook a format reads, for kernels that stage it in LDS (the GEMV: a per-lane lookup in global memory touches up to
// 64 cache lines per wave-level load; in LDS it is one ds_read)
template <int T> struct IqGrid { static constexpr int BYTES = 0; static __device__ __forceinline__ const void* table() { return nullptr; } };
#define GGQ_IQ_GRID(T, TAB) \
  template <> struct IqGrid<T> { static constexpr int BYTES = (int)sizeof(TAB); static __device__ __forceinline__ const void* table() { return TAB; } }
GGQ_IQ_GRID(GGQ_TYPE_IQ2_XXS, ggq_iq2xxs_grid);
GGQ_IQ_GRID(GGQ_TYPE_IQ2_XS, ggq_iq2xs_grid);
GGQ_IQ_GRID(GGQ_TYPE_IQ2_S, ggq_iq2s_grid);
GGQ_IQ_GRID(GGQ_TYPE_IQ3_XXS, ggq_iq3xxs_grid);
GGQ_IQ_GRID(GGQ_TYPE_IQ3_S, ggq_iq3xs_grid);
GGQ_IQ_GRID(GGQ_TYPE_IQ1_S, ggq_iq1s_grid_gpu);
GGQ_IQ_GRID(GGQ_TYPE_IQ1_M, ggq_iq1s_grid_gpu);
#undef GGQ_IQ_GRID

// the eight signed int8 values of 8-element run `il` of 32-element sub-block `ib` as two dwords, and the sub-block's
// float scale factor (the reference's `d` without x.d), per format.  `grid` = the format's codebook (IqGrid<T>::table() or a
// copy of it in LDS).
template <int T> struct IqRun;

template <> struct IqRun<GGQ_TYPE_IQ2_XXS> {
  static __device__ __forceinline__ void get(const void* grid, const uint8_t* b, int ib, int il, uint32_t& lo, uint32_t& hi, float& mul) {
    const u32x2_a2 q = ld_u32x2(b + off::IQ2_XXS_QS + 8 * ib);   // {4 grid indices, scale << 28 | 4 x 7 sign bits}
    const uint64_t g = ((const uint64_t*)grid)[(q.v[0] >> (8 * il)) & 0xFF];
    const uint32_t s = iq_signs8((q.v[1] >> (7 * il)) & 127);
    lo = iq_apply_signs4((uint32_t)g, iq_sign_mask4(s));
    hi = iq_apply_signs4((uint32_t)(g >> 32), iq_sign_mask4(s >> 4));
    mul = 0.5f + (float)(q.v[1] >> 28);
  }
  static constexpr float post = 0.25f;
};
template <> struct IqRun<GGQ_TYPE_IQ2_XS> {
  static __device__ __forceinline__ void get(const void* grid, const uint8_t* b, int ib, int il, uint32_t& lo, uint32_t& hi, float& mul) {
    const uint32_t q2 = ld_u16(b + off::IQ2_XS_QS + 8 * ib + 2 * il);
    const uint64_t g = ((const uint64_t*)grid)[q2 & 511];
    const uint32_t s = iq_signs8(q2 >> 9);
    lo = iq_apply_signs4((uint32_t)g, iq_sign_mask4(s));
    hi = iq_apply_signs4((uint32_t)(g >> 32), iq_sign_mask4(s >> 4));
    mul = 0.5f + (float)((b[off::IQ2_XS_SC + ib] >> (4 * (il >> 1))) & 0xF);
  }
  static constexpr float post = 0.25f;
};
template <> struct IqRun<GGQ_TYPE_IQ2_S> {
  static __device__ __forceinline__ void get(const void* grid, const uint8_t* b, int ib, int il, uint32_t& lo, uint32_t& hi, float& mul) {
    const uint32_t idx = b[off::IQ2_S_QS + 4 * ib + il] | (((uint32_t)b[off::IQ2_S_QH + ib] << (8 - 2 * il)) & 0x300);
    const uint64_t g = ((const uint64_t*)grid)[idx];
    const uint32_t s = b[off::IQ2_S_SIGNS + 4 * ib + il];
    lo = iq_apply_signs4((uint32_t)g, iq_sign_mask4(s));
    hi = iq_apply_signs4((uint32_t)(g >> 32), iq_sign_mask4(s >> 4));
    mul = 0.5f + (float)((b[off::IQ2_S_SC + ib] >> (4 * (il >> 1))) & 0xF);
  }
  static constexpr float post = 0.25f;
};
template <> struct IqRun<GGQ_TYPE_IQ3_XXS> {
  static __device__ __forceinline__ void get(const void* grid, const uint8_t* b, int ib, int il, uint32_t& lo, uint32_t& hi, float& mul) {
    const uint32_t q3 = ld_u16(b + off::IQ3_XXS_QS + 8 * ib + 2 * il);
    const uint32_t aux = ld_u32(b + off::IQ3_XXS_GAS + 4 * ib);
    const uint32_t s = iq_signs8((aux >> (7 * il)) & 127);
    lo = iq_apply_signs4(((const uint32_t*)grid)[q3 & 0xFF], iq_sign_mask4(s));
    hi = iq_apply_signs4(((const uint32_t*)grid)[q3 >> 8], iq_sign_mask4(s >> 4));
    mul = 0.5f + (float)(aux >> 28);
  }
  static constexpr float post = 0.5f;
};
template <> struct IqRun<GGQ_TYPE_IQ3_S> {
  static __device__ __forceinline__ void get(const void* grid, const uint8_t* b, int ib, int il, uint32_t& lo, uint32_t& hi, float& mul) {
    const uint32_t q3 = ld_u16(b + off::IQ3_S_QS + 8 * ib + 2 * il), qh = b[off::IQ3_S_QH + ib];
    const uint32_t s = b[off::IQ3_S_SIGNS + 4 * ib + il];
    lo = iq_apply_signs4(((const uint32_t*)grid)[(q3 & 0xFF) | ((qh << (8 - 2 * il)) & 256)], iq_sign_mask4(s));
    hi = iq_apply_signs4(((const uint32_t*)grid)[(q3 >> 8) | ((qh << (7 - 2 * il)) & 256)], iq_sign_mask4(s >> 4));
    mul = 0.5f + (float)((b[off::IQ3_S_SC + (ib >> 1)] >> (4 * (ib & 1))) & 0xF);
  }
  static constexpr float post = 0.5f;
};

// IQ1_S / IQ1_M: eight nibble values 0..2 (two dwords of bytes), the run's delta and integer scale
__device__ __forceinline__ void iq1_grid(const void* grid, uint32_t idx11, uint32_t& lo, uint32_t& hi) {
  const uint32_t g = ((const uint32_t*)grid)[idx11];
  lo = g & 0x0F0F0F0Fu;
  hi = (g >> 4) & 0x0F0F0F0Fu;
}
__device__ __forceinline__ float iq1m_super_scale(const uint8_t* b) {   // iq1m_scale_t, dequantize.cuh:481-482
  const u32x2_a2 s = ld_u32x2(b + off::IQ1_M_SC);
  const uint32_t sc0 = s.v[0] & 0xFFFF, sc1 = s.v[0] >> 16, sc2 = s.v[1] & 0xFFFF, sc3 = s.v[1] >> 16;
  return bits_h_f32((sc0 >> 12) | ((sc1 >> 8) & 0x00F0) | ((sc2 >> 4) & 0x0F00) | (sc3 & 0xF000));
}

}  // namespace ggq
