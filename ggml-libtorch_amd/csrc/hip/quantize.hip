// quantize.hip — activations (fp32/fp16/bf16) -> Q8_1 int8 blocks, gfx950.
//
// Replaces quantize_q8_1 / quantize_row_q8_1_cuda (HK/ggml/ggml_kernel.cu:13-66) and
// quantize_mmq_q8_1 / quantize_mmq_q8_1_cuda (HK/ggml/mmq.cu:109-177).
//
// Bit-exactness contract: per 32 elements d = amax/127 (IEEE fp32 divide),
// q = roundf(x/d) (0 when amax == 0), sum = the reference's 32-lane xor-butterfly
// (masks 16,8,4,2,1) in fp32 — reproduced here as the same pairing tree.
//
// Mapping: one lane owns 4 consecutive elements (one 16/8-byte load, one dword of
// int8 out), 8 lanes form a 32-element group, a wave covers 256 elements.  Butterfly
// levels 16/8/4 are lane exchanges (xor 4/2/1 in lane units), levels 2/1 are in-lane.
#include "ggq_common.h"
#include <type_traits>

namespace ggq {

template <int DT>
__device__ __forceinline__ void load4(const void* x, int64_t base, int64_t ix, int64_t k, float v[4]) {
  // elements ix..ix+3 of a row of k (zero beyond k: the reference pads with zeros).  ONE 8- / 16-byte load where the four elements are
  // inside the row together and aligned (k % 4 == 0 and x at a 4-element boundary: a kernel-uniform test); four predicated 2-byte loads,
  // each in its own exec-masked block, were what the kernel spent its time on (4096 x 4096 tokens: 26.6 us per launch).
  constexpr int ES = DT == GGQ_F32 ? 4 : 2;
  if ((k & 3) == 0 && ((uintptr_t)x & (4 * ES - 1)) == 0) {
    if (ix >= k) {
      v[0] = v[1] = v[2] = v[3] = 0.0f;
    } else if constexpr (DT == GGQ_F32) {
      const float4 f = *(const float4*)((const float*)x + base + ix);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
      const uint2 u = *(const uint2*)((const uint16_t*)x + base + ix);
      const uint16_t h[4] = {(uint16_t)(u.x & 0xFFFF), (uint16_t)(u.x >> 16), (uint16_t)(u.y & 0xFFFF), (uint16_t)(u.y >> 16)};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = Elem<DT>::ld(h, i);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (ix + i < k) ? Elem<DT>::ld(x, base + ix + i) : 0.0f;
}

// LAYOUT 0: block_q8_1 {half d; half s; int8 qs[32]}, row-major [batch][padded/32]
// LAYOUT 1: block_q8_1_mmq {half2 ds[4]; int8 qs[128]}, index (ix/128)*batch + token
// LAYOUT 2: the same 144 bytes per (128 elements, token), regrouped so that one MFMA B fragment
//           (32 tokens x 32 elements) is 1 KB contiguous in lane order: per (ix/128, token/32) a
//           4608-byte tile { int8 qs[4 groups][2 halves][32 tokens][16]; ds[2 pairs][32 tokens][2] }
// LAYOUT 3: the same values for the 16-token-tile kernel (mmq_t16.hip): per (ix/256, token/16) a 4608-byte tile
//           { int8 frag[4][4 K-chunks][16 tokens][16]; ds[2 halves][4 token quads][4 groups][4 tokens] } — one MFMA
//           16x16x64 operand fragment is 1 KB contiguous in lane order (lane = token + 16 * K-chunk).  Which sixteen
//           elements of the 256 a (fragment, K-chunk) slot holds depends on the WEIGHT format's bit layout (the slot's
//           partner is the 16 bytes a weight lane loads): `inv` holds, per 16-element run v of the 256, its slot as
//           nibble v (t16_inverse_perm below).
// LAYOUT 4: LAYOUT 3 for at most 8 tokens, half the bytes: 2304-byte tiles { int8 frag[4][4 K-chunks][8 tokens][16];
//           ds[2 halves][2 token quads][4 groups][4 tokens] } (what ggq_quantize_q8_1_t16 writes for batch <= 8).
// element e256 of a 256-element unit -> (16-byte slot 4 f + c of the tile, byte inside the slot).  `inv` = the format's nibble
// table of 16-element runs, or GGQ_T16_RUN8 for the 32-element nibble blocks (Q4_0 Q4_1 Q5_0 Q5_1), whose weight lane
// (K-chunk c of MFMA f) holds 8 raw bytes = elements 8 h .. 8 h + 7 (low nibbles) and 16 + 8 h .. (high nibbles) of block
// 2 f + (c >> 1), h = c & 1: the slot is those two 8-element runs, low-nibble run first.
// LAYOUT 5: the x64 layout (mmq_x64.hip): per (ix/256, token/32) a 10240-byte record
//           { int8 frag[8 groups][2 K-halves][32 tokens][16]                        8192 bytes, one MFMA operand fragment = 1 KB in lane order
//             float d8[8 groups][2 halves h][4 quads qd][4 e]                        token 8 qd + 4 h + e: accumulator-register order of lane half h;
//                                                                                    the fp16-rounded d as fp32 for the need_sum formats, else d
//             fp16 s8[2 kh][32 tokens][8] = s8(4kh) s8(4kh+1) s8(4kh) s8(4kh+1) s8(4kh+2) s8(4kh+3) s8(4kh+2) s8(4kh+3)   at byte 9216: the
//                                                                                    operand of the min-term MFMA (need_sum formats) }
//           records of a 256-element K step are contiguous over the token tiles, whose count is rounded up to even (64-token units).
#define GGQ_T16_RUN8 (~0ull)
__device__ __forceinline__ void t16_slot(uint64_t inv, int e256, int& slot, int& byte) {
  if (inv == GGQ_T16_RUN8) {
    const int b = e256 >> 5, w = e256 & 31, h = (w >> 3) & 1, hi = w >> 4;
    slot = 4 * (b >> 1) + 2 * (b & 1) + h;
    byte = 8 * hi + (w & 7);
  } else {
    slot = (int)((inv >> (4 * (e256 >> 4))) & 15);
    byte = e256 & 15;
  }
}

template <int DT, int LAYOUT, bool NEED_SUM>
__global__ void __launch_bounds__(256) quantize_q8_1_kernel(const void* __restrict__ x,
                                                            uint8_t* __restrict__ q, int64_t batch,
                                                            int64_t k, int64_t padded, int64_t tok_off,
                                                            uint64_t inv) {
  const int64_t t = blockIdx.y + tok_off;
  // Lane l (bits b2 b1 b0) of an 8-lane group takes the 4-element chunk c = (b2, b1^b2, b0^b2) of its 32-group, so that the
  // three cross-lane levels of the reference's warp reduction (element ^16, ^8, ^4: ggml_kernel.cu quantize_q8_1, same fp32
  // order) are row_half_mirror, quad_perm[2,3,0,1] and quad_perm[1,0,3,2] — DPP modifiers of the max / add itself instead
  // of ds_bpermute round trips through the LDS pipe (15 per thread before).
  const int l8 = threadIdx.x & 7;
  const int64_t ix = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~7) + (((l8 >> 2) * 7) ^ (l8 & 3))) * 4;
  if (ix >= padded) return;  // padded % 32 == 0, so a 32-group is never split by this exit
  float v[4];
  load4<DT>(x, t * k, ix, k, v);

  auto dppf = [](float a, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), decltype(ctrl)::value, 0xF, 0xF, false)); };
  using HM = std::integral_constant<int, 0x141>;   // row_half_mirror: lane <-> 7 - lane
  using QP2 = std::integral_constant<int, 0x4E>;   // quad_perm [2,3,0,1]: lane ^ 2
  using QP1 = std::integral_constant<int, 0xB1>;   // quad_perm [1,0,3,2]: lane ^ 1
  float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  amax = fmaxf(amax, dppf(amax, HM{}));
  amax = fmaxf(amax, dppf(amax, QP2{}));
  amax = fmaxf(amax, dppf(amax, QP1{}));

  float sum = 0.0f;
  if (NEED_SUM) {
    float s[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = s[i] + dppf(s[i], HM{});    // element mask 16
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = s[i] + dppf(s[i], QP2{});   // element mask 8
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = s[i] + dppf(s[i], QP1{});   // element mask 4
    // element mask 2: (e, e^2) -> s0+s2, s1+s3 ; element mask 1: their sum
    sum = (s[0] + s[2]) + (s[1] + s[3]);
  }

  const float d = amax / 127;
  int qi[4];
  quant4(v, amax, d, qi);
  const uint32_t packed = (uint32_t)(qi[0] & 0xFF) | ((uint32_t)(qi[1] & 0xFF) << 8) |
                          ((uint32_t)(qi[2] & 0xFF) << 16) | ((uint32_t)(qi[3] & 0xFF) << 24);

  const int64_t g = ix >> 5;       // 32-element group index in the row
  const int e = (int)(ix & 31);    // element offset inside the group
  if (LAYOUT == 0) {
    uint8_t* blk = q + (t * (padded >> 5) + g) * 36;
    *(uint32_t*)(blk + 4 + e) = packed;
    if (e == 0) {
      const uint16_t hd = __builtin_bit_cast(uint16_t, (_Float16)d);
      const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
      *(uint32_t*)blk = (uint32_t)hd | ((uint32_t)hs << 16);
    }
  } else if (LAYOUT == 5) {
    const int64_t n_tt = ((batch + 63) >> 6) << 1;
    uint8_t* rec = q + ((g >> 3) * n_tt + (t >> 5)) * 10240;
    const int g8 = (int)(g & 7), tl = (int)(t & 31);
    *(uint32_t*)(rec + g8 * 1024 + (e >> 4) * 512 + tl * 16 + (e & 15)) = packed;
    if (e == 0) {
      const int idx = ((tl >> 2) & 1) * 16 + (tl >> 3) * 4 + (tl & 3);   // [h][qd][e]
      if (NEED_SUM) {   // the reference stores half2(d, sum) for these formats (mmq.cu:137-143): the kernel must see the fp16-rounded values
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
        *(float*)(rec + 8192 + g8 * 128 + idx * 4) = (float)(_Float16)d;
        uint16_t* s8 = (uint16_t*)(rec + 9216 + ((g8 >> 2) * 32 + tl) * 16);
        const int j = g8 & 3, p0 = (j >> 1) * 4 + (j & 1);
        s8[p0] = hs;
        s8[p0 + 2] = hs;
      } else {
        *(float*)(rec + 8192 + g8 * 128 + idx * 4) = d;
      }
    }
  } else if (LAYOUT == 4) {
    // batch <= 8: per ix/256 a 2304-byte tile { int8 frag[4][4 K-chunks][8 tokens][16]; ds[2 halves][2 token quads][4 groups][4 tokens] }
    uint8_t* tile = q + (g >> 3) * 2304;
    const int e256 = (int)(ix & 255), tl = (int)(t & 7);
    int slot, byte;
    t16_slot(inv, e256, slot, byte);
    *(uint32_t*)(tile + (slot >> 2) * 512 + ((slot & 3) * 8 + tl) * 16 + byte) = packed;
    if (e == 0) {
      const int g8 = (int)(g & 7);
      uint8_t* ds = tile + 2048 + ((((g8 >> 2) * 2 + (tl >> 2)) * 4 + (g8 & 3)) * 4 + (tl & 3)) * 4;
      if (NEED_SUM) {
        const uint16_t hd = __builtin_bit_cast(uint16_t, (_Float16)d);
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
        *(uint32_t*)ds = (uint32_t)hd | ((uint32_t)hs << 16);
      } else {
        *(float*)ds = d;
      }
    }
  } else if (LAYOUT == 3) {
    const int64_t n_tt = (batch + 15) >> 4;
    uint8_t* tile = q + ((g >> 3) * n_tt + (t >> 4)) * 4608;
    const int e256 = (int)(ix & 255), tl = (int)(t & 15);
    int slot, byte;   // fragment = slot >> 2, K-chunk = slot & 3
    t16_slot(inv, e256, slot, byte);
    *(uint32_t*)(tile + (slot >> 2) * 1024 + ((slot & 3) * 16 + tl) * 16 + byte) = packed;
    if (e == 0) {
      const int g8 = (int)(g & 7);
      uint8_t* ds = tile + 4096 + ((((g8 >> 2) * 4 + (tl >> 2)) * 4 + (g8 & 3)) * 4 + (tl & 3)) * 4;
      if (NEED_SUM) {
        const uint16_t hd = __builtin_bit_cast(uint16_t, (_Float16)d);
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
        *(uint32_t*)ds = (uint32_t)hd | ((uint32_t)hs << 16);
      } else {
        *(float*)ds = d;
      }
    }
  } else if (LAYOUT == 2) {
    const int64_t n_tt = (batch + 31) >> 5;
    uint8_t* tile = q + ((g >> 2) * n_tt + (t >> 5)) * 4608;
    const int slot = (int)(g & 3), tl = (int)(t & 31);
    *(uint32_t*)(tile + slot * 1024 + (e >> 4) * 512 + tl * 16 + (e & 15)) = packed;
    if (e == 0) {
      uint8_t* ds = tile + 4096 + (slot >> 1) * 256 + tl * 8 + (slot & 1) * 4;
      if (NEED_SUM) {
        const uint16_t hd = __builtin_bit_cast(uint16_t, (_Float16)d);
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
        *(uint32_t*)ds = (uint32_t)hd | ((uint32_t)hs << 16);
      } else {
        *(float*)ds = d;
      }
    }
  } else {
    uint8_t* blk = q + ((g >> 2) * batch + t) * 144;
    const int slot = (int)(g & 3);
    *(uint32_t*)(blk + 16 + 32 * slot + e) = packed;
    if (e == 0) {
      if (NEED_SUM) {
        const uint16_t hd = __builtin_bit_cast(uint16_t, (_Float16)d);
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)sum);
        *(uint32_t*)(blk + 4 * slot) = (uint32_t)hd | ((uint32_t)hs << 16);
      } else {
        *(float*)(blk + 4 * slot) = d;
      }
    }
  }
}

template <int LAYOUT, bool NEED_SUM>
static int launch_quant(const void* x, int dt, void* q, int64_t batch, int64_t k, int64_t padded,
                        hipStream_t s, uint64_t inv = 0) {
  if (batch == 0) return GGQ_OK;
  const dim3 block(256);
  const unsigned gx = (unsigned)((padded / 4 + 255) / 256);
  for (int64_t off = 0; off < batch; off += 65535) {  // grid.y limit, as ggml_kernel.cu:56-64
    const int64_t nb = batch - off < 65535 ? batch - off : 65535;
    const dim3 grid(gx, (unsigned)nb);
    const void* xo = x;
    uint8_t* qo = (uint8_t*)q;
    switch (dt) {
      case GGQ_F32:
        GGQ_HIP_PRE_LAUNCH();
        hipLaunchKernelGGL((quantize_q8_1_kernel<GGQ_F32, LAYOUT, NEED_SUM>), grid, block, 0, s, xo, qo, batch, k, padded, off, inv);
        break;
      case GGQ_F16:
        GGQ_HIP_PRE_LAUNCH();
        hipLaunchKernelGGL((quantize_q8_1_kernel<GGQ_F16, LAYOUT, NEED_SUM>), grid, block, 0, s, xo, qo, batch, k, padded, off, inv);
        break;
      case GGQ_BF16:
        GGQ_HIP_PRE_LAUNCH();
        hipLaunchKernelGGL((quantize_q8_1_kernel<GGQ_BF16, LAYOUT, NEED_SUM>), grid, block, 0, s, xo, qo, batch, k, padded, off, inv);
        break;
      default: return GGQ_ERR_DTYPE;
    }
    GGQ_HIP_CHECK_LAUNCH();
  }
  return GGQ_OK;
}

// Which 16-element run of a 256-element unit the (fragment f, K-chunk c) slot 4 f + c of the 16-token-tile layout holds,
// per weight format — the mirror of what the weight lane (row, c) of mmq_t16.hip gets out of its 16 loaded bytes:
//   Q4_K / Q5_K  f = 2 q + hi: lane c of load q holds bytes 16 (c & 1) .. + 15 of the 32-byte chunk of pair 2 q + (c >> 1);
//                its low nibbles are elements 64 p + 16 (c & 1) + 0..15 (group 2 p), its high nibbles the same of group 2 p + 1
//   legacy nibble formats (Q4_0 Q4_1 Q5_0 Q5_1): 8-element runs, see t16_slot (GGQ_T16_RUN8)
//   Q8_0         lane c of step s (64 elements) holds elements 64 s + 16 c .. + 15: identity
static uint64_t t16_inverse_perm(int type) {
  if (type == GGQ_TYPE_Q4_0 || type == GGQ_TYPE_Q4_1 || type == GGQ_TYPE_Q5_0 || type == GGQ_TYPE_Q5_1) return GGQ_T16_RUN8;
  int perm[16];
  for (int f = 0; f < 4; ++f)
    for (int c = 0; c < 4; ++c) {
      int v;
      if (type == GGQ_TYPE_Q4_K || type == GGQ_TYPE_Q5_K) v = 8 * (f >> 1) + 4 * (c >> 1) + 2 * (f & 1) + (c & 1);
      else v = 4 * f + c;   // Q8_0: identity
      perm[4 * f + c] = v;
    }
  uint64_t inv = 0;
  for (int s = 0; s < 16; ++s) inv |= (uint64_t)s << (4 * perm[s]);
  return inv;
}

}  // namespace ggq

extern "C" int ggq_quantize_q8_1_t16(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                                     int type, void* stream) {
  using namespace ggq;
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (x_dtype < GGQ_F32 || x_dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (!ggq_mmq_t16_type_supported(type)) return GGQ_ERR_TYPE;
  if (batch == 0) return GGQ_OK;
  if (!x || !q) return GGQ_ERR_ARG;
  if ((uintptr_t)q & 15) return GGQ_ERR_ALIGN;
  const int64_t padded = ggq_mmq_padded_k(k);
  const uint64_t inv = t16_inverse_perm(type);
  if (batch <= 8) {   // the 8-token form of the layout (mmq_t16.hip's M8 mode)
    if (ggq_mmq_need_sum(type)) return launch_quant<4, true>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream, inv);
    return launch_quant<4, false>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream, inv);
  }
  if (ggq_mmq_need_sum(type))
    return launch_quant<3, true>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream, inv);
  return launch_quant<3, false>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream, inv);
}

extern "C" int ggq_quantize_q8_1_x64(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                                     int type, void* stream) {
  using namespace ggq;
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (x_dtype < GGQ_F32 || x_dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (!ggq_mmq_x64_type_supported(type)) return GGQ_ERR_TYPE;
  if (batch == 0) return GGQ_OK;
  if (!ggq_mmq_x64_supported(type, k, batch)) return GGQ_ERR_SHAPE;
  if (!x || !q) return GGQ_ERR_ARG;
  if ((uintptr_t)q & 15) return GGQ_ERR_ALIGN;
  if (ggq_mmq_need_sum(type)) return launch_quant<5, true>(x, x_dtype, q, batch, k, k, (hipStream_t)stream);
  return launch_quant<5, false>(x, x_dtype, q, batch, k, k, (hipStream_t)stream);
}

extern "C" int ggq_quantize_q8_1(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                                 void* stream) {
  using namespace ggq;
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (x_dtype < GGQ_F32 || x_dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (batch == 0) return GGQ_OK;
  if (!x || !q) return GGQ_ERR_ARG;
  if ((uintptr_t)q & 3) return GGQ_ERR_ALIGN;
  return launch_quant<0, true>(x, x_dtype, q, batch, k, ggq_mmvq_padded_k(k), (hipStream_t)stream);
}

extern "C" int ggq_quantize_q8_1_tiled(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                                       int type, void* stream) {
  using namespace ggq;
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (x_dtype < GGQ_F32 || x_dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (batch == 0) return GGQ_OK;
  if (!x || !q) return GGQ_ERR_ARG;
  if ((uintptr_t)q & 15) return GGQ_ERR_ALIGN;
  const int64_t padded = ggq_mmq_padded_k(k);
  if (ggq_mmq_need_sum(type))
    return launch_quant<2, true>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream);
  return launch_quant<2, false>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream);
}

extern "C" int ggq_quantize_q8_1_mmq(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                                     int type, void* stream) {
  using namespace ggq;
  if (batch < 0 || k <= 0) return GGQ_ERR_ARG;
  if (x_dtype < GGQ_F32 || x_dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (batch == 0) return GGQ_OK;
  if (!x || !q) return GGQ_ERR_ARG;
  if ((uintptr_t)q & 15) return GGQ_ERR_ALIGN;
  const int64_t padded = ggq_mmq_padded_k(k);
  if (ggq_mmq_need_sum(type))
    return launch_quant<1, true>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream);
  return launch_quant<1, false>(x, x_dtype, q, batch, k, padded, (hipStream_t)stream);
}
