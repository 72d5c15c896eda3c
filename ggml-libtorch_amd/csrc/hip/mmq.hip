// mmq.hip — quantised GEMM  Y[B,N] = X[B,K] (Q8_1) · W[N,K]^T (block-quant), int8 MFMA, gfx950.
//
// Replaces mul_mat_q (HK/ggml/mmq.cuh:1917-1986), its tile loaders / vec_dot bodies
// (mmq.cuh:257-1737), the NVIDIA mma.sync fragments (HK/ggml/mma.cuh), the tile
// heuristic (kernel_instances/mmq_kernel.cuh:11-86) and the op body of ggml_mul_mat_a8
// (HK/ggml/mmq.cu:180-255).
//
// Numerical contract ("MMQ canon"): per 32-element group the integer
// contraction C = Σ q_w·q8 is exact (int8 MFMA, int32 accumulate) with the operands of
// the reference's tensor-core bodies (Q4_0: nibble-8, Q5_0: q-16, Q6_K: q-32, Q3_K:
// q2-4·¬h, raw unsigned for Q4_1/Q5_1/Q4_K/Q5_K/Q2_K); the float combination uses the
// same factors (fp16 products for Q4_1/Q5_1, fp16 d8/s8 for need_sum formats, fp32 d8
// otherwise, s8 — not d8·Σq8 — for the Q4_K/Q5_K min term); fp32 accumulation order ours.
//
// Structure (v1):
//   workgroup = 256 threads = 4 waves = TB token-blocks x KS k-splits (TB·KS = 4);
//   workgroup tile = 32 weight rows x 32·TB tokens; per outer step KS consecutive
//   256-element K slabs are staged:
//     * weights: global -> registers -> unpacked to signed int8 in LDS ([row][256+16 pad],
//       the pad makes the 32-lane ds_read_b128 A-fragment reads conflict-free), per-(row,
//       group) float scales in LDS laid out [group][row] so a lane fetches the 16 row
//       scales of its accumulator registers with 4 broadcast ds_read_b128;
//     * activations: the block_q8_1_mmq scratch is already MFMA-friendly (144-byte token
//       pitch = 9 x 16 B, conflict-free for ds_read_b128), copied verbatim with 16-B accesses;
//   wave (tb, ks): one v_mfma_i32_32x32x32_i8 per 32-element group (two 32x32x16 for the
//   16-element-scale format Q6_K), lane = token, accumulator register = weight row, then
//   acc[i] += (float(C[i]) · d8_lane) · sA[i]  (+ mA[i] · s8_lane).
//   KS > 1 (small batches, HBM-bound): the k-split partials are reduced through LDS.
#include "ggq_common.h"

namespace ggq {

constexpr int WROW = 272;  // LDS pitch of one unpacked int8 weight row (256 + 16)

template <int T> struct MmqTraits {
  static constexpr bool kquant = Fmt<T>::QK == 256;
  static constexpr bool need_sum = T == GGQ_TYPE_Q4_0 || T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_1 ||
                                   T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K;  // mmq.cu:84-106
  static constexpr bool fp16_prod = T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_1;  // __hmul2(dm, ds8)
  static constexpr bool has_min = T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K || fp16_prod;
  static constexpr bool two_tiles = T == GGQ_TYPE_Q2_K;  // second int8 tile carries the mins
  static constexpr bool half_scales = T == GGQ_TYPE_Q6_K;  // scale per 16 elements -> K=16 MFMAs
  // float scale arrays per (group,row): 1 = sA; 2 = sA + (mA | sA1 | dmin)
  static constexpr int n_scale = (has_min || two_tiles || half_scales) ? 2 : 1;
};

// (x - c) per byte for x in [0, 2c): exact, no inter-byte borrow
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t c4) {
  return ((x | 0x80808080u) - c4) ^ 0x80808080u;
}
__device__ __forceinline__ uint32_t spread4b(uint32_t x) {
  return ((x & 1) << 4) | ((x & 2) << 11) | ((x & 4) << 18) | ((x & 8) << 25);
}

// Unpack the 32-element group G (global index along K) of one weight row into 8 dwords of
// signed int8 (w[]), an optional second tile (w2[], Q2_K mins) and its float scales.
template <int T>
__device__ __forceinline__ void unpack_group(const uint8_t* row, int G, uint32_t w[8], uint32_t w2[8],
                                             float& s0, float& s1) {
  s0 = 0.0f; s1 = 0.0f;
  if constexpr (T == GGQ_TYPE_Q4_0) {
    const uint8_t* b = row + (int64_t)G * 18;
    const u32x4_a2 q = ld_u32x4(b + off::Q4_0_QS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes(q.v[i] & 0x0F0F0F0F, 0x08080808u);           // mmq.cuh:359
      w[4 + i] = sub_bytes((q.v[i] >> 4) & 0x0F0F0F0F, 0x08080808u);
    }
    s0 = bits_h_f32(ld_u16(b));
  } else if constexpr (T == GGQ_TYPE_Q4_1) {
    const uint8_t* b = row + (int64_t)G * 20;
    const u32x4_a2 q = ld_u32x4(b + off::Q4_1_QS);
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = q.v[i] & 0x0F0F0F0F; w[4 + i] = (q.v[i] >> 4) & 0x0F0F0F0F; }
    const uint32_t dm = ld_u32(b);
    s0 = bits_h_f32(dm & 0xFFFF); s1 = bits_h_f32(dm >> 16);
  } else if constexpr (T == GGQ_TYPE_Q5_0) {
    const uint8_t* b = row + (int64_t)G * 22;
    const uint32_t qh = ld_u32(b + off::Q5_0_QH);
    const u32x4_a2 q = ld_u32x4(b + off::Q5_0_QS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes((q.v[i] & 0x0F0F0F0F) | spread4b(qh >> (4 * i)), 0x10101010u);  // mmq.cuh:561
      w[4 + i] = sub_bytes(((q.v[i] >> 4) & 0x0F0F0F0F) | spread4b(qh >> (16 + 4 * i)), 0x10101010u);
    }
    s0 = bits_h_f32(ld_u16(b));
  } else if constexpr (T == GGQ_TYPE_Q5_1) {
    const uint8_t* b = row + (int64_t)G * 24;
    const uint32_t qh = ld_u32(b + off::Q5_1_QH);
    const u32x4_a2 q = ld_u32x4(b + off::Q5_1_QS);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = (q.v[i] & 0x0F0F0F0F) | spread4b(qh >> (4 * i));
      w[4 + i] = ((q.v[i] >> 4) & 0x0F0F0F0F) | spread4b(qh >> (16 + 4 * i));
    }
    const uint32_t dm = ld_u32(b);
    s0 = bits_h_f32(dm & 0xFFFF); s1 = bits_h_f32(dm >> 16);
  } else if constexpr (T == GGQ_TYPE_Q8_0) {
    const uint8_t* b = row + (int64_t)G * 34;
    const u32x4_a2 q0 = ld_u32x4(b + off::Q8_0_QS), q1 = ld_u32x4(b + off::Q8_0_QS + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = q0.v[i]; w[4 + i] = q1.v[i]; }
    s0 = bits_h_f32(ld_u16(b));
  } else if constexpr (T == GGQ_TYPE_Q2_K) {
    const int ib = G >> 3, gl = G & 7, n = gl >> 2, j = gl & 3;
    const uint8_t* b = row + (int64_t)ib * 84;
    const u32x4_a2 q0 = ld_u32x4(b + off::Q2_K_QS + 32 * n), q1 = ld_u32x4(b + off::Q2_K_QS + 32 * n + 16);
    const int sc0 = b[off::Q2_K_SC + 2 * gl], sc1 = b[off::Q2_K_SC + 2 * gl + 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // bytes <= 3 * 15: the dword multiply cannot carry between bytes
      w[i] = ((q0.v[i] >> (2 * j)) & 0x03030303u) * (uint32_t)(sc0 & 0xF);
      w[4 + i] = ((q1.v[i] >> (2 * j)) & 0x03030303u) * (uint32_t)(sc1 & 0xF);
      w2[i] = 0x01010101u * (uint32_t)(sc0 >> 4);
      w2[4 + i] = 0x01010101u * (uint32_t)(sc1 >> 4);
    }
    const uint32_t dm = ld_u32(b + off::Q2_K_D);
    s0 = bits_h_f32(dm & 0xFFFF); s1 = bits_h_f32(dm >> 16);
  } else if constexpr (T == GGQ_TYPE_Q3_K) {
    const int ib = G >> 3, gl = G & 7, n = gl >> 2, j = gl & 3;
    const uint8_t* b = row + (int64_t)ib * 110;
    const u32x4_a2 q0 = ld_u32x4(b + off::Q3_K_QS + 32 * n), q1 = ld_u32x4(b + off::Q3_K_QS + 32 * n + 16);
    const u32x4_a2 h0 = ld_u32x4(b + off::Q3_K_HM), h1 = ld_u32x4(b + off::Q3_K_HM + 16);
    const u32x3_a2 s = ld_u32x3(b + off::Q3_K_SC);
    const int sc0 = q3k_scale(s.v[0], s.v[1], s.v[2], 2 * gl), sc1 = q3k_scale(s.v[0], s.v[1], s.v[2], 2 * gl + 1);
    // tile holds -(q3 * sc) in [-128, 124] (q3*sc itself reaches +128); the sign goes into s0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t o0 = 0, o1 = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int v0 = (int)((q0.v[i] >> (8 * c + 2 * j)) & 3) - (((h0.v[i] >> (8 * c + gl)) & 1) ? 0 : 4);
        const int v1 = (int)((q1.v[i] >> (8 * c + 2 * j)) & 3) - (((h1.v[i] >> (8 * c + gl)) & 1) ? 0 : 4);
        o0 |= (uint32_t)((-(v0 * sc0)) & 0xFF) << (8 * c);
        o1 |= (uint32_t)((-(v1 * sc1)) & 0xFF) << (8 * c);
      }
      w[i] = o0; w[4 + i] = o1;
    }
    s0 = -bits_h_f32(ld_u16(b + off::Q3_K_D));
  } else if constexpr (T == GGQ_TYPE_Q4_K) {
    const int ib = G >> 3, gl = G & 7, il = gl >> 1, nib = gl & 1;
    const uint8_t* b = row + (int64_t)ib * 144;
    const u32x4_a2 hd = ld_u32x4(b);
    const u32x4_a2 q0 = ld_u32x4(b + off::Q4_K_QS + 32 * il), q1 = ld_u32x4(b + off::Q4_K_QS + 32 * il + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = (q0.v[i] >> (4 * nib)) & 0x0F0F0F0F;
      w[4 + i] = (q1.v[i] >> (4 * nib)) & 0x0F0F0F0F;
    }
    int sc, mn;
    k4_scale_min(hd.v[1], hd.v[2], hd.v[3], gl, sc, mn);
    s0 = bits_h_f32(hd.v[0] & 0xFFFF) * (float)sc;
    s1 = -(bits_h_f32(hd.v[0] >> 16) * (float)mn);
  } else if constexpr (T == GGQ_TYPE_Q5_K) {
    const int ib = G >> 3, gl = G & 7, il = gl >> 1, nib = gl & 1;
    const uint8_t* b = row + (int64_t)ib * 176;
    const u32x4_a2 hd = ld_u32x4(b);
    const u32x4_a2 q0 = ld_u32x4(b + off::Q5_K_QS + 32 * il), q1 = ld_u32x4(b + off::Q5_K_QS + 32 * il + 16);
    const u32x4_a2 h0 = ld_u32x4(b + off::Q5_K_QH), h1 = ld_u32x4(b + off::Q5_K_QH + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = ((q0.v[i] >> (4 * nib)) & 0x0F0F0F0F) | (((h0.v[i] >> gl) & 0x01010101u) << 4);
      w[4 + i] = ((q1.v[i] >> (4 * nib)) & 0x0F0F0F0F) | (((h1.v[i] >> gl) & 0x01010101u) << 4);
    }
    int sc, mn;
    k4_scale_min(hd.v[1], hd.v[2], hd.v[3], gl, sc, mn);
    s0 = bits_h_f32(hd.v[0] & 0xFFFF) * (float)sc;
    s1 = -(bits_h_f32(hd.v[0] >> 16) * (float)mn);
  } else if constexpr (T == GGQ_TYPE_Q6_K) {
    const int ib = G >> 3, gl = G & 7, ip = gl >> 2, j = gl & 3;
    const uint8_t* b = row + (int64_t)ib * 210;
    const uint8_t* pl = b + off::Q6_K_QL + 64 * ip + 32 * (j & 1);
    const uint8_t* ph = b + off::Q6_K_QH + 32 * ip;
    const u32x4_a2 l0 = ld_u32x4(pl), l1 = ld_u32x4(pl + 16);
    const u32x4_a2 h0 = ld_u32x4(ph), h1 = ld_u32x4(ph + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = sub_bytes(((l0.v[i] >> (4 * (j >> 1))) & 0x0F0F0F0F) | (((h0.v[i] >> (2 * j)) & 0x03030303u) << 4), 0x20202020u);
      w[4 + i] = sub_bytes(((l1.v[i] >> (4 * (j >> 1))) & 0x0F0F0F0F) | (((h1.v[i] >> (2 * j)) & 0x03030303u) << 4), 0x20202020u);
    }
    const float d = bits_h_f32(ld_u16(b + off::Q6_K_D));
    s0 = d * (float)(int8_t)b[off::Q6_K_SC + 2 * gl];
    s1 = d * (float)(int8_t)b[off::Q6_K_SC + 2 * gl + 1];
  }
}

// LDS carve-up (bytes), all offsets multiples of 16
template <int T, int TB, int KS> struct MmqLds {
  static constexpr int TT = 32 * TB;                       // tokens per workgroup tile
  static constexpr int W_TILE = 32 * WROW;                 // one int8 weight tile
  static constexpr int W_BYTES = KS * W_TILE * (MmqTraits<T>::two_tiles ? 2 : 1);
  static constexpr int S_BYTES = KS * MmqTraits<T>::n_scale * 8 * 32 * 4;
  static constexpr int A_SLAB = 2 * TT * 144;              // one 256-element K slab of activations
  static constexpr int A_BYTES = KS * A_SLAB;
  static constexpr int W_OFF = 0;
  static constexpr int S_OFF = W_OFF + W_BYTES;
  static constexpr int A_OFF = S_OFF + S_BYTES;
  static constexpr int TOTAL = A_OFF + A_BYTES;
  static constexpr int RED_BYTES = KS > 1 ? 4 * 16 * 64 * 4 : 0;  // k-split reduction (aliases the tiles)
  static constexpr int BYTES = TOTAL > RED_BYTES ? TOTAL : RED_BYTES;
};

template <int T, int DT, int TB, int KS>
__global__ void __launch_bounds__(256) mmq_kernel(const uint8_t* __restrict__ w,
                                                  const uint8_t* __restrict__ q8,
                                                  void* __restrict__ y, int k, int n_rows, int batch,
                                                  int64_t ldy) {
  using L = MmqLds<T, TB, KS>;
  using TR = MmqTraits<T>;
  constexpr int TT = L::TT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* lw = lds + L::W_OFF;
  float* ls = (float*)(lds + L::S_OFF);
  uint8_t* la = lds + L::A_OFF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tb = wave % TB, ks = wave / TB;
  const int n0 = blockIdx.x * 32;       // first weight row of the tile
  const int t0 = blockIdx.y * TT;       // first token of the tile
  const int n_valid_tok = min(TT, batch - t0);
  const int64_t row_bytes = (int64_t)(k / Fmt<T>::QK) * Fmt<T>::BS;
  const int n_groups = k / 32;          // 32-element groups along K
  const int n_slabs = (k + 255) / 256;  // 256-element K slabs

  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

  // staging role of this thread: weight row sr (0..31), group sg (0..7) of a slab
  const int sr = tid >> 3, sg = tid & 7;
  const uint8_t* srow = w + (int64_t)min(n0 + sr, n_rows - 1) * row_bytes;

  for (int slab0 = 0; slab0 < n_slabs; slab0 += KS) {
    // ---- stage weights (unpack) ----
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int G = (slab0 + s) * 8 + sg;
      uint32_t wq[8], wq2[8];
      float s0 = 0.0f, s1 = 0.0f;
      if (G < n_groups) {
        unpack_group<T>(srow, G, wq, wq2, s0, s1);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { wq[i] = 0; wq2[i] = 0; }
      }
      uint8_t* dst = lw + s * L::W_TILE + sr * WROW + 32 * sg;
      *(v4i*)dst = v4i{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
      *(v4i*)(dst + 16) = v4i{(int)wq[4], (int)wq[5], (int)wq[6], (int)wq[7]};
      if constexpr (TR::two_tiles) {
        uint8_t* dst2 = dst + KS * L::W_TILE;
        *(v4i*)dst2 = v4i{(int)wq2[0], (int)wq2[1], (int)wq2[2], (int)wq2[3]};
        *(v4i*)(dst2 + 16) = v4i{(int)wq2[4], (int)wq2[5], (int)wq2[6], (int)wq2[7]};
      }
      float* sdst = ls + (s * TR::n_scale) * 256 + sg * 32 + sr;
      sdst[0] = s0;
      if constexpr (TR::n_scale == 2) sdst[256] = s1;
    }
    // ---- stage activations: KS slabs x 2 blocks x n_valid_tok x 144 B, contiguous per block ----
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int64_t kblock = (int64_t)(slab0 + s) * 2 + kb;
        const uint8_t* src = q8 + (kblock * batch + t0) * 144;
        uint8_t* dst = la + s * L::A_SLAB + kb * TT * 144;
        // scratch holds padded/128 >= 2*n_slabs blocks per token: every kblock of a real slab exists
        const int n16 = (slab0 + s < n_slabs) ? n_valid_tok * 9 : 0;  // 144 B = 9 x 16 B per token
        for (int i = tid; i < n16; i += 256) *(v4i*)(dst + 16 * i) = *(const v4i*)(src + 16 * i);
      }
    }
    __syncthreads();

    // ---- compute: wave (tb, ks) on slab slab0+ks ----
    if (slab0 + ks < n_slabs) {
      const uint8_t* wt = lw + ks * L::W_TILE;
      const float* st = ls + (ks * TR::n_scale) * 256;
      const uint8_t* at = la + ks * L::A_SLAB;
      const int r = lane & 31, h = lane >> 5;
      const int tl = tb * 32 + r;  // token within the tile
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const uint8_t* ablk = at + ((g >> 2) * TT + tl) * 144;
        const uint32_t dsw = *(const uint32_t*)(ablk + 4 * (g & 3));
        float bs, bm = 0.0f;
        if constexpr (TR::need_sum) { bs = bits_h_f32(dsw & 0xFFFF); bm = bits_h_f32(dsw >> 16); }
        else bs = __builtin_bit_cast(float, dsw);

        v16i c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
        if constexpr (TR::half_scales) {
          const long a0 = *(const long*)(wt + r * WROW + 32 * g + 8 * h);
          const long a1 = *(const long*)(wt + r * WROW + 32 * g + 16 + 8 * h);
          const long b0 = *(const long*)(ablk + 16 + 32 * (g & 3) + 8 * h);
          const long b1 = *(const long*)(ablk + 16 + 32 * (g & 3) + 16 + 8 * h);
          c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a0, b0, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, b1, c1, 0, 0, 0);
        } else {
          const v4i a = *(const v4i*)(wt + r * WROW + 32 * g + 16 * h);
          const v4i b = *(const v4i*)(ablk + 16 + 32 * (g & 3) + 16 * h);
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
          if constexpr (TR::two_tiles) {
            const v4i a2 = *(const v4i*)(wt + KS * L::W_TILE + r * WROW + 32 * g + 16 * h);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b, c1, 0, 0, 0);
          }
        }
        // accumulator register i <-> tile row (i&3) + 8(i>>2) + 4h : 4 runs of 4 consecutive rows
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const v4f sa = *(const v4f*)(st + g * 32 + 8 * qd + 4 * h);
          v4f sb = {0, 0, 0, 0};
          if constexpr (TR::n_scale == 2) sb = *(const v4f*)(st + 256 + g * 32 + 8 * qd + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * qd + e;
            // explicit fmaf + -ffp-contract=off: every accumulator register sees the same
            // instruction sequence, so a row's result does not depend on its position in the tile
            if constexpr (TR::fp16_prod) {  // mmq.cuh:527-529 / :840-842
              const float lo = (float)((_Float16)sa[e] * (_Float16)bs);
              const float hi = (float)((_Float16)sb[e] * (_Float16)bm);
              acc[i] += __builtin_fmaf(lo, (float)c0[i], hi);
            } else if constexpr (TR::two_tiles) {  // Q2_K: d8 (dall·Σsc q q8 − dmin·Σ m q8), mmq.cuh:47
              acc[i] = __builtin_fmaf(bs, __builtin_fmaf(sa[e], (float)c0[i], -(sb[e] * (float)c1[i])), acc[i]);
            } else if constexpr (TR::half_scales) {  // Q6_K: mmq.cuh:1726-1732
              acc[i] = __builtin_fmaf((float)c0[i] * bs, sa[e], acc[i]);
              acc[i] = __builtin_fmaf((float)c1[i] * bs, sb[e], acc[i]);
            } else if constexpr (TR::has_min) {  // Q4_K/Q5_K: dall sc C d8 − dmin m s8, mmq.cuh:1352-1359
              acc[i] = __builtin_fmaf((float)c0[i] * bs, sa[e], acc[i]);
              acc[i] = __builtin_fmaf(sb[e], bm, acc[i]);
            } else {  // Q4_0 / Q5_0 / Q8_0 / Q3_K: d_w d8 C
              acc[i] = __builtin_fmaf((float)c0[i] * bs, sa[e], acc[i]);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- k-split reduction through LDS ----
  if constexpr (KS > 1) {
    float* red = (float*)lds;  // [wave][16][64]
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (ks == 0) {
#pragma unroll
      for (int s = 1; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[((s * TB + tb) * 16 + i) * 64 + lane];
    }
  }

  // ---- write back: lane = token, register i = row ----
  if (ks == 0) {
    const int t = t0 + tb * 32 + (lane & 31);
    const int h = lane >> 5;
    if (t < batch) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < n_rows) Elem<DT>::st(y, (int64_t)t * ldy + row, acc[i]);
      }
    }
  }
}

template <int T, int DT, int TB, int KS>
static int launch_mmq_cfg(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                          int64_t ldy, hipStream_t s) {
  using L = MmqLds<T, TB, KS>;
  auto kern = mmq_kernel<T, DT, TB, KS>;
  static bool attr_set = false;  // one-time per instantiation (the reference does it on every call)
  if (L::BYTES > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES) != hipSuccess)
      return GGQ_ERR_LAUNCH;
    attr_set = true;
  }
  const dim3 grid((unsigned)((n + 31) / 32), (unsigned)((batch + L::TT - 1) / L::TT));
  if (grid.y > 65535) return GGQ_ERR_SHAPE;
  hipLaunchKernelGGL(kern, grid, dim3(256), L::BYTES, s, (const uint8_t*)w, (const uint8_t*)q8, y,
                     (int)k, (int)n, (int)batch, ldy);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T, int DT>
static int launch_mmq_t(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                        int64_t ldy, hipStream_t s) {
  if (batch <= 32) return launch_mmq_cfg<T, DT, 1, 4>(w, q8, y, batch, k, n, ldy, s);
  if (batch <= 64) return launch_mmq_cfg<T, DT, 2, 2>(w, q8, y, batch, k, n, ldy, s);
  return launch_mmq_cfg<T, DT, 4, 1>(w, q8, y, batch, k, n, ldy, s);
}

template <int T>
static int launch_mmq(const void* w, const void* q8, void* y, int dt, int64_t batch, int64_t k,
                      int64_t n, int64_t ldy, hipStream_t s) {
  switch (dt) {
    case GGQ_F32: return launch_mmq_t<T, GGQ_F32>(w, q8, y, batch, k, n, ldy, s);
    case GGQ_F16: return launch_mmq_t<T, GGQ_F16>(w, q8, y, batch, k, n, ldy, s);
    case GGQ_BF16: return launch_mmq_t<T, GGQ_BF16>(w, q8, y, batch, k, n, ldy, s);
    default: return GGQ_ERR_DTYPE;
  }
}

}  // namespace ggq

extern "C" int ggq_mul_mat_q_prequant(const void* w, const void* q, void* y, int type, int dtype,
                                      int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                                      void* stream) {
  using namespace ggq;
  if (k <= 0 || n_rows < 0 || batch < 0 || ldy < n_rows) return GGQ_ERR_ARG;
  if (!ggq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (k > (1 << 30) || n_rows > 0x7fffffffLL - 64 || batch > 0x7fffffffLL / 256) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0 || batch == 0) return GGQ_OK;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 15)) return GGQ_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
    case GGQ_TYPE_Q4_0: return launch_mmq<GGQ_TYPE_Q4_0>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q4_1: return launch_mmq<GGQ_TYPE_Q4_1>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q5_0: return launch_mmq<GGQ_TYPE_Q5_0>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q5_1: return launch_mmq<GGQ_TYPE_Q5_1>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q8_0: return launch_mmq<GGQ_TYPE_Q8_0>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q2_K: return launch_mmq<GGQ_TYPE_Q2_K>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q3_K: return launch_mmq<GGQ_TYPE_Q3_K>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q4_K: return launch_mmq<GGQ_TYPE_Q4_K>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q5_K: return launch_mmq<GGQ_TYPE_Q5_K>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    case GGQ_TYPE_Q6_K: return launch_mmq<GGQ_TYPE_Q6_K>(w, q, y, dtype, batch, k, n_rows, ldy, s);
    default: return GGQ_ERR_TYPE;
  }
}

extern "C" int ggq_mul_mat_q_ld(const void* w, const void* x, void* y, int type, int dtype,
                                int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                                void* scratch, void* stream) {
  if (!scratch) return GGQ_ERR_ARG;
  int rc = ggq_quantize_q8_1_mmq(x, dtype, scratch, batch, k, type, stream);
  if (rc != GGQ_OK) return rc;
  return ggq_mul_mat_q_prequant(w, scratch, y, type, dtype, batch, k, n_rows, ldy, stream);
}

extern "C" int ggq_mul_mat_q(const void* w, const void* x, void* y, int type, int dtype,
                             int64_t batch, int64_t k, int64_t n_rows, void* scratch, void* stream) {
  return ggq_mul_mat_q_ld(w, x, y, type, dtype, batch, k, n_rows, n_rows, scratch, stream);
}
