// mmvq.hip — quantised GEMV  y[1,N] = W[N,K] (block-quant) · x[1,K] (Q8_1), gfx950.
//
// Replaces mul_mat_vec_q + the per-format launchers (HK/ggml/mmvq.cuh:2-128), the
// vec_dot_<fmt>_q8_1 library (HK/ggml/vecdotq.cuh:43-605) and the op body of
// ggml_mul_mat_vec_a8 (HK/ggml/ggml_kernel.cu:80-193).
//
// Numerical contract ("MMVQ canon"): the integer dot products are exact;
// each unit's float combination uses the formula of the matching vec_dot_*_q8_1_impl
// (incl. the fp16 products of Q4_1/Q5_1 and d8*Σq8 for the Q4_K/Q5_K/Q2_K min term);
// only the fp32 summation order over the row differs from the reference.
//
// Design (HBM-bound: the weight row is read exactly once, nothing else matters):
//   * weights go global -> VGPR with 16-byte loads, no LDS (each byte is used by one lane);
//     a lane owns one 16-byte quant slice ("unit") per step, consecutive lanes own
//     consecutive slices, so a wave's loads sweep a contiguous span of the row;
//   * the Q8_1 activation row (K bytes + scales) is staged once per workgroup in LDS,
//     de-interleaved into a contiguous int8 array (16-byte aligned ds_read_b128),
//     float d / s arrays and per-16 integer sums (min / offset terms);
//   * one wave per row group, ROWS rows in flight per wave for memory-level parallelism,
//     64-lane shuffle reduction at the end of each row.
#include "ggq_common.h"
#include "iq_common.h"
#include <type_traits>

#ifndef GGQ_MMVQ_UNROLL
#define GGQ_MMVQ_UNROLL 1
#endif
#ifndef GGQ_MMVQ_KSPLIT
#define GGQ_MMVQ_KSPLIT 1   // 0: ablation build without the K-split instantiation (scripts/build_variant.sh)
#endif
#ifndef GGQ_MMVQ_ROWS
#define GGQ_MMVQ_ROWS 3   // rows in flight per wave in the fused kernel
#endif

#if defined(GGQ_VSTAMP)
__device__ unsigned long long g_vstamps[8192 * 8];
extern "C" int ggq_debug_read_vstamps(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vstamps), n * 8); }
#define VSTAMP(i) do { if ((threadIdx.x & 63) == 0) g_vstamps[((blockIdx.x * 16 + (threadIdx.x >> 6)) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define VSTAMP(i) do {} while (0)
#endif

namespace ggq {

struct ActLds {
  const int8_t* xq;    // [K]     int8 activations, element order
  const float* xd;     // [K/32]  d8
  const float* xs;     // [K/32]  s8 (fp16-rounded Σx, as stored by quantize_q8_1)
  const int* xi16;     // [K/16]  Σ q8 over each 16 elements (exact)
  const void* grid;    // the IQ format's codebook, staged in LDS by the kernel (nullptr for the other formats)
};

__device__ __forceinline__ v4i lds_ld16(const int8_t* p) { return *(const v4i*)p; }
__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

__device__ __forceinline__ int dot16(const uint32_t v[4], v4i a) {
  int s = sdot4((int)v[0], a[0], 0);
  s = sdot4((int)v[1], a[1], s);
  s = sdot4((int)v[2], a[2], s);
  return sdot4((int)v[3], a[3], s);
}

// spread the low 4 bits of x to bit 4 of the four bytes of a dword
__device__ __forceinline__ uint32_t spread4(uint32_t x) {   // one multiply: the four partial products do not overlap
  return (((x & 0xF) * 0x00204081u) & 0x01010101u) << 4;
}

// UnitDot<T>::run(row pointer, unit index u, activations) -> this lane's partial sum.
template <int T> struct UnitDot;
// formats whose UnitDot has the load / dot split (Raw, load(), dot())
template <int T> struct UnitHasPre {
  static constexpr bool value = T == GGQ_TYPE_Q4_0 || T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_0 || T == GGQ_TYPE_Q5_1 ||
                                T == GGQ_TYPE_Q8_0 || T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q6_K || T == GGQ_TYPE_IQ4_NL || T == GGQ_TYPE_IQ4_XS;
  // (Q5_K: three 16-byte loads per unit x 3 rows spill at 128 VGPRs)
};

template <> struct UnitDot<GGQ_TYPE_Q4_0> {  // vecdotq.cuh:45-65, 347-363
  static constexpr int UPB = 1;
  // (load / dot split: the fused kernel issues the loads of a wave's first rows BEFORE it quantises x)
  struct Raw { uint32_t d; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 18;
    return Raw{ld_u16(b), ld_u32x4_stream(b + off::Q4_0_QS)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const float d4 = bits_h_f32(R.d);
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = q.v[i] & 0x0F0F0F0F; hi[i] = (q.v[i] >> 4) & 0x0F0F0F0F; }
    const int sumi = dot16(lo, a0) + dot16(hi, a1);
    return d4 * (sumi * A.xd[u] - 8 * A.xs[u]);
  }
};
template <> struct UnitDot<GGQ_TYPE_Q4_1> {  // vecdotq.cuh:69-91, 365-381
  static constexpr int UPB = 1;
  struct Raw { uint32_t dm; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 20;
    return Raw{ld_u32(b), ld_u32x4_stream(b + off::Q4_1_QS)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const uint32_t dm = R.dm;
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = q.v[i] & 0x0F0F0F0F; hi[i] = (q.v[i] >> 4) & 0x0F0F0F0F; }
    const int sumi = dot16(lo, a0) + dot16(hi, a1);
    const float d4d8 = (float)(bits_h(dm & 0xFFFF) * (_Float16)A.xd[u]);  // __hmul2(dm4, ds8)
    const float m4s8 = (float)(bits_h(dm >> 16) * (_Float16)A.xs[u]);
    return sumi * d4d8 + m4s8;
  }
};
template <> struct UnitDot<GGQ_TYPE_Q5_0> {  // vecdotq.cuh:95-124, 383-401
  static constexpr int UPB = 1;
  struct Raw { uint32_t d, qh; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 22;
    return Raw{ld_u16(b), ld_u32(b + off::Q5_0_QH), ld_u32x4_stream(b + off::Q5_0_QS)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const float d5 = bits_h_f32(R.d);
    const uint32_t qh = R.qh;
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = (q.v[i] & 0x0F0F0F0F) | spread4(qh >> (4 * i));
      hi[i] = ((q.v[i] >> 4) & 0x0F0F0F0F) | spread4(qh >> (16 + 4 * i));
    }
    const int sumi = dot16(lo, a0) + dot16(hi, a1);
    return d5 * (sumi * A.xd[u] - 16 * A.xs[u]);
  }
};
template <> struct UnitDot<GGQ_TYPE_Q5_1> {  // vecdotq.cuh:128-158, 403-421
  static constexpr int UPB = 1;
  struct Raw { uint32_t dm, qh; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 24;
    return Raw{ld_u32(b), ld_u32(b + off::Q5_1_QH), ld_u32x4_stream(b + off::Q5_1_QS)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const uint32_t dm = R.dm, qh = R.qh;
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = (q.v[i] & 0x0F0F0F0F) | spread4(qh >> (4 * i));
      hi[i] = ((q.v[i] >> 4) & 0x0F0F0F0F) | spread4(qh >> (16 + 4 * i));
    }
    const int sumi = dot16(lo, a0) + dot16(hi, a1);
    const float d5d8 = (float)(bits_h(dm & 0xFFFF) * (_Float16)A.xd[u]);
    const float m5s8 = (float)(bits_h(dm >> 16) * (_Float16)A.xs[u]);
    return sumi * d5d8 + m5s8;
  }
};
template <> struct UnitDot<GGQ_TYPE_Q8_0> {  // vecdotq.cuh:162-174, 423-438
  static constexpr int UPB = 1;
  struct Raw { uint32_t d; u32x4_a2 q0, q1; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 34;
    return Raw{ld_u16(b), ld_u32x4_stream(b + off::Q8_0_QS), ld_u32x4_stream(b + off::Q8_0_QS + 16)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const float d = bits_h_f32(R.d);
    const u32x4_a2& q0 = R.q0;
    const u32x4_a2& q1 = R.q1;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    const int sumi = dot16(q0.v, a0) + dot16(q1.v, a1);
    return d * A.xd[u] * sumi;
  }
};

// Q2_K / Q3_K: unit = 16 bytes of qs = bytes 16c..16c+15 of half n -> for each bit pair j
// the 16 elements 128n + 32j + 16c + (0..15) of q8 group 4n+j.
template <> struct UnitDot<GGQ_TYPE_Q2_K> {  // vecdotq.cuh:195-223, 440-462
  static constexpr int UPB = 4;
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) {
    const int ib = u >> 2, n = (u >> 1) & 1, c = u & 1;
    const uint8_t* b = row + (int64_t)ib * 84;
    const u32x4_a2 q = ld_u32x4(b + off::Q2_K_QS + 32 * n + 16 * c);
    const u32x2_a2 sc8 = ld_u32x2(b + off::Q2_K_SC + 8 * n);  // scales of this half
    const uint32_t dm = ld_u32(b + off::Q2_K_D);
    float sumf_d = 0.0f, sumf_m = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = 8 * ib + 4 * n + j;  // q8 group
      const int sidx = 2 * j + c;        // scale byte within the half
      const int sc = (sc8.v[sidx >> 2] >> (8 * (sidx & 3))) & 0xFF;
      const v4i a = lds_ld16(A.xq + 32 * g + 16 * c);
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (q.v[i] >> (2 * j)) & 0x03030303;
      const float d8 = A.xd[g];
      sumf_d += d8 * (float)(dot16(v, a) * (sc & 0xF));
      sumf_m += d8 * (float)(A.xi16[2 * g + c] * (sc >> 4));
    }
    return bits_h_f32(dm & 0xFFFF) * sumf_d - bits_h_f32(dm >> 16) * sumf_m;
  }
};
template <> struct UnitDot<GGQ_TYPE_Q3_K> {  // vecdotq.cuh:227-260, 464-490
  static constexpr int UPB = 4;
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) {
    const int ib = u >> 2, n = (u >> 1) & 1, c = u & 1;
    const uint8_t* b = row + (int64_t)ib * 110;
    const u32x4_a2 q = ld_u32x4(b + off::Q3_K_QS + 32 * n + 16 * c);
    const u32x4_a2 hm = ld_u32x4(b + off::Q3_K_HM + 16 * c);
    const u32x3_a2 s = ld_u32x3(b + off::Q3_K_SC);
    const float d = bits_h_f32(ld_u16(b + off::Q3_K_D));
    float sumf = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = 8 * ib + 4 * n + j;
      const int sc = q3k_scale(s.v[0], s.v[1], s.v[2], 8 * n + 2 * j + c);
      const v4i a = lds_ld16(A.xq + 32 * g + 16 * c);
      uint32_t v[4], nb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = (q.v[i] >> (2 * j)) & 0x03030303;
        nb[i] = ((~hm.v[i]) >> (4 * n + j)) & 0x01010101;  // 1 where the mask bit is clear
      }
      const int dot = dot16(v, a) - 4 * dot16(nb, a);  // Σ (q2 - 4·¬h) q8
      sumf += A.xd[g] * (float)(dot * sc);
    }
    return d * sumf;
  }
};

// Q4_K / Q5_K: unit = 16 bytes of qs = bytes 16hf..16hf+15 of segment il -> 16 elements of
// group 2il (low nibbles) and 16 of group 2il+1 (high nibbles), positions 16hf..16hf+15.
template <> struct UnitDot<GGQ_TYPE_Q4_K> {  // vecdotq.cuh:264-291, 492-537
  static constexpr int UPB = 8;
  struct Raw { u32x4_a2 hd, q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)(u >> 3) * 144;
    return Raw{ld_u32x4(b), ld_u32x4(b + off::Q4_K_QS + 32 * ((u >> 1) & 3) + 16 * (u & 1))};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const int ib = u >> 3, il = (u >> 1) & 3, hf = u & 1;
    const u32x4_a2& hd = R.hd;
    const u32x4_a2& q = R.q;
    float sumf_d = 0.0f, sumf_m = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gl = 2 * il + i, g = 8 * ib + gl;
      int sc, mn;
      k4_scale_min(hd.v[1], hd.v[2], hd.v[3], gl, sc, mn);
      const v4i a = lds_ld16(A.xq + 32 * g + 16 * hf);
      uint32_t v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (q.v[e] >> (4 * i)) & 0x0F0F0F0F;
      const float d8 = A.xd[g];
      sumf_d += d8 * (float)(dot16(v, a) * sc);
      sumf_m += d8 * (float)(A.xi16[2 * g + hf] * mn);
    }
    return bits_h_f32(hd.v[0] & 0xFFFF) * sumf_d - bits_h_f32(hd.v[0] >> 16) * sumf_m;
  }
};
template <> struct UnitDot<GGQ_TYPE_Q5_K> {  // vecdotq.cuh:295-323, 539-585
  static constexpr int UPB = 8;
  struct Raw { u32x4_a2 hd, q, qh; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)(u >> 3) * 176;
    return Raw{ld_u32x4(b), ld_u32x4(b + off::Q5_K_QS + 32 * ((u >> 1) & 3) + 16 * (u & 1)), ld_u32x4(b + off::Q5_K_QH + 16 * (u & 1))};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const int ib = u >> 3, il = (u >> 1) & 3, hf = u & 1;
    const u32x4_a2& hd = R.hd;
    const u32x4_a2& q = R.q;
    const u32x4_a2& qh = R.qh;
    float sumf_d = 0.0f, sumf_m = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gl = 2 * il + i, g = 8 * ib + gl;
      int sc, mn;
      k4_scale_min(hd.v[1], hd.v[2], hd.v[3], gl, sc, mn);
      const v4i a = lds_ld16(A.xq + 32 * g + 16 * hf);
      uint32_t v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = ((q.v[e] >> (4 * i)) & 0x0F0F0F0F) | (((qh.v[e] >> gl) & 0x01010101) << 4);
      const float d8 = A.xd[g];
      sumf_d += d8 * (float)(dot16(v, a) * sc);
      sumf_m += d8 * (float)(A.xi16[2 * g + hf] * mn);
    }
    return bits_h_f32(hd.v[0] & 0xFFFF) * sumf_d - bits_h_f32(hd.v[0] >> 16) * sumf_m;
  }
};
// Q6_K: unit = 16 bytes of ql = bytes 16c..16c+15 (c = 0..3) of half ip -> low nibbles: 16
// elements of group 4ip + c/2, high nibbles: 16 elements of group 4ip + 2 + c/2, positions 16(c&1)..
template <> struct UnitDot<GGQ_TYPE_Q6_K> {  // vecdotq.cuh:327-345, 587-605
  static constexpr int UPB = 8;
  struct Raw { u32x4_a2 ql, qh; uint32_t d; int sc[2]; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const int ib = u >> 3, ip = (u >> 2) & 1, c = u & 3;
    const uint8_t* b = row + (int64_t)ib * 210;
    Raw R;
    R.ql = ld_u32x4(b + off::Q6_K_QL + 64 * ip + 16 * c);
    R.qh = ld_u32x4(b + off::Q6_K_QH + 32 * ip + 16 * (c & 1));
    R.d = ld_u16(b + off::Q6_K_D);
#pragma unroll
    for (int i = 0; i < 2; ++i) R.sc[i] = (int8_t)b[off::Q6_K_SC + 8 * ip + 2 * ((c >> 1) + 2 * i) + (c & 1)];
    return R;
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const int ib = u >> 3, ip = (u >> 2) & 1, c = u & 3;
    const u32x4_a2& ql = R.ql;
    const u32x4_a2& qh = R.qh;
    const float d = bits_h_f32(R.d);
    float sumf = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = (c >> 1) + 2 * i;          // 32-element group inside the half
      const int g = 8 * ib + 4 * ip + j;
      const int sc = R.sc[i];
      const v4i a = lds_ld16(A.xq + 32 * g + 16 * (c & 1));
      uint32_t v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = ((ql.v[e] >> (4 * i)) & 0x0F0F0F0F) | (((qh.v[e] >> (2 * j)) & 0x03030303) << 4);
      const int dot = dot16(v, a) - 32 * A.xi16[2 * g + (c & 1)];  // Σ (q6 - 32) q8
      sumf += A.xd[g] * (float)(dot * sc);
    }
    return d * sumf;
  }
};

// IQ4_NL / IQ4_XS (vecdotq.cuh:842-888): the nibbles index the 16-entry int8 codebook; unit = the 16 quant bytes
// of one 32-element (sub-)block: low nibbles x q8[0..15], high nibbles x q8[16..31]; float part d · (ls - 32) · d8.
template <> struct UnitDot<GGQ_TYPE_IQ4_NL> {
  static constexpr int UPB = 1;
  struct Raw { uint32_t d; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)u * 18;
    return Raw{ld_u16(b), ld_u32x4_stream(b + off::IQ4_NL_QS)};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * u), a1 = lds_ld16(A.xq + 32 * u + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = iq4nl_lookup4(q.v[i] & 0x0F0F0F0F); hi[i] = iq4nl_lookup4((q.v[i] >> 4) & 0x0F0F0F0F); }
    const float d = bits_h_f32(R.d) * A.xd[u];
    return d * (float)(dot16(lo, a0) + dot16(hi, a1));
  }
};
template <> struct UnitDot<GGQ_TYPE_IQ4_XS> {
  static constexpr int UPB = 8;
  struct Raw { u32x2_a2 hd; u32x4_a2 q; };
  static __device__ __forceinline__ Raw load(const uint8_t* row, int u) {
    const uint8_t* b = row + (int64_t)(u >> 3) * 136;
    return Raw{ld_u32x2(b), ld_u32x4(b + off::IQ4_XS_QS + 16 * (u & 7))};
  }
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) { return dot(load(row, u), u, A); }
  static __device__ __forceinline__ float dot(const Raw& R, int u, const ActLds& A) {
    const int ib32 = u & 7, g = u;   // group g = 8 ib + ib32
    const u32x2_a2& hd = R.hd;
    const u32x4_a2& q = R.q;
    const v4i a0 = lds_ld16(A.xq + 32 * g), a1 = lds_ld16(A.xq + 32 * g + 16);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = iq4nl_lookup4(q.v[i] & 0x0F0F0F0F); hi[i] = iq4nl_lookup4((q.v[i] >> 4) & 0x0F0F0F0F); }
    const float d = bits_h_f32(hd.v[0] & 0xFFFF) * (float)iq4xs_scale(hd.v[0] >> 16, hd.v[1], ib32) * A.xd[g];
    return d * (float)(dot16(lo, a0) + dot16(hi, a1));
  }
};

// The grid-codebook IQ formats (vecdotq.cuh:607-826, launchers mmvq.cuh:130-209: vdr 1, one 32-element sub-block per
// lane): unit u = sub-block u & 7 of super-block u >> 3.  Signed grid bytes x q8 with v_dot4 (exact), then the reference's
// float expression for the format.
__device__ __forceinline__ int iq_dot8(uint32_t lo, uint32_t hi, const int8_t* a8) {
  const v2i a = *(const v2i*)a8;
  return sdot4((int)hi, a[1], sdot4((int)lo, a[0], 0));
}
template <int T> struct IqUnitDot {   // IQ2_XXS, IQ3_XXS, IQ3_S: one scale per sub-block; IQ2_XS, IQ2_S: one per 16 elements
  static constexpr int UPB = 8;
  static constexpr bool split = T == GGQ_TYPE_IQ2_XS || T == GGQ_TYPE_IQ2_S;
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) {
    const uint8_t* b = row + (int64_t)(u >> 3) * Fmt<T>::BS;
    const int ib = u & 7;
    int sumi[2] = {0, 0};
    float mul[2] = {0.0f, 0.0f};
#pragma unroll
    for (int il = 0; il < 4; ++il) {
      uint32_t lo, hi;
      float m;
      IqRun<T>::get(A.grid, b, ib, il, lo, hi, m);
      sumi[il >> 1] += iq_dot8(lo, hi, A.xq + 32 * u + 8 * il);
      mul[il >> 1] = m;
    }
    const float dh = bits_h_f32(ld_u16(b));
    if constexpr (split) {   // d * ((0.5f + ls1) * sumi1 + (0.5f + ls2) * sumi2), d = half2float(d) * d8 * 0.25f
      const float d = dh * A.xd[u] * IqRun<T>::post;
      return d * (mul[0] * (float)sumi[0] + mul[1] * (float)sumi[1]);
    } else {                 // d * sumi, d = half2float(d) * (0.5f + scale) * d8 * post
      const float d = dh * mul[0] * A.xd[u] * IqRun<T>::post;
      return d * (float)(sumi[0] + sumi[1]);
    }
  }
};
template <> struct UnitDot<GGQ_TYPE_IQ2_XXS> : IqUnitDot<GGQ_TYPE_IQ2_XXS> {};
template <> struct UnitDot<GGQ_TYPE_IQ2_XS> : IqUnitDot<GGQ_TYPE_IQ2_XS> {};
template <> struct UnitDot<GGQ_TYPE_IQ2_S> : IqUnitDot<GGQ_TYPE_IQ2_S> {};
template <> struct UnitDot<GGQ_TYPE_IQ3_XXS> : IqUnitDot<GGQ_TYPE_IQ3_XXS> {};
template <> struct UnitDot<GGQ_TYPE_IQ3_S> : IqUnitDot<GGQ_TYPE_IQ3_S> {};

template <> struct UnitDot<GGQ_TYPE_IQ1_S> {   // vecdotq.cuh:750-781
  static constexpr int UPB = 8;
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) {
    const uint8_t* b = row + (int64_t)(u >> 3) * 50;
    const int ib = u & 7;
    const uint32_t qh = ld_u16(b + off::IQ1_S_QH + 2 * ib);
    const uint32_t qs = ld_u32(b + off::IQ1_S_QS + 4 * ib);
    int sumi = 0;
#pragma unroll
    for (int il = 0; il < 4; ++il) {
      uint32_t lo, hi;
      iq1_grid(A.grid, ((qs >> (8 * il)) & 0xFF) | (((qh >> (3 * il)) & 7) << 8), lo, hi);
      sumi += iq_dot8(lo, hi, A.xq + 32 * u + 8 * il);
    }
    const float d1q = bits_h_f32(ld_u16(b)) * (float)(((qh >> 11) & 0x0E) + 1);
    const float delta = -1.0f + IQ1_DELTA - (float)(qh & 0x8000) * (2.0f * IQ1_DELTA / 0x8000);
    return d1q * (A.xd[u] * (float)sumi + A.xs[u] * delta);
  }
};
template <> struct UnitDot<GGQ_TYPE_IQ1_M> {   // vecdotq.cuh:783-826
  static constexpr int UPB = 8;
  static __device__ __forceinline__ float run(const uint8_t* row, int u, const ActLds& A) {
    const uint8_t* b = row + (int64_t)(u >> 3) * 56;
    const int ib = u & 7;
    const uint32_t qs = ld_u32(b + off::IQ1_M_QS + 4 * ib);
    const uint32_t qh2 = ld_u16(b + off::IQ1_M_QH + 2 * ib);
    int sumi[2] = {0, 0};
    float sumf[2] = {0.0f, 0.0f};
#pragma unroll
    for (int il = 0; il < 4; ++il) {
      const uint32_t qhl = (qh2 >> (8 * (il >> 1))) >> (4 * (il & 1));
      uint32_t lo, hi;
      iq1_grid(A.grid, ((qs >> (8 * il)) & 0xFF) | ((qhl & 7) << 8), lo, hi);
      const v2i a = *(const v2i*)(A.xq + 32 * u + 8 * il);
      sumi[il >> 1] = sdot4((int)hi, a[1], sdot4((int)lo, a[0], sumi[il >> 1]));
      const float delta = -1.0f + IQ1_DELTA - (float)(qhl & 0x08) * (2.0f * IQ1_DELTA / 0x08);
      const int sumy = sdot4(0x01010101, a[1], sdot4(0x01010101, a[0], 0));
      sumf[il >> 1] += delta * (float)sumy;
    }
    const float d = iq1m_super_scale(b) * A.xd[u];
    const uint32_t tmp = ld_u16(b + off::IQ1_M_SC + 2 * (ib >> 1)) >> (6 * (ib & 1));
    const int sc0 = 2 * (int)(tmp & 7) + 1, sc1 = 2 * (int)((tmp >> 3) & 7) + 1;
    return d * (((float)sumi[0] + sumf[0]) * (float)sc0 + ((float)sumi[1] + sumf[1]) * (float)sc1);
  }
};

// LDS bytes for a row of k activations: int8[k] + float[k/32]*2 + int[k/16]
// x as int8 + d8 / s8 floats + per-16 integer sums (k % 32 == 0: a multiple of 16 bytes), then the IQ codebook
static inline size_t mmvq_lds_bytes(int64_t k, int grid_bytes = 0) { return (size_t)k + (size_t)(k / 32) * 8 + (size_t)(k / 16) * 4 + (size_t)grid_bytes; }

// FUSED: q8 is the activation row itself (dtype DT); every workgroup quantises it into LDS with the
// arithmetic of quantize.hip (bit-identical d, q, sum) while its first weight bytes are in flight —
// one launch instead of two (the second launch cost 2.3 of 8.8 us at the headline shape).
template <int T, int DT, int ROWS, bool FUSED, bool KSPLIT>
__global__ void __launch_bounds__(FUSED ? 1024 : 256) mmvq_kernel(const uint8_t* __restrict__ w,
                                                   const uint8_t* __restrict__ q8,
                                                   void* __restrict__ y, int k, int n_rows,
                                                   int rows_per_wave, GatherOut go) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  int8_t* xq = (int8_t*)lds;
  float* xd = (float*)(lds + k);
  float* xs = xd + k / 32;
  int* xi16 = (int*)(xs + k / 32);
  const int64_t row_bytes = (int64_t)(k / Fmt<T>::QK) * Fmt<T>::BS;

  VSTAMP(0);
  // ---- fused kernel, formats with a load / dot split: the weight bytes of the wave's first ROWS rows (first 64-unit
  //      step of each) are requested right after the x loads and before x is quantised — their HBM round trip overlaps
  //      the 1.2 us prologue.  Q4_0 7.6 -> 7.1 us warm / 10.0 -> 9.25 cold, Q4_K 8.7 -> 8.2 / 10.5 -> 9.8, Q8_0 11.2 -> 9.9 /
  //      15.1 -> 13.5; issued BEFORE the x loads it is a loss (x queues behind the HBM misses), two steps too. ----
  constexpr bool PRE = FUSED && UnitHasPre<T>::value;
  constexpr int PS = 1;   // prefetched 64-unit steps per row (Q4_K: two 16-byte loads per unit; two steps spill at 128 VGPRs)
  struct NoRaw {};
  using RawT = typename std::conditional<PRE, typename UnitDot<PRE ? T : GGQ_TYPE_Q4_0>::Raw, NoRaw>::type;
  RawT pre[ROWS][PS];
  if constexpr (FUSED) {
    // ---- quantise x -> Q8_1 in LDS (fused launches use 16-wave workgroups, one per CU, and split the rows evenly over
    //      all waves: every workgroup repeats the quantisation, so there must be few of them).  A byte-per-line
    //      "touch" of the wave's weight span before this prologue used to be here: with this prologue it costs
    //      0.7 us warm and 1.1 - 1.7 us cold (every line goes through the texture path twice), so it is gone. ----
    // Lane l (3 bits b2 b1 b0) of an 8-lane group takes the 4-element chunk c = (b2, b1^b2, b0^b2) of its 32-group: the
    // three butterfly levels of quantize.hip's sum (element ^16, ^8, ^4 — the fp32 order of the reference's warp
    // reduction) are then row_half_mirror, quad_perm[2,3,0,1] and quad_perm[1,0,3,2]: DPP modifiers of the add itself
    // instead of 15 ds_bpermute round trips through the LDS pipe, and the four lanes of a quad hold one 16-element
    // half, so the per-16 integer sums fall out of the same pass (no second pass, no second barrier).
    const int l8 = threadIdx.x & 7;
    const int chunk = ((l8 >> 2) * 7) ^ (l8 & 3);
    const int g0 = threadIdx.x >> 3;
    float v[4];
    if (g0 < k / 32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = Elem<DT>::ld(q8, g0 * 32 + chunk * 4 + i);
    }
    // (after the x loads: vector loads return in order, and x must not queue behind this wave's HBM misses)
    if constexpr (PRE) {
      const int n_waves = gridDim.x * 16, wave0 = blockIdx.x * 16 + (threadIdx.x >> 6);
      const int rb = (int)((int64_t)wave0 * n_rows / n_waves);
      const int units0 = k / Fmt<T>::QK * UnitDot<T>::UPB;
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int sp = 0; sp < PS; ++sp)   // clamped, never predicated: rows / units past the end repeat valid bytes
          pre[r][sp] = KSPLIT   // K-split launch (see the row loop): ROWS K-slices of the wave's one row
                           ? UnitDot<T>::load(w + (int64_t)min(rb, n_rows - 1) * row_bytes, min((int)(threadIdx.x & 63) + 64 * (sp * ROWS + r), units0 - 1))
                           : UnitDot<T>::load(w + (int64_t)min(rb + r, n_rows - 1) * row_bytes, min((int)(threadIdx.x & 63) + 64 * sp, units0 - 1));
    }
    auto dppf = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false)); };
    auto dppi = [](int x, auto ctrl) { return __builtin_amdgcn_update_dpp(0, x, decltype(ctrl)::value, 0xF, 0xF, false); };
    using HM = std::integral_constant<int, 0x141>;   // row_half_mirror: lane <-> 7 - lane
    using Q2 = std::integral_constant<int, 0x4E>;    // quad_perm [2,3,0,1]: lane ^ 2
    using Q1 = std::integral_constant<int, 0xB1>;    // quad_perm [1,0,3,2]: lane ^ 1
    for (int g = g0; g < k / 32; g += 128) {   // k % 32 == 0: a group is never split
      const int ix = g * 32 + chunk * 4;
      if (g != g0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = Elem<DT>::ld(q8, ix + i);
      }
      float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
      amax = fmaxf(amax, dppf(amax, HM{}));
      amax = fmaxf(amax, dppf(amax, Q2{}));
      amax = fmaxf(amax, dppf(amax, Q1{}));
      float sm[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) sm[i] = sm[i] + dppf(sm[i], HM{});   // element ^ 16
#pragma unroll
      for (int i = 0; i < 4; ++i) sm[i] = sm[i] + dppf(sm[i], Q2{});   // element ^ 8
#pragma unroll
      for (int i = 0; i < 4; ++i) sm[i] = sm[i] + dppf(sm[i], Q1{});   // element ^ 4
      const float sum = (sm[0] + sm[2]) + (sm[1] + sm[3]);
      const float d = amax / 127;
      int qi[4];
      quant4(v, amax, d, qi);   // (ggq_common.h: the exact division only for quotients near a rounding boundary)
      ((uint32_t*)xq)[ix >> 2] = (uint32_t)(qi[0] & 0xFF) | ((uint32_t)(qi[1] & 0xFF) << 8) |
                                 ((uint32_t)(qi[2] & 0xFF) << 16) | ((uint32_t)(qi[3] & 0xFF) << 24);
      int s16 = (qi[0] + qi[1]) + (qi[2] + qi[3]);   // exact: the quad's four chunks are one 16-element half
      s16 += dppi(s16, Q2{});
      s16 += dppi(s16, Q1{});
      if ((l8 & 3) == 0) xi16[2 * g + (l8 >> 2)] = s16;
      if (l8 == 0) {   // block_q8_1 stores half d, half sum: keep their fp16 rounding
        xd[g] = (float)(_Float16)d;
        xs[g] = (float)(_Float16)sum;
      }
    }
  } else {
    // ---- stage the Q8_1 row: block_q8_1 {half d, half s, int8 qs[32]} -> de-interleaved ----
    for (int i = threadIdx.x; i < k / 4; i += 256) {
      const int g = i >> 3, j = i & 7;
      ((uint32_t*)xq)[i] = *(const uint32_t*)(q8 + (int64_t)g * 36 + 4 + 4 * j);
    }
    for (int g = threadIdx.x; g < k / 32; g += 256) {
      const uint32_t ds = *(const uint32_t*)(q8 + (int64_t)g * 36);
      xd[g] = bits_h_f32(ds & 0xFFFF);
      xs[g] = bits_h_f32(ds >> 16);
    }
  }
  if constexpr (IqGrid<T>::BYTES != 0) {   // the codebook: 1 - 8 KB, 16 bytes per thread and pass
    v4i* lg = (v4i*)(xi16 + k / 16);
    const v4i* gg = (const v4i*)IqGrid<T>::table();
    for (int i = threadIdx.x; i < IqGrid<T>::BYTES / 16; i += FUSED ? 1024 : 256) lg[i] = gg[i];
  }
  VSTAMP(1);
  __syncthreads();
  VSTAMP(2);
  if constexpr (!FUSED) {
    for (int i = threadIdx.x; i < k / 16; i += 256) {
      const v4i a = *(const v4i*)(xq + 16 * i);
      int s = sdot4(0x01010101, a[0], 0);
      s = sdot4(0x01010101, a[1], s);
      s = sdot4(0x01010101, a[2], s);
      xi16[i] = sdot4(0x01010101, a[3], s);
    }
    __syncthreads();
  }
  const ActLds A{xq, xd, xs, xi16, IqGrid<T>::BYTES ? (const void*)(xi16 + k / 16) : nullptr};

  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (FUSED ? 16 : 4) + (threadIdx.x >> 6);
  const int units = k / Fmt<T>::QK * UnitDot<T>::UPB;
  int row_begin, row_end;
  if constexpr (FUSED) {
    const int n_waves = gridDim.x * 16;
    row_begin = (int)((int64_t)wave * n_rows / n_waves);
    row_end = (int)((int64_t)(wave + 1) * n_rows / n_waves);
  } else {
    row_begin = wave * rows_per_wave;
    row_end = min(n_rows, (wave + 1) * rows_per_wave);
  }
  VSTAMP(3);

  for (int r0 = row_begin; r0 < row_end; r0 += ROWS) {
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;
    int u_first = lane;
    if constexpr (PRE) {
      if (!KSPLIT && r0 == row_begin) {   // wave-uniform: the prefetched steps of the first row group
#pragma unroll
        for (int sp = 0; sp < PS; ++sp) {
          const int u = lane + 64 * sp;
          if (u < units) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
              if (r0 + r < row_end) acc[r] += UnitDot<T>::dot(pre[r][sp], u, A);
          }
        }
        u_first = lane + 64 * PS;
      }
    }
    if constexpr (FUSED && KSPLIT) {
      // Long rows, few of them (n_rows <= the launch's wave count, e.g. the 4096 x 11008 down projection: one row per wave):
      // the ROWS in-flight slots become K-slices of the wave's one row, so that a lane still has ROWS independent loads
      // outstanding.  A separate instantiation chosen per launch, not per row group: every row of the matrix is summed in the same
      // order, and the rows-in-flight form keeps its register budget (a runtime flag cost Q4_K / Q6_K 53 spilled registers).
      {
        if constexpr (PRE) {
#pragma unroll
          for (int sp = 0; sp < PS; ++sp)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              const int u = lane + 64 * (sp * ROWS + r);   // clamped + selected, never predicated: no divergent control flow
              const float d = UnitDot<T>::dot(pre[r][sp], min(u, units - 1), A);
              acc[r] += u < units ? d : 0.0f;
            }
          u_first = lane + 64 * ROWS * PS;
        }
        for (int u = u_first; u < units; u += 64 * ROWS) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float d = UnitDot<T>::run(w + (int64_t)r0 * row_bytes, min(u + 64 * r, units - 1), A);
            acc[r] += u + 64 * r < units ? d : 0.0f;
          }
        }
        float tot = acc[0];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) tot += acc[r];
        tot = wave_sum(tot);
        if (lane == 0) {
          Elem<DT>::st(y, r0, tot);
          gather_store<DT>(go, r0, tot);
        }
        continue;
      }
    }
#pragma unroll GGQ_MMVQ_UNROLL
    for (int u = u_first; u < units; u += 64) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if constexpr (FUSED) {   // wave-uniform: a short last group simply skips the row
          if (r0 + r < row_end) acc[r] += UnitDot<T>::run(w + (int64_t)(r0 + r) * row_bytes, u, A);
        } else {
          const int row = min(r0 + r, n_rows - 1);  // clamp: duplicate work, never out of bounds
          acc[r] += UnitDot<T>::run(w + row * row_bytes, u, A);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float tot = wave_sum(acc[r]);
      if (lane == 0 && r0 + r < row_end) {
        Elem<DT>::st(y, r0 + r, tot);
        gather_store<DT>(go, r0 + r, tot);
      }
    }
  }
  VSTAMP(4);
  if (go.n_flag > 0) {   // (kernel-uniform) multi-destination launch: every wave drains its own write-through stores and arrives;
    // the last arrival of the launch writes the flags (the one release at system scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const uint32_t before = __hip_atomic_fetch_add(go.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == gridDim.x * (blockDim.x >> 6) - 1u) {
        __hip_atomic_store(go.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = 0; d < go.n_flag; ++d) __hip_atomic_store(go.flag[d], go.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

template <int T, int DT, bool FUSED>
static int launch_mmvq_t(const void* w, const void* q8, void* y, int64_t k, int64_t n, hipStream_t s, const GatherOut& go) {
  // rows in flight per wave: three in the fused kernel (a wave owns 2.7 rows at 11008 rows on 4096 waves: all of them go
  // out at once — Q4_0 10.7 -> 10.0 us cold, 7.9 -> 7.7 warm), two for Q8_0 (34-byte blocks: 15.3 vs 15.8 us cold)
  constexpr int ROWS = !FUSED ? 2 : (T == GGQ_TYPE_Q8_0 ? 2 : GGQ_MMVQ_ROWS);
  const size_t lds = mmvq_lds_bytes(k, IqGrid<T>::BYTES);
  if (lds > 160 * 1024) return GGQ_ERR_SHAPE;
  // many short-lived waves keep more weight bytes in flight (measured: 2 rows per wave beats
  // 4-8 at N = 11008); rows_per_wave = n / 8192, even, in [2, 16]
  int rpw = (int)(n / 8192);
  rpw = rpw < ROWS ? ROWS : (rpw > 16 ? 16 : rpw);
  rpw = (rpw + ROWS - 1) / ROWS * ROWS;
  const int64_t waves = (n + rpw - 1) / rpw;
  int64_t grid = (waves + 3) / 4;
  if (FUSED) {   // one 16-wave workgroup per CU (fewer when there is less than one row per wave)
    grid = (n + 15) / 16;   // at least one row per wave before a CU is left without a workgroup
    grid = grid > 256 ? 256 : grid;
  }
  const bool ksplit = GGQ_MMVQ_KSPLIT && FUSED && n <= grid * 16;   // at most one row per wave: K-split over the in-flight slots (see the kernel)
  auto kern = ksplit ? mmvq_kernel<T, DT, ROWS, FUSED, FUSED> : mmvq_kernel<T, DT, ROWS, FUSED, false>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GGQ_ERR_LAUNCH;
  }
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FUSED ? 1024 : 256), lds, s, (const uint8_t*)w,
                     (const uint8_t*)q8, y, (int)k, (int)n, rpw, go);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T>
static int launch_mmvq(const void* w, const void* q8, void* y, int dt, int64_t k, int64_t n, bool fused, hipStream_t s, const GatherOut& go) {
  switch (dt) {
    case GGQ_F32: return fused ? launch_mmvq_t<T, GGQ_F32, true>(w, q8, y, k, n, s, go) : launch_mmvq_t<T, GGQ_F32, false>(w, q8, y, k, n, s, go);
    case GGQ_F16: return fused ? launch_mmvq_t<T, GGQ_F16, true>(w, q8, y, k, n, s, go) : launch_mmvq_t<T, GGQ_F16, false>(w, q8, y, k, n, s, go);
    case GGQ_BF16: return fused ? launch_mmvq_t<T, GGQ_BF16, true>(w, q8, y, k, n, s, go) : launch_mmvq_t<T, GGQ_BF16, false>(w, q8, y, k, n, s, go);
    default: return GGQ_ERR_DTYPE;
  }
}

static int mmvq_dispatch(const void* w, const void* q, void* y, int type, int dtype, int64_t k, int64_t n_rows,
                         bool fused, hipStream_t s, const GatherOut& go = GatherOut{}) {
  switch (type) {
    case GGQ_TYPE_Q4_0: return launch_mmvq<GGQ_TYPE_Q4_0>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q4_1: return launch_mmvq<GGQ_TYPE_Q4_1>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q5_0: return launch_mmvq<GGQ_TYPE_Q5_0>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q5_1: return launch_mmvq<GGQ_TYPE_Q5_1>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q8_0: return launch_mmvq<GGQ_TYPE_Q8_0>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q2_K: return launch_mmvq<GGQ_TYPE_Q2_K>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q3_K: return launch_mmvq<GGQ_TYPE_Q3_K>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q4_K: return launch_mmvq<GGQ_TYPE_Q4_K>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q5_K: return launch_mmvq<GGQ_TYPE_Q5_K>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_Q6_K: return launch_mmvq<GGQ_TYPE_Q6_K>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ4_NL: return launch_mmvq<GGQ_TYPE_IQ4_NL>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ4_XS: return launch_mmvq<GGQ_TYPE_IQ4_XS>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ2_XXS: return launch_mmvq<GGQ_TYPE_IQ2_XXS>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ2_XS: return launch_mmvq<GGQ_TYPE_IQ2_XS>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ2_S: return launch_mmvq<GGQ_TYPE_IQ2_S>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ3_XXS: return launch_mmvq<GGQ_TYPE_IQ3_XXS>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ3_S: return launch_mmvq<GGQ_TYPE_IQ3_S>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ1_S: return launch_mmvq<GGQ_TYPE_IQ1_S>(w, q, y, dtype, k, n_rows, fused, s, go);
    case GGQ_TYPE_IQ1_M: return launch_mmvq<GGQ_TYPE_IQ1_M>(w, q, y, dtype, k, n_rows, fused, s, go);
    default: return GGQ_ERR_TYPE;
  }
}

}  // namespace ggq

extern "C" int ggq_mul_mat_vec_q_prequant(const void* w, const void* q, void* y, int type, int dtype,
                                          int64_t k, int64_t n_rows, void* stream) {
  using namespace ggq;
  if (k <= 0 || n_rows < 0) return GGQ_ERR_ARG;
  if (!ggq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (k > 0x7fffffffLL / 64 || n_rows > 0x7fffffffLL) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0) return GGQ_OK;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 3)) return GGQ_ERR_ALIGN;
  return mmvq_dispatch(w, q, y, type, dtype, k, n_rows, false, (hipStream_t)stream);
}

extern "C" int ggq_mul_mat_vec_q(const void* w, const void* x, void* y, int type, int dtype,
                                 int64_t k, int64_t n_rows, void* scratch, void* stream) {
  using namespace ggq;
  if (!scratch) return GGQ_ERR_ARG;   // kept in the signature (reference: quant_X), unused by the fused kernel
  if (k <= 0 || n_rows < 0) return GGQ_ERR_ARG;
  if (!ggq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (k > 0x7fffffffLL / 64 || n_rows > 0x7fffffffLL) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0) return GGQ_OK;
  if (!w || !x || !y) return GGQ_ERR_ARG;
  if ((uintptr_t)w & 1) return GGQ_ERR_ALIGN;
  static const char* e = GGQ_TUNING_ENV("GGQ_MMVQ_FUSED");   // 0: two launches (quantize_q8_1 + mul_mat_vec_q), for comparison
  if (e && e[0] == '0') {
    const int rc = ggq_quantize_q8_1(x, dtype, scratch, 1, k, stream);
    if (rc != GGQ_OK) return rc;
    return ggq_mul_mat_vec_q_prequant(w, scratch, y, type, dtype, k, n_rows, stream);
  }
  return mmvq_dispatch(w, x, y, type, dtype, k, n_rows, true, (hipStream_t)stream);
}

// The fused GEMV with the multi-destination write-back (the batch-1 side of ggq_mul_mat_q_gather): Y (n_rows elements) goes to
// dsts[0 .. n_dst), and once every wave of the launch has drained its stores `generation` is written into flags[0 .. n_flag).
extern "C" int ggq_mul_mat_vec_q_gather(const void* w, const void* x, void* const* dsts, int n_dst, void* const* flags, int n_flag,
                                        uint32_t generation, void* arrivals, int type, int dtype, int64_t k, int64_t n_rows,
                                        void* stream) {
  using namespace ggq;
  if (n_dst < 1 || n_dst > 8 || n_flag < 0 || n_flag > 8 || !dsts || (n_flag && (!flags || !arrivals))) return GGQ_ERR_ARG;
  if (k <= 0 || n_rows < 0) return GGQ_ERR_ARG;
  if (!ggq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (k > 0x7fffffffLL / 64 || n_rows > 0x7fffffffLL) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0) return GGQ_ERR_SHAPE;   // nothing would publish the flags
  if (!w || !x) return GGQ_ERR_ARG;
  if ((uintptr_t)w & 1) return GGQ_ERR_ALIGN;
  GatherOut go{};
  const uintptr_t emask = dtype == GGQ_F32 ? 3 : 1;
  for (int d = 0; d < n_dst; ++d) {
    if (!dsts[d]) return GGQ_ERR_ARG;
    if ((uintptr_t)dsts[d] & emask) return GGQ_ERR_ALIGN;
    go.dst[d] = dsts[d];
  }
  for (int d = 0; d < n_flag; ++d) {
    if (!flags[d] || ((uintptr_t)flags[d] & 3)) return GGQ_ERR_ARG;
    go.flag[d] = (uint32_t*)flags[d];
  }
  go.arrivals = (uint32_t*)arrivals;
  go.generation = generation;
  go.n_dst = n_dst;
  go.n_flag = n_flag;
  return mmvq_dispatch(w, x, dsts[0], type, dtype, k, n_rows, true, (hipStream_t)stream, go);
}
