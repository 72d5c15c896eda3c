// ggq_common.h — shared device helpers for the gfx950 ggml block-quant kernels.
//
// Block wire formats follow HK/ggml/ggml-common.h:17-108 of the reference
// (byte offsets restated below); everything else here is ours.
// Compile with -ffp-contract=off: the fp16 dequantise sequences must round after
// every operation exactly like the reference's __hmul/__hsub/__hadd intrinsics.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../../include/ggq.h"

namespace ggq {

constexpr int WAVE = 64;

// ---- vector types ---------------------------------------------------------
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- unaligned loads (gfx950 global/LDS accesses need no natural alignment;
//      the packed structs tell the compiler only 1/2-byte alignment is known) ----
struct __attribute__((packed, aligned(2))) u16x1_a2 { uint16_t v; };
struct __attribute__((packed, aligned(2))) u32x1_a2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) u32x2_a2 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) u32x3_a2 { uint32_t v[3]; };
struct __attribute__((packed, aligned(2))) u32x4_a2 { uint32_t v[4]; };

typedef uint32_t u32x4_al2 __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return ((const u16x1_a2*)p)->v; }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return ((const u32x1_a2*)p)->v; }
__device__ __forceinline__ u32x2_a2 ld_u32x2(const uint8_t* p) { return *(const u32x2_a2*)p; }
__device__ __forceinline__ u32x3_a2 ld_u32x3(const uint8_t* p) { return *(const u32x3_a2*)p; }
__device__ __forceinline__ u32x4_a2 ld_u32x4(const uint8_t* p) { return *(const u32x4_a2*)p; }
// the same load with the nontemporal hint, for weight bytes a kernel reads exactly ONCE per launch (the GEMV's quant
// payload): they do not displace what is re-read (block headers, activations) from L2.  Measured round 3 with the
// weights streamed from HBM (scripts/sweep_mmvq.py): Q4_0 9.33 -> 8.58 us, Q8_0 14.9 -> 13.9; replayed on ONE resident
// tensor it costs 1 - 22 % (Q8_0 9.97 -> 12.2): the hint gives up what a back-to-back replay of the same 25 - 48 MB gains
// from the Infinity Cache, which a model larger than that cache never sees.  Used for the 32-element block formats only:
// for Q4_K / Q5_K (payload-only hint, headers re-read by eight units) it measured worse in both states (9.79 -> 10.06 cold).
__device__ __forceinline__ u32x4_a2 ld_u32x4_stream(const uint8_t* p) {
  const u32x4_al2 t = __builtin_nontemporal_load((const u32x4_al2*)p);
  return u32x4_a2{{t[0], t[1], t[2], t[3]}};
}

__device__ __forceinline__ _Float16 bits_h(uint32_t b) {
  uint16_t s = (uint16_t)b;
  return __builtin_bit_cast(_Float16, s);
}
__device__ __forceinline__ float bits_h_f32(uint32_t b) { return (float)bits_h(b); }

// ---- block geometry ------------------------------------------------------
template <int T> struct Fmt;
#define GGQ_FMT(T, QK_, BS_) \
  template <> struct Fmt<T> { static constexpr int QK = QK_; static constexpr int BS = BS_; };
GGQ_FMT(GGQ_TYPE_Q4_0, 32, 18)
GGQ_FMT(GGQ_TYPE_Q4_1, 32, 20)
GGQ_FMT(GGQ_TYPE_Q5_0, 32, 22)
GGQ_FMT(GGQ_TYPE_Q5_1, 32, 24)
GGQ_FMT(GGQ_TYPE_Q8_0, 32, 34)
GGQ_FMT(GGQ_TYPE_Q2_K, 256, 84)
GGQ_FMT(GGQ_TYPE_Q3_K, 256, 110)
GGQ_FMT(GGQ_TYPE_Q4_K, 256, 144)
GGQ_FMT(GGQ_TYPE_Q5_K, 256, 176)
GGQ_FMT(GGQ_TYPE_Q6_K, 256, 210)
GGQ_FMT(GGQ_TYPE_IQ4_NL, 32, 18)
GGQ_FMT(GGQ_TYPE_IQ4_XS, 256, 136)
GGQ_FMT(GGQ_TYPE_IQ2_XXS, 256, 66)
GGQ_FMT(GGQ_TYPE_IQ2_XS, 256, 74)
GGQ_FMT(GGQ_TYPE_IQ2_S, 256, 82)
GGQ_FMT(GGQ_TYPE_IQ3_XXS, 256, 98)
GGQ_FMT(GGQ_TYPE_IQ3_S, 256, 110)
GGQ_FMT(GGQ_TYPE_IQ1_S, 256, 50)
GGQ_FMT(GGQ_TYPE_IQ1_M, 256, 56)
#undef GGQ_FMT

// byte offsets inside a block (HK/ggml/ggml-common.h:20-108)
namespace off {
constexpr int Q4_0_D = 0, Q4_0_QS = 2;
constexpr int Q4_1_D = 0, Q4_1_M = 2, Q4_1_QS = 4;
constexpr int Q5_0_D = 0, Q5_0_QH = 2, Q5_0_QS = 6;
constexpr int Q5_1_D = 0, Q5_1_M = 2, Q5_1_QH = 4, Q5_1_QS = 8;
constexpr int Q8_0_D = 0, Q8_0_QS = 2;
constexpr int Q2_K_SC = 0, Q2_K_QS = 16, Q2_K_D = 80, Q2_K_DMIN = 82;
constexpr int Q3_K_HM = 0, Q3_K_QS = 32, Q3_K_SC = 96, Q3_K_D = 108;
constexpr int Q4_K_D = 0, Q4_K_DMIN = 2, Q4_K_SC = 4, Q4_K_QS = 16;
constexpr int Q5_K_D = 0, Q5_K_DMIN = 2, Q5_K_SC = 4, Q5_K_QH = 16, Q5_K_QS = 48;
constexpr int Q6_K_QL = 0, Q6_K_QH = 128, Q6_K_SC = 192, Q6_K_D = 208;
constexpr int IQ4_NL_D = 0, IQ4_NL_QS = 2;                                // HK/ggml/ggml-common.h:179-182
constexpr int IQ4_XS_D = 0, IQ4_XS_SH = 2, IQ4_XS_SL = 4, IQ4_XS_QS = 8;   // HK/ggml/ggml-common.h:186-191
}  // namespace off

// kvalues_iq4nl (HK/ggml/ggml-common.h:1060), the 16-entry non-linear 4-bit codebook, as four little-endian dwords:
// {-127,-104,-83,-65, -49,-35,-22,-10, 1,13,25,38, 53,69,89,113}.  Four nibbles (one per byte of `idx4`, 0..15) are
// looked up at once with two v_perm_b32 (an 8-byte table half each) and a per-byte select on bit 3 — the job of
// get_int_from_table_16 (HK/ggml/vecdotq.cuh:828-840) without its sixteen byte loads.
__device__ __forceinline__ uint32_t iq4nl_lookup4(uint32_t idx4) {
  constexpr uint32_t T0 = 0xBFAD9881u, T1 = 0xF6EADDCFu, T2 = 0x26190D01u, T3 = 0x71594535u;
  const uint32_t sel = idx4 & 0x07070707u;
  const uint32_t lo = __builtin_amdgcn_perm(T1, T0, sel);   // entries 0..7
  const uint32_t hi = __builtin_amdgcn_perm(T3, T2, sel);   // entries 8..15
  const uint32_t m = ((idx4 >> 3) & 0x01010101u) * 0xFFu;   // 0xFF in the bytes whose index has bit 3 set
  return (lo & ~m) | (hi & m);
}
// the 6-bit scale of 32-element sub-block ib of an IQ4_XS super-block (dequantize.cuh:428, vecdotq.cuh:876), minus 32
__device__ __forceinline__ int iq4xs_scale(uint32_t scales_h, uint32_t scales_l, int ib) {
  return (int)(((scales_l >> (4 * ib)) & 0xF) | (((scales_h >> (2 * ib)) & 3) << 4)) - 32;
}

// 6-bit (scale, min) pair j of the 12-byte Q4_K/Q5_K scale field, from the three
// little-endian dwords s0,s1,s2 of that field (layout: HK/ggml/dequantize.cuh:154-161).
__device__ __forceinline__ void k4_scale_min(uint32_t s0, uint32_t s1, uint32_t s2, int j,
                                             int& sc, int& mn) {
  // branch-free (j is a per-lane value in every caller: a branch runs both sides anyway, and it would split the
  // scheduling region so that the loads behind it cannot be issued early)
  const int sh = 8 * (j & 3);
  const uint32_t a0 = s0 >> sh, a1 = s1 >> sh, b = (s2 >> sh) & 0xFF;
  const int lo_sc = a0 & 63, lo_mn = a1 & 63;
  const int hi_sc = (b & 0xF) | ((a0 >> 2) & 0x30);
  const int hi_mn = (b >> 4) | ((a1 >> 2) & 0x30);
  sc = j < 4 ? lo_sc : hi_sc;
  mn = j < 4 ? lo_mn : hi_mn;
}

// Q3_K 6-bit scale i (0..15) minus 32, from the dwords of the 12-byte field
// (low nibble (i/8) of byte i%8, bit pair (i/4) of byte 8+i%4: HK/ggml/dequantize.cuh:140-143).
__device__ __forceinline__ int q3k_scale(uint32_t s0, uint32_t s1, uint32_t s2, int i) {
  const int bl = i & 7;
  const uint32_t wl = bl < 4 ? s0 : s1;
  const int lo = (wl >> (8 * (bl & 3) + 4 * (i >> 3))) & 0xF;
  const int hi = (s2 >> (8 * (i & 3) + 2 * (i >> 2))) & 3;
  return (lo | (hi << 4)) - 32;
}

// ---- conversions ----------------------------------------------------------
template <int DT> struct Elem;
template <> struct Elem<GGQ_F32> {
  typedef float type;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct Elem<GGQ_F16> {
  typedef _Float16 type;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)((const _Float16*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((_Float16*)p)[i] = (_Float16)v; }
};
template <> struct Elem<GGQ_BF16> {
  typedef uint16_t type;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) {
    return __builtin_bit_cast(float, ((uint32_t)((const uint16_t*)p)[i]) << 16);
  }
  static __device__ __forceinline__ uint16_t cvt(float v) {  // RNE, NaN stays NaN
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = cvt(v); }
};

// wave-wide sum (all lanes receive the total); order is ours, the parity budget for
// the fp accumulate is 1e-3 relative (the oracle sums the same terms in a tree).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// GGQ_HIP_PRE_LAUNCH() before a launch drops a sticky error some earlier, unrelated HIP call left behind, so that
// GGQ_HIP_CHECK_LAUNCH() after it reports this launch only.
#define GGQ_HIP_PRE_LAUNCH() ((void)hipGetLastError())
#define GGQ_HIP_CHECK_LAUNCH()                          \
  do {                                                  \
    hipError_t e_ = hipGetLastError();                  \
    if (e_ != hipSuccess) return GGQ_ERR_LAUNCH;        \
  } while (0)

// Multi-destination write-back of a GEMM (ggq_mul_mat_q_gather): the output slab goes to dst[0 .. n_dst) — the caller's own slot
// first, then the same slot of every peer's gather buffer (mapped with ggq_peer_import) — and, once every workgroup of the launch
// has released its stores at system scope, `generation` is written into flag[0 .. n_flag).  n_dst = 0: a plain launch.
struct GatherOut {
  void* dst[8];
  uint32_t* flag[8];
  uint32_t* arrivals;    // zero-initialised 4-byte word of the caller's memory; the kernel leaves it zero
  uint32_t generation;
  int n_dst, n_flag;
};

// one element into every peer slot (the scalar tail of a multi-destination write-back; system-scope relaxed stores)
template <int DT>
__device__ __forceinline__ void gather_store(const GatherOut& go, int64_t idx, float v) {
  for (int d = 1; d < go.n_dst; ++d) {   // (kernel-uniform trip count; 0 iterations in a plain launch)
    if constexpr (DT == GGQ_F32) {
      __hip_atomic_store((uint32_t*)go.dst[d] + idx, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      uint16_t bits;
      Elem<DT>::st(&bits, 0, v);
      __hip_atomic_store((uint16_t*)go.dst[d] + idx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// q = (int)roundf(x / d), bit for bit, without the IEEE division where it cannot matter.  t = x * rcp(d) is within 4.6e-5 of the correctly
// rounded quotient Q (|Q| <= 127.00002; v_rcp_f32: 1 ulp, the product: half an ulp, Q itself: half an ulp), so whenever |t| is further
// than 1e-3 from every k + 0.5 both round (half away from zero) to the same integer, and trunc(t + copysign(0.5, t)) is that integer.
// Anything else — a quotient near a rounding boundary (0.2 % of the elements), NaN / inf, a scale outside 2^-100 .. 2^100 where rcp or
// the product could leave the normal range — takes the exact division; the branch is per lane, skipped by waves without such a lane.
// The division + roundf were ~60 % of the kernel's instructions: 4096 x 4096 tokens 26.6 -> see profiles/r04c_quantize_fast_path.txt.
__device__ __forceinline__ void quant4(const float v[4], float amax, float d, int qi[4]) {
  const float r = __builtin_amdgcn_rcpf(d);
  const uint32_t db = __builtin_bit_cast(uint32_t, d) >> 23;          // sign 0 (d >= 0): the biased exponent
  bool slow = !(db >= 27u && db <= 227u);                              // also d == 0 (amax == 0), inf, NaN
  float t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t[i] = v[i] * r;
    slow |= !(fabsf(__builtin_amdgcn_fractf(fabsf(t[i])) - 0.5f) > 1e-3f);   // (NaN / inf: fract is NaN -> slow)
  }
  if (__builtin_expect(slow, 0)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) qi[i] = amax == 0.0f ? 0 : (int)roundf(v[i] / d);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) qi[i] = (int)(t[i] + copysignf(0.5f, t[i]));
  }
}

// Experiment knobs (environment variables that force a kernel variant) exist only in -DGGQ_TUNING builds
// (scripts/build_variant.sh); the shipped library takes no decision from the environment.
#ifdef GGQ_TUNING
#define GGQ_TUNING_ENV(name) getenv(name)
#else
#define GGQ_TUNING_ENV(name) ((const char*)nullptr)
#endif

}  // namespace ggq
