// dequant.hip — block-quantised weights -> fp16, gfx950.
//
// Replaces HK/ggml/dequantize.cuh:3-254 (kernels) + :435-568 (launch/dispatch) and
// ggml_dequantize (HK/ggml/ggml_kernel.cu:68-78) of the reference.
//
// Result contract: bit-identical to the reference kernels, i.e. every __hmul/__hsub/
// __hadd/__int2half_rn of dequantize.cuh is one IEEE fp16 operation here too
// (-ffp-contract=off; the compiler would otherwise fuse hsub(hmul()) into v_fma_f16).
//
// Design (HBM-bound: 0.56-1.06 B/elem in, 2 B/elem out):
//   * one thread = 8 consecutive output elements = one 16-byte store; a wave stores
//     1 KiB contiguous per instruction, so the 78 % of traffic that is writes is
//     perfectly coalesced;
//   * the 8 source bytes of a thread are contiguous in every format, so each thread
//     issues one 8-byte quant load (+ one 4-16 byte scale/header load that is shared
//     by the 4-32 neighbouring lanes of the same block and served by one L1 line);
//     a wave's loads cover a contiguous byte range of the weight row;
//   * no LDS: nothing is reused across lanes except the block header.
// Measured round 2 (scripts/sweep_dequant.py, 11008 x 4096): with the tensor resident in L2 + Infinity Cache ("warm")
// Q4_K runs at 72-75 % of the 8 TB/s roof; streamed from HBM with 4 distinct 90 MB outputs ("cold") at 60 % (48 % with
// one chunk per thread), Q8_0 at 66 %.  On the same box a plain device copy of 86 MB reaches 5.1 TB/s (64 %) cold, a
// fill 6.2 TB/s, and a kernel with this traffic mix and no arithmetic 5.2-5.7 TB/s (scripts/ubench_expand.hip).
// Round 3: nontemporal stores for the output (see the store below): Q4_K 70.6 % cold / 86.7 % warm.
// Tried for the cold case without gain: two ADJACENT chunks per thread (49 -> 40 %: a lane's stores are then 32 bytes
// apart), a grid-stride loop that touches the next chunk's bytes one iteration ahead (45 %).
#include "ggq_common.h"
#include "iq_common.h"

#ifndef GGQ_DEQUANT_CH
// 8-element chunks per thread, 256 chunks apart (so that every store instruction of a wave still writes 1 KiB
// contiguous) with ALL loads of the thread issued first.  Streamed from HBM the one-chunk form is bound by latency x
// resident waves (8192 waves x 1.3 KB per ~2.5 us round trip = 4 TB/s); three chunks in flight per thread give
// Q4_K 48 -> 60 % of the roof cold (29.8 -> 23.9 us) at +-2 % warm.  Sweep 1/2/3/4/6/8 over all formats in
// profiles/r02_dequant_chunks.txt: 3 or 4 is best everywhere.  (Q3_K was the exception while its four loads per chunk were
// 2-byte aligned and bound it in the texture path; with the aligned loads of its decoder below it is 44 -> 55.6 % of the roof
// cold at three chunks, 53.4 % at one.)  The loads can only be hoisted because the decoders are branch-free.
#define GGQ_DEQUANT_CH(T) 3
#endif

#ifndef GGQ_DEQUANT_LDS_GRID
#define GGQ_DEQUANT_LDS_GRID 1
#endif

namespace ggq {

__device__ __forceinline__ _Float16 i2h(int v) { return (_Float16)v; }  // __int2half_rn

// Packed path of the decoders: the integer field extraction runs on four bytes per dword, the bytes
// (0..255, exact in fp16) become half2 pairs via v_cvt_f32_ubyteN + v_cvt_pkrtz_f16_f32, and every __hmul / __hsub
// of the reference is a v_pk_*_f16 on two elements — the same IEEE operation per element, half the vector
// instructions (these formats were VALU-bound at 55-62 % of the HBM roof with the scalar sequence).
__device__ __forceinline__ h2 u8pair_to_h2(uint32_t p, int pair) {   // bytes 2·pair, 2·pair+1 of p
  const float a = (float)((p >> (16 * pair)) & 0xFF), b = (float)((p >> (16 * pair + 8)) & 0xFF);
  return __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
}
__device__ __forceinline__ void st_h2(_Float16* y, int i, h2 v) { y[i] = v[0]; y[i + 1] = v[1]; }
// bit k (k = 0..3) of x -> bit 0 of byte k (the four partial products do not overlap)
__device__ __forceinline__ uint32_t bits4_to_bytes(uint32_t x) { return ((x & 0xF) * 0x00204081u) & 0x01010101u; }

// An 8-byte field at a 2-byte aligned address p, read with ALIGNED loads: three dwords from p - (p & 2), funnelled into
// place (v_alignbyte_b32).  The texture path charges a 2-byte-aligned dword load about three times an aligned one
// (scripts/ubench_gload_align.hip).  Reads bytes [p - 2, p + 10) or [p, p + 12): only for fields with 4 valid bytes behind them
// and 2 in front (inside a block, or in the block before).
struct __attribute__((aligned(4))) u32x3 { uint32_t v[3]; };
struct u32x2 { uint32_t v[2]; };
__device__ __forceinline__ u32x2 ld8_aligned(const uint8_t* p, uint32_t sh) {   // sh = p & 2 (the block's: field offsets are multiples of 4)
  const u32x3 a = *(const u32x3*)(p - sh);
  return u32x2{{__builtin_amdgcn_alignbyte(a.v[1], a.v[0], sh), __builtin_amdgcn_alignbyte(a.v[2], a.v[1], sh)}};
}

// ---- per-format decode of the 8-element chunk `sub` of one block -----------
template <int T> struct Decode;

// Q4_0 / Q4_1 / Q5_0 / Q5_1: element e<16 -> low nibble of qs[e], e>=16 -> high nibble of qs[e-16]
template <> struct Decode<GGQ_TYPE_Q4_0> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const _Float16 d = bits_h(ld_u16(b + off::Q4_0_D));
    const u32x2_a2 q = ld_u32x2(b + off::Q4_0_QS + 8 * (sub & 1));
    const int sh = 4 * (sub >> 1);
    const h2 d2 = {d, d}, off2 = {(_Float16)8.0f, (_Float16)8.0f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = (q.v[w] >> sh) & 0x0F0F0F0Fu;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, (u8pair_to_h2(v4, pr) - off2) * d2);  // dequantize.cuh:11-15
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q4_1> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const uint32_t dm = ld_u32(b + off::Q4_1_D);
    const _Float16 d = bits_h(dm & 0xFFFF), m = bits_h(dm >> 16);
    const u32x2_a2 q = ld_u32x2(b + off::Q4_1_QS + 8 * (sub & 1));
    const int sh = 4 * (sub >> 1);
    const h2 d2 = {d, d}, m2 = {m, m};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = (q.v[w] >> sh) & 0x0F0F0F0Fu;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, u8pair_to_h2(v4, pr) * d2 + m2);  // dequantize.cuh:27-31
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q5_0> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const _Float16 d = bits_h(ld_u16(b + off::Q5_0_D));
    const uint32_t qh = ld_u32(b + off::Q5_0_QH) >> (8 * sub);  // element e <- bit e
    const u32x2_a2 q = ld_u32x2(b + off::Q5_0_QS + 8 * (sub & 1));
    const int sh = 4 * (sub >> 1);
    const h2 d2 = {d, d}, off2 = {(_Float16)16.0f, (_Float16)16.0f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = ((q.v[w] >> sh) & 0x0F0F0F0Fu) | bits4_to_bytes(qh >> (4 * w)) << 4;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, (u8pair_to_h2(v4, pr) - off2) * d2);  // dequantize.cuh:45-49
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q5_1> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const uint32_t dm = ld_u32(b + off::Q5_1_D);
    const _Float16 d = bits_h(dm & 0xFFFF), m = bits_h(dm >> 16);
    const uint32_t qh = ld_u32(b + off::Q5_1_QH) >> (8 * sub);
    const u32x2_a2 q = ld_u32x2(b + off::Q5_1_QS + 8 * (sub & 1));
    const int sh = 4 * (sub >> 1);
    const h2 d2 = {d, d}, m2 = {m, m};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = ((q.v[w] >> sh) & 0x0F0F0F0Fu) | bits4_to_bytes(qh >> (4 * w)) << 4;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, u8pair_to_h2(v4, pr) * d2 + m2);  // dequantize.cuh:64-68
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q8_0> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const _Float16 d = bits_h(ld_u16(b + off::Q8_0_D));
    const u32x2_a2 q = ld_u32x2(b + off::Q8_0_QS + 8 * sub);
    const h2 d2 = {d, d}, bias2 = {(_Float16)128.0f, (_Float16)128.0f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t u4 = q.v[w] ^ 0x80808080u;   // int8 + 128 as unsigned bytes; minus 128 in fp16 is exact
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, (u8pair_to_h2(u4, pr) - bias2) * d2);  // dequantize.cuh:74-77
    }
  }
};

// K-quants. Chunk `sub` (0..31) holds elements 8*sub..8*sub+7 of the super-block.
// Q2_K/Q3_K: element 128n + 32j + l <- (qs[32n+l] >> 2j) & 3      (dequantize.cuh:105-120)
template <> struct Decode<GGQ_TYPE_Q2_K> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int n = sub >> 4, j = (sub >> 2) & 3, l0 = 8 * (sub & 3);
    const uint32_t dm = ld_u32(b + off::Q2_K_D);
    const _Float16 dall = bits_h(dm & 0xFFFF), dmin = bits_h(dm >> 16);
    const int sc = b[off::Q2_K_SC + (sub >> 1)];  // scale of the 16-element group e/16
    const u32x2_a2 q = ld_u32x2(b + off::Q2_K_QS + 32 * n + l0);
    const _Float16 mterm = dmin * i2h(sc >> 4);
    const h2 d2 = {dall, dall}, m2 = {mterm, mterm};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = ((q.v[w] >> (2 * j)) & 0x03030303u) * (uint32_t)(sc & 0xF);   // bytes <= 45: no carries
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, d2 * u8pair_to_h2(v4, pr) - m2);  // dequantize.cuh:117-120
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q3_K> {
  // 110-byte blocks: every field of an even block is 4-byte aligned and every field of an odd block is 2 mod 4.  The texture
  // path charges a 2-byte-aligned dword load about three times an aligned one (scripts/ubench_gload_align.hip), which held
  // this format at 38 - 44 % of the HBM roof cold with four such loads per chunk.  So: the two 8-byte fields are read with
  // ld8_aligned (the surplus bytes are the block's own neighbours, or the last two bytes of the block before — block 0 of
  // a 4-byte aligned tensor is even), the two scale bytes and d as naturally aligned byte / halfword loads.
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int n = sub >> 4, j = (sub >> 2) & 3, l0 = 8 * (sub & 3);
    const _Float16 d_all = bits_h(ld_u16(b + off::Q3_K_D));
    const int i = sub >> 1;   // 16-element group: low nibble (i / 8) of scale byte i % 8, bit pair (i / 4) of byte 8 + i % 4 (dequantize.cuh:140-143)
    const uint32_t blo = b[off::Q3_K_SC + (i & 7)], bhi = b[off::Q3_K_SC + 8 + (i & 3)];
    const int us = (int)(((blo >> (4 * (i >> 3))) & 0xF) | (((bhi >> (2 * (i >> 2))) & 3) << 4)) - 32;
    const _Float16 dl = d_all * i2h(us);  // dequantize.cuh:144-145 (us already minus 32)
    const uint32_t sh = (uint32_t)((uintptr_t)b & 2);
    const u32x2 qq = ld8_aligned(b + off::Q3_K_QS + 32 * n + l0, sh), hh = ld8_aligned(b + off::Q3_K_HM + l0, sh);
    const uint32_t q[2] = {qq.v[0], qq.v[1]}, hm[2] = {hh.v[0], hh.v[1]};
    const int hbit = 4 * n + j;
    const h2 dl2 = {dl, dl}, four = {(_Float16)4.0f, (_Float16)4.0f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      // bytes: q2 + 4·h in 0..7; (q2 - (h ? 0 : 4)) = that minus 4 — exact either way (dequantize.cuh:151)
      const uint32_t v4 = ((q[w] >> (2 * j)) & 0x03030303u) + (((hm[w] >> hbit) & 0x01010101u) << 2);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, dl2 * (u8pair_to_h2(v4, pr) - four));
    }
  }
};
// Q4_K/Q5_K: element 64il + 32h + l <- nibble h of qs[32il + l]   (dequantize.cuh:176-193)
template <> struct Decode<GGQ_TYPE_Q4_K> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int il = sub >> 3, hsel = (sub >> 2) & 1, l0 = 8 * (sub & 3);
    const u32x4_a2 hd = ld_u32x4(b);  // dm + 12 scale bytes
    const _Float16 dall = bits_h(hd.v[0] & 0xFFFF), dmin = bits_h(hd.v[0] >> 16);
    int sc, mn;
    k4_scale_min(hd.v[1], hd.v[2], hd.v[3], sub >> 2, sc, mn);
    const _Float16 d1 = dall * i2h(sc), m1 = dmin * i2h(mn);
    const u32x2_a2 q = ld_u32x2(b + off::Q4_K_QS + 32 * il + l0);
    const h2 d2 = {d1, d1}, m2 = {m1, m1};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = (q.v[w] >> (4 * hsel)) & 0x0F0F0F0Fu;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, d2 * u8pair_to_h2(v4, pr) - m2);  // dequantize.cuh:190-191
    }
  }
};
template <> struct Decode<GGQ_TYPE_Q5_K> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int il = sub >> 3, hsel = (sub >> 2) & 1, l0 = 8 * (sub & 3);
    const u32x4_a2 hd = ld_u32x4(b);
    const _Float16 dall = bits_h(hd.v[0] & 0xFFFF), dmin = bits_h(hd.v[0] >> 16);
    int sc, mn;
    k4_scale_min(hd.v[1], hd.v[2], hd.v[3], sub >> 2, sc, mn);
    const _Float16 d1 = dall * i2h(sc), m1 = dmin * i2h(mn);
    const u32x2_a2 q = ld_u32x2(b + off::Q5_K_QS + 32 * il + l0);
    const u32x2_a2 qh = ld_u32x2(b + off::Q5_K_QH + l0);
    const int hbit = sub >> 2;  // bit 2il + h of qh[l]
    const h2 d2 = {d1, d1}, m2 = {m1, m1};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = ((q.v[w] >> (4 * hsel)) & 0x0F0F0F0Fu) + (((qh.v[w] >> hbit) & 0x01010101u) << 4);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) st_h2(y, 4 * w + 2 * pr, d2 * u8pair_to_h2(v4, pr) - m2);  // dequantize.cuh:222-227
    }
  }
};
// Q6_K: element 128ip + 32j + l <- nibble (j/2) of ql[64ip + 32(j%2) + l] | bits 2j..2j+1 of qh[32ip+l] << 4
template <> struct Decode<GGQ_TYPE_Q6_K> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int ip = sub >> 4, j = (sub >> 2) & 3, l0 = 8 * (sub & 3);
    const _Float16 d = bits_h(ld_u16(b + off::Q6_K_D));
    const int sc = (int8_t)b[off::Q6_K_SC + (sub >> 1)];
    // 210-byte blocks: odd blocks are 2 mod 4 — aligned loads as for Q3_K
    const uint32_t sh = (uint32_t)((uintptr_t)b & 2);
    const u32x2 ql = ld8_aligned(b + off::Q6_K_QL + 64 * ip + 32 * (j & 1) + l0, sh);
    const u32x2 qh = ld8_aligned(b + off::Q6_K_QH + 32 * ip + l0, sh);
    // i2h(sc·(q-32)) = one round-to-nearest of an exact integer product = the fp16 product of the exact halves
    const h2 d2 = {d, d}, sc2 = {i2h(sc), i2h(sc)}, off2 = {(_Float16)32.0f, (_Float16)32.0f}, zero2 = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t v4 = ((ql.v[w] >> (4 * (j >> 1))) & 0x0F0F0F0Fu) | (((qh.v[w] >> (2 * j)) & 0x03030303u) << 4);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)   // (+0: the integer product 0 converts to +0, sc·(+0) would be -0 for sc < 0)
        st_h2(y, 4 * w + 2 * pr, d2 * (sc2 * (u8pair_to_h2(v4, pr) - off2) + zero2));  // dequantize.cuh:250-253
    }
  }
};

// IQ4_NL / IQ4_XS (HK/ggml/dequantize.cuh:399-433): fp32 arithmetic — d (x (ls - 32)) x codebook value — and ONE
// rounding to fp16 (`__float2half`), unlike the fp16 sequences above.  Chunk `sub` holds elements 8 sub..8 sub + 7
// of a 32-element (sub-)block: low nibbles of qs[0..15] are elements 0..15, high nibbles elements 16..31.
__device__ __forceinline__ void iq4_chunk(const uint8_t* qs16, int sub4, float d, _Float16* y) {
  const u32x2_a2 q = ld_u32x2(qs16 + 8 * (sub4 & 1));
  const int sh = 4 * (sub4 >> 1);
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const uint32_t v4 = iq4nl_lookup4((q.v[w] >> sh) & 0x0F0F0F0Fu);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p = d * (float)(int8_t)(v4 >> (8 * e));
      // keep the product a plain v_mul_f32: folded with the conversion hipcc emits v_fma_mix*_f16 p = d·v + (+0),
      // which turns the reference's -0 (d = 0 or ls = 32 times a negative codebook value) into +0
      asm volatile("" : "+v"(p));
      y[4 * w + e] = (_Float16)p;
    }
  }
}
template <> struct Decode<GGQ_TYPE_IQ4_NL> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    iq4_chunk(b + off::IQ4_NL_QS, sub, bits_h_f32(ld_u16(b + off::IQ4_NL_D)), y);
  }
};
template <> struct Decode<GGQ_TYPE_IQ4_XS> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) {
    const int ib = sub >> 2;   // 32-element sub-block
    const u32x2_a2 hd = ld_u32x2(b);   // {d | scales_h << 16, scales_l}
    const float d = bits_h_f32(hd.v[0] & 0xFFFF) * (float)iq4xs_scale(hd.v[0] >> 16, hd.v[1], ib);
    iq4_chunk(b + off::IQ4_XS_QS + 16 * ib, sub & 3, d, y);
  }
};

// The grid-codebook IQ formats (HK/ggml/dequantize.cuh:256-398, 471-512): chunk `sub` = 8-element run il = sub & 3 of
// 32-element sub-block ib = sub >> 2 (the reference's thread (il, ib)); fp32 arithmetic  d * grid[j] * (+-1)  with
// d = half2float(x.d) * (0.5f + scale) * 0.25f | 0.5f evaluated left to right, ONE rounding to fp16.  (Multiplying by
// +-1 commutes with the rounding: the product with the signed grid value is the same number, zero signs included.)
template <int T> struct IqDecode {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) { run_g(IqGrid<T>::table(), b, sub, y); }
  static __device__ __forceinline__ void run_g(const void* grid, const uint8_t* b, int sub, _Float16* y) {   // grid: the codebook (global or an LDS copy)
    uint32_t lo, hi;
    float mul;
    IqRun<T>::get(grid, b, sub >> 2, sub & 3, lo, hi, mul);
    const float d = bits_h_f32(ld_u16(b)) * mul * IqRun<T>::post;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p0 = d * (float)(int8_t)(lo >> (8 * e)), p1 = d * (float)(int8_t)(hi >> (8 * e));
      asm volatile("" : "+v"(p0), "+v"(p1));   // plain v_mul_f32 (see iq4_chunk: a fused mix-fma would lose the sign of -0)
      y[e] = (_Float16)p0;
      y[4 + e] = (_Float16)p1;
    }
  }
};
template <> struct Decode<GGQ_TYPE_IQ2_XXS> : IqDecode<GGQ_TYPE_IQ2_XXS> {};
template <> struct Decode<GGQ_TYPE_IQ2_XS> : IqDecode<GGQ_TYPE_IQ2_XS> {};
template <> struct Decode<GGQ_TYPE_IQ2_S> : IqDecode<GGQ_TYPE_IQ2_S> {};
template <> struct Decode<GGQ_TYPE_IQ3_XXS> : IqDecode<GGQ_TYPE_IQ3_XXS> {};
template <> struct Decode<GGQ_TYPE_IQ3_S> : IqDecode<GGQ_TYPE_IQ3_S> {};

// IQ1_S / IQ1_M (dequantize.cuh:354-398): y = d * (q + delta), q in {0, 1, 2} from the 2048-entry grid, delta = -1 +- 0.125,
// d = half2float(d) * (2 * scale3 + 1)
__device__ __forceinline__ void iq1_emit(uint32_t lo, uint32_t hi, float d, float delta, _Float16* y) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float p0 = d * ((float)(int8_t)(lo >> (8 * e)) + delta), p1 = d * ((float)(int8_t)(hi >> (8 * e)) + delta);
    asm volatile("" : "+v"(p0), "+v"(p1));
    y[e] = (_Float16)p0;
    y[4 + e] = (_Float16)p1;
  }
}
template <> struct Decode<GGQ_TYPE_IQ1_S> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) { run_g(ggq_iq1s_grid_gpu, b, sub, y); }
  static __device__ __forceinline__ void run_g(const void* grid, const uint8_t* b, int sub, _Float16* y) {
    const int ib = sub >> 2, il = sub & 3;
    const uint32_t qh = ld_u16(b + off::IQ1_S_QH + 2 * ib);
    uint32_t lo, hi;
    iq1_grid(grid, b[off::IQ1_S_QS + 4 * ib + il] | (((qh >> (3 * il)) & 7) << 8), lo, hi);
    const float delta = (qh & 0x8000) ? -1.0f - IQ1_DELTA : -1.0f + IQ1_DELTA;
    iq1_emit(lo, hi, bits_h_f32(ld_u16(b + off::IQ1_S_D)) * (float)(2 * ((qh >> 12) & 7) + 1), delta, y);
  }
};
template <> struct Decode<GGQ_TYPE_IQ1_M> {
  static __device__ __forceinline__ void run(const uint8_t* b, int sub, _Float16* y) { run_g(ggq_iq1s_grid_gpu, b, sub, y); }
  static __device__ __forceinline__ void run_g(const void* grid, const uint8_t* b, int sub, _Float16* y) {
    const int ib = sub >> 2, il = sub & 3;
    const int ib16 = 2 * ib + (il >> 1);
    const uint32_t sc = ld_u16(b + off::IQ1_M_SC + 2 * (ib16 >> 2));
    const uint32_t qh = b[off::IQ1_M_QH + ib16] >> (4 * (il & 1));
    uint32_t lo, hi;
    iq1_grid(grid, b[off::IQ1_M_QS + 4 * ib + il] | ((qh & 7) << 8), lo, hi);
    const float delta = (qh & 0x08) ? -1.0f - IQ1_DELTA : -1.0f + IQ1_DELTA;
    iq1_emit(lo, hi, iq1m_super_scale(b) * (float)(2 * ((sc >> (3 * (ib16 & 3))) & 7) + 1), delta, y);
  }
};

// CH = 8-element chunks per thread, 256 apart (GGQ_DEQUANT_CH above).
template <int T, int CH>
__global__ void __launch_bounds__(256) dequant_kernel(const uint8_t* __restrict__ w,
                                                      _Float16* __restrict__ out,
                                                      int64_t n_chunks) {
  constexpr int CPB = Fmt<T>::QK / 8;  // 8-element chunks per block
  const int64_t c0 = (int64_t)blockIdx.x * (256 * CH) + threadIdx.x;
  // the grid-codebook IQ formats: the codebook (1 - 8 KB) is staged in LDS — a per-lane lookup in global memory touches up to 64
  // cache lines per wave-level load, in LDS it is one ds_read (GGQ_DEQUANT_LDS_GRID = 0: ablation)
  constexpr int GB = GGQ_DEQUANT_LDS_GRID ? IqGrid<T>::BYTES : 0;
  __shared__ __attribute__((aligned(16))) uint8_t lgrid[GB ? GB : 16];
  if constexpr (GB != 0) {
    const v4i* gg = (const v4i*)IqGrid<T>::table();
    for (int i = threadIdx.x; i < GB / 16; i += 256) ((v4i*)lgrid)[i] = gg[i];
    __syncthreads();
  }
  h8 v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int64_t c = min(c0 + i * 256, n_chunks - 1);   // clamped, never predicated: the loads of all chunks go out first
    const int64_t ib = c / CPB;
    _Float16 y[8];
    if constexpr (GB != 0) Decode<T>::run_g(lgrid, w + ib * Fmt<T>::BS, (int)(c - ib * CPB), y);
    else Decode<T>::run(w + ib * Fmt<T>::BS, (int)(c - ib * CPB), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = y[e];
  }
#pragma unroll
  for (int i = 0; i < CH; ++i)
    if (c0 + i * 256 < n_chunks) {
      // nontemporal: the 90 MB of fp16 output are written once and not re-read by this kernel — kept out of L2 they stop
      // evicting the input stream (round 3, scripts/sweep_dequant.py: Q4_K 24.3 -> 20.5 us cold = 59.5 -> 70.6 % of the
      // 8 TB/s roof, 19.4 -> 16.7 us warm; Q4_0 61.9 -> 72.7 % cold; round 2 had measured this as "+-0" on the
      // one-chunk-per-thread form)
      __builtin_nontemporal_store(v[i], (h8*)(out + (c0 + i * 256) * 8));
    }

}

template <int T>
static int launch_dequant(const void* w, void* out, int64_t k, hipStream_t s) {
  constexpr int CH = GGQ_DEQUANT_CH(T);
  const int64_t n_chunks = k / 8;
  if (n_chunks == 0) return GGQ_OK;
  const int64_t grid = (n_chunks + 256 * CH - 1) / (256 * CH);
  if (grid > 0x7fffffffLL) return GGQ_ERR_SHAPE;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL((dequant_kernel<T, CH>), dim3((unsigned)grid), dim3(256), 0, s,
                     (const uint8_t*)w, (_Float16*)out, n_chunks);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

}  // namespace ggq

extern "C" int ggq_dequantize_f16(const void* w, void* out, int type, int64_t m, int64_t n,
                                  void* stream) {
  using namespace ggq;
  if (m < 0 || n < 0) return GGQ_ERR_ARG;
  const int qk = ggq_block_elems(type);
  if (qk == 0 || type == GGQ_TYPE_Q8_1) return GGQ_ERR_TYPE;
  const int64_t k = m * n;
  if (k % qk) return GGQ_ERR_SHAPE;
  if (k == 0) return GGQ_OK;
  if (!w || !out) return GGQ_ERR_ARG;
  if (((uintptr_t)out & 15) || ((uintptr_t)w & 1)) return GGQ_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
    case GGQ_TYPE_Q4_0: return launch_dequant<GGQ_TYPE_Q4_0>(w, out, k, s);
    case GGQ_TYPE_Q4_1: return launch_dequant<GGQ_TYPE_Q4_1>(w, out, k, s);
    case GGQ_TYPE_Q5_0: return launch_dequant<GGQ_TYPE_Q5_0>(w, out, k, s);
    case GGQ_TYPE_Q5_1: return launch_dequant<GGQ_TYPE_Q5_1>(w, out, k, s);
    case GGQ_TYPE_Q8_0: return launch_dequant<GGQ_TYPE_Q8_0>(w, out, k, s);
    case GGQ_TYPE_Q2_K: return launch_dequant<GGQ_TYPE_Q2_K>(w, out, k, s);
    case GGQ_TYPE_Q3_K: return launch_dequant<GGQ_TYPE_Q3_K>(w, out, k, s);
    case GGQ_TYPE_Q4_K: return launch_dequant<GGQ_TYPE_Q4_K>(w, out, k, s);
    case GGQ_TYPE_Q5_K: return launch_dequant<GGQ_TYPE_Q5_K>(w, out, k, s);
    case GGQ_TYPE_Q6_K: return launch_dequant<GGQ_TYPE_Q6_K>(w, out, k, s);
    case GGQ_TYPE_IQ4_NL: return launch_dequant<GGQ_TYPE_IQ4_NL>(w, out, k, s);
    case GGQ_TYPE_IQ4_XS: return launch_dequant<GGQ_TYPE_IQ4_XS>(w, out, k, s);
    case GGQ_TYPE_IQ2_XXS: return launch_dequant<GGQ_TYPE_IQ2_XXS>(w, out, k, s);
    case GGQ_TYPE_IQ2_XS: return launch_dequant<GGQ_TYPE_IQ2_XS>(w, out, k, s);
    case GGQ_TYPE_IQ2_S: return launch_dequant<GGQ_TYPE_IQ2_S>(w, out, k, s);
    case GGQ_TYPE_IQ3_XXS: return launch_dequant<GGQ_TYPE_IQ3_XXS>(w, out, k, s);
    case GGQ_TYPE_IQ3_S: return launch_dequant<GGQ_TYPE_IQ3_S>(w, out, k, s);
    case GGQ_TYPE_IQ1_S: return launch_dequant<GGQ_TYPE_IQ1_S>(w, out, k, s);
    case GGQ_TYPE_IQ1_M: return launch_dequant<GGQ_TYPE_IQ1_M>(w, out, k, s);
    default: return GGQ_ERR_TYPE;
  }
}
