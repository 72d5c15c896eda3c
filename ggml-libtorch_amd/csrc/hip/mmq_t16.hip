// mmq_t16.hip — quantised GEMM for the HBM-bound batches (2 .. 32 tokens; Q4_K Q5_K, and Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q6_K up to 16
// tokens — their operand paths are described at `F::legacy` / `F::k6` below): 16-row x 16-token MFMA tiles, a wave's whole
// share of the weight matrix requested before anything is waited for.  gfx950 only.
//
// Same contract as mmq.hip (mul_mat_q, HK/ggml/mmq.cuh:1917-1986, "MMQ canon" of SURVEY §8a: exact int8 contraction per
// 32-element group, the float factors of the reference's tensor-core bodies, fp32 accumulation order ours); what differs is
// the regime it is built for.  Between the GEMV and the 32-token MFMA tile the matmul is bound by how fast the weight
// matrix streams out of HBM, i.e. by bytes in flight and by what the texture addresser pays per 128-byte line a wave-level
// load touches (measured with the per-wave stamps of scripts/stamps_t16.py: the first version of this kernel
// spent 2.5 - 5 us just ISSUING its requests, because 16 rows x 64 bytes per load is 16 lines per KB and the activation
// fragments cost as many lines again).  So:
//   * workgroup = one 16-row weight tile x KS K-slices (one wave each); a wave's slice is at most MAXU 256-element units and
//     the first thing a wave does is request all of it by LDS-DMA (global_load_lds_dwordx4: no destination registers, so
//     the whole slice is in flight whatever the register budget) in ROW-MAJOR LINEAR order — consecutive lanes = consecutive
//     16-byte chunks of a row's slice, ~9.5 lines per KB, block headers included — into a wave-private LDS image
//     [16 rows][slice bytes].  The K loop is a rolled loop over units that reads its MFMA operands from that image.
//   * v_mfma_i32_16x16x64_i8, A = activations (rows = tokens), B = weights (columns = 16 weight rows): lane (row j = lane & 15,
//     K-chunk c = lane >> 4) holds bytes 16 c .. 16 c + 15 of a 64-byte half of the unit's nibble field: low nibbles of one
//     32-group half and high nibbles of the next group's half; lanes c < 2 belong to one group pair, lanes c >= 2 to the next.
//       batch <= 8  ("M8"): the 16 token rows of A are tokens 0-7 twice — rows 0-7 carry the activations only in the K-chunks
//         of the first pair, rows 8-15 only in those of the second (the other chunks read as zeros: buffer loads past the
//         descriptor's range, no memory traffic) — so ONE MFMA yields two groups' sums (rows 0-7 / 8-15), the weight
//         operand needs no masking, and a lane scales 4 results per MFMA instead of 8 useful ones out of 16.
//       batch 9-16 and NTT > 1: one MFMA per group with the other pair's lanes of B zeroed.
//   * output: lane = weight row, register r = token 4 c + r (M8: token 4 (c & 1) + r, group pair c >> 1).  The row scale
//     d * sc is a lane scalar decoded from the row's header; the token scales half2(d8, s8) come from a wave-private LDS copy
//     of the slice's scale table (broadcast ds_read_b128), applied with v_fma_mix_f32 (reads the fp16 d8 in place).
//   * Q4_K / Q5_K min term  sum_g s8[token][g] * -(dmin m_g)[row]: K = 4 per unit half -> v_mfma_f32_16x16x4_f32 straight
//     into the accumulators (exact fp32 FMAs in a fixed order, off the vector ALU).
//   * activations: the 16- / 8-token-tile fragment layouts of quantize.hip (LAYOUT 3 / 4), one coalesced load per MFMA
//     operand, read from L2 by every wave, requested one unit ahead.
//   * K-slice partial sums meet once in LDS, summed in slice order (fixed order: bit-reproducible, independent of the
//     wave that adds).
#include "ggq_common.h"
#include "mmq_unpack.h"

#ifndef GGQ_T16_STAMP
#define GGQ_T16_STAMP 0   // 1: per-wave timestamps (scripts/stamps_t16.py); 0 in every shipped build
#endif
#if GGQ_T16_STAMP
__device__ unsigned long long g_t16_stamps[4096 * 16 * 8];
extern "C" int ggq_debug_read_t16_stamps(void* dst, long long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_t16_stamps), n * 8);
}
#define T16_STAMP(i)                                                                                          \
  do {                                                                                                        \
    if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0) g_t16_stamps[(blockIdx.x * 16 + ks) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define T16_STAMP(i) do {} while (0)
#endif

namespace ggq {

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

struct Epi16 { int kind; const void* aux; GatherOut go; };
template <int DT>
__device__ __forceinline__ float t16_epilogue(float v, int epi, const void* aux, int64_t yi, int row) {
  if (epi == GGQ_EPI_BIAS) return v + Elem<DT>::ld(aux, row);
  if (epi == GGQ_EPI_SILU_MUL) {
    const float g = Elem<DT>::ld(aux, yi);
    return v * (g / (1.0f + expf(-g)));
  }
  return v;
}

template <int T> struct T16Fmt {
  static_assert(T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K || T == GGQ_TYPE_Q8_0 || T == GGQ_TYPE_Q4_0 || T == GGQ_TYPE_Q4_1 ||
                T == GGQ_TYPE_Q5_0 || T == GGQ_TYPE_Q5_1 || T == GGQ_TYPE_Q6_K || T == GGQ_TYPE_Q3_K, "format");
  static constexpr bool k6 = T == GGQ_TYPE_Q6_K;             // 210-byte super-blocks, one int8 scale per 16 elements
  static constexpr bool k3 = T == GGQ_TYPE_Q3_K;             // 110-byte super-blocks, one 6-bit scale per 16 elements
  static constexpr bool sub16 = k6 || k3;                    // a scale per 16-element sub-block, units that are no multiple of 16 bytes
  static constexpr bool legacy = Fmt<T>::QK == 32;           // 32-element blocks {fp16 d [, fp16 m] [, u32 qh], qs}, no super-block header
  static constexpr int UB = 256 / Fmt<T>::QK * Fmt<T>::BS;   // bytes of one 256-element unit of a weight row — a multiple of 16
                                                             // (eight 18 / 20 / 22 / 24 / 34-byte blocks: 144 .. 272) except Q6_K's 210
  static constexpr int QS = T == GGQ_TYPE_Q4_K ? off::Q4_K_QS : T == GGQ_TYPE_Q5_K ? off::Q5_K_QS : 2;
  static constexpr bool has_qh = T == GGQ_TYPE_Q5_K;
  // 32-element blocks
  static constexpr int BS = Fmt<T>::BS;
  static constexpr bool nib = legacy && T != GGQ_TYPE_Q8_0;                       // 4- / 5-bit values, 8 raw bytes per lane
  static constexpr bool blk_qh = T == GGQ_TYPE_Q5_0 || T == GGQ_TYPE_Q5_1;        // fifth bits: u32 behind the scale(s)
  static constexpr bool blk_m = T == GGQ_TYPE_Q4_1 || T == GGQ_TYPE_Q5_1;         // half2(d, m), fp16 products (mmq.cuh:527-529)
  static constexpr int BQH = blk_m ? 4 : 2;                                        // offset of qh
  static constexpr int BQS = (blk_m ? 4 : 2) + (blk_qh ? 4 : 0);                   // offset of qs
  static constexpr int sub = T == GGQ_TYPE_Q4_0 ? 8 : T == GGQ_TYPE_Q5_0 ? 16 : 0; // value = raw - sub (exact integer contraction)
  static constexpr bool d8_half = T == GGQ_TYPE_Q4_0 || blk_m;                     // need_sum formats: the table holds half2(d8, s8)
};

// wave-private LDS (bytes): W [16 rows][MAXU units] raw bytes | TAB scale tables
template <int T, bool M8, int NTT, int MAXU> struct T16Lds {
  // LDS pitch of one row's slice: whole 16-byte DMA chunks.  For the 32-element-block formats an ODD number of them (one pad chunk
  // that repeats valid bytes): the 16 lanes of a K-chunk read at this pitch, which covers the 64 banks exactly once only then
  // (Q4_0 batch 16 12.2 -> 11.0 us, Q8_0 12.3 -> 11.6; for the K-quants the same pad measured 0 .. +7 %: not applied).
  static constexpr int SB0 = (MAXU * T16Fmt<T>::UB + 15) / 16 * 16;
  static constexpr int SB = (T16Fmt<T>::legacy && (SB0 / 16) % 2 == 0) ? SB0 + 16 : SB0;
  static constexpr int NI = (16 * SB + 1023) / 1024;       // DMA instructions (1 KB each) of the weight image (the last may be partial:
                                                           // its surplus lanes repeat the final chunk into the padding)
  static constexpr int TT = M8 ? 256 : 512;                // scale table of one (unit, token tile)
  static constexpr int W = 0;
  static constexpr int TAB = NI * 1024;
  static constexpr int WAVE = TAB + (M8 ? 1024 : NTT * ((MAXU + 1) / 2) * 1024);   // whole table DMA instructions
  static constexpr int TILE = M8 ? 2304 : 4608;            // activation tile of one (unit, token tile)
  static constexpr int FRAG = M8 ? 512 : 1024;
};

typedef const __attribute__((address_space(1))) void* t16_gptr;
typedef __attribute__((address_space(3))) void* t16_lptr;

// M8: at most 8 tokens (NTT = 1).  NTT = 16-token tiles per workgroup.  MAXU = units of a wave's slice held in LDS.
template <int T, int DT, bool M8, int NTT, int MAXU, int MAXKS>
__global__ void __launch_bounds__(64 * MAXKS) mmq_t16_kernel(const uint8_t* __restrict__ w, const uint8_t* __restrict__ q8,
                                                             void* __restrict__ y, int n_units, int ups, int n_rows, int batch,
                                                             int64_t ldy, int n_tt, int epi, const void* __restrict__ aux, GatherOut go) {
  using F = T16Fmt<T>;
  using L = T16Lds<T, M8, NTT, MAXU>;
  static_assert(!M8 || NTT == 1, "M8: one token tile");
  constexpr int UB = F::UB, SB = L::SB;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [wave] T16Lds ; aliased by the K-slice reduction
  const int lane = threadIdx.x & 63;
  const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KS = blockDim.x >> 6;
  const int j = lane & 15, c = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int tt0 = blockIdx.y * NTT;
  const int u0 = ks * ups, u1 = min(n_units, u0 + ups);   // the launcher makes every slice non-empty
  const uint32_t row_bytes = (uint32_t)n_units * UB;
  const uint8_t* wtile = w + (int64_t)n0 * row_bytes;
  const int rmax = min(15, n_rows - 1 - n0);               // rows past the tensor repeat the last one (never stored)
  uint8_t* wl = lds + ks * L::WAVE;
  T16_STAMP(0);

  int k6_shift = 0;                                        // (Q6_K) see request_round
  const bool k6_last_row = n0 + min(j, rmax) == n_rows - 1;
  // (Q6_K) descriptor over the whole weight tensor (the launcher admits it below 4 GiB): the shifted copy of the last row may
  // start in the row before this tile
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((uint32_t)n_rows * row_bytes), 0x00020000);
  // activations: one descriptor over the scratch, scalar tile offsets, the lane's 16 bytes of a fragment
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q8, 0, (int)0x7FFFFFFF, 0x00020000);
  auto tile_off = [&](int u, int jj) { return (uint32_t)((u * n_tt + min(tt0 + jj, n_tt - 1)) * L::TILE); };
  // M8: lane (A row j, K-chunk c) holds token j & 7 in the chunks of its row half's group pair and ZERO in the others
  // (offset beyond the descriptor's range: the load returns 0 and touches no memory)
  const uint32_t frag_voff = !M8 ? (uint32_t)lane * 16 : (F::sub16 || (j < 8) == (c < 2)) ? (uint32_t)(c * 8 + (j & 7)) * 16 : 0x80000000u;
  auto ld_frag = [&](uint32_t toff, int f) {
    const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)frag_voff, (int)(toff + f * L::FRAG), 0);
    return v4i{(int)t[0], (int)t[1], (int)t[2], (int)t[3]};
  };
  // fragments of one unit (two halves x NTT token tiles x {low, high} nibble operands), requested one unit ahead
  struct Frags { v4i a[2][NTT][2]; };
  auto request_frags = [&](Frags& Fr, int u) {
    const int ue = min(u, u1 - 1);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int jj = 0; jj < NTT; ++jj) {
        const uint32_t toff = tile_off(ue, jj);
        Fr.a[q][jj][0] = ld_frag(toff, 2 * q);
        Fr.a[q][jj][1] = ld_frag(toff, 2 * q + 1);
      }
  };
  // ---- everything the wave reads of units [ub, ub + MAXU) is requested here by LDS-DMA: scale tables, then the weight
  //      image in row-major linear order (chunk n = 64 i + lane of instruction i: row n / CPR, 16-byte column n % CPR) ----
  auto request_round = [&](int ub) {
    if constexpr (M8) {   // 256 bytes per unit: lane = (unit lane >> 4, 16 bytes lane & 15); units past MAXU land in the padding
      const uint32_t o = tile_off(min(ub + min(lane >> 4, MAXU - 1), u1 - 1), 0) + 2048 + (lane & 15) * 16;
      __builtin_amdgcn_global_load_lds((t16_gptr)(q8 + o), (t16_lptr)(wl + L::TAB), 16, 0, 0);
    } else {
#pragma unroll
      for (int jj = 0; jj < NTT; ++jj)
#pragma unroll
        for (int sp = 0; sp < (MAXU + 1) / 2; ++sp) {   // 512 bytes per (unit, token tile): lanes 0-31 unit 2 sp, lanes 32-63 unit 2 sp + 1
          const uint32_t o = tile_off(min(ub + 2 * sp + (lane >> 5), u1 - 1), jj) + 4096 + (lane & 31) * 16;
          __builtin_amdgcn_global_load_lds((t16_gptr)(q8 + o), (t16_lptr)(wl + L::TAB + (jj * ((MAXU + 1) / 2) + sp) * 1024), 16, 0, 0);
        }
    }
    constexpr int CPR = SB / 16, CPU = UB / 16;   // 16-byte chunks per row slice / per unit
    if constexpr (F::sub16) {
      // 210- / 110-byte units: a slice is not a whole number of chunks and starts at a 2-byte aligned address.  The copy runs along
      // the row from the slice's first byte and reads SB bytes: past the slice that is the next slice or the next row — except
      // in the tensor's LAST row, whose copy is shifted down by `k6_shift` bytes so that it ends with the row (the reader adds
      // the shift back; LDS reads need no alignment).  The buffer descriptor ends with the tensor in any case.
      k6_shift = max(0, ub * UB + SB - (int)row_bytes);
#pragma unroll
      for (int i = 0; i < L::NI; ++i) {
        const int n = min(64 * i + lane, 16 * CPR - 1);
        const int row = min(n / CPR, rmax), col = n - (n / CPR) * CPR;
        const uint32_t voff = (uint32_t)(n0 + row) * row_bytes + (uint32_t)ub * UB + 16u * (uint32_t)col - (n0 + row == n_rows - 1 ? (uint32_t)k6_shift : 0u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (t16_lptr)(wl + L::W + i * 1024), 16, (int)voff, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      const int n = min(64 * i + lane, 16 * CPR - 1);
      const int row = n / CPR, col = n - row * CPR;
      const int su = col / CPU, within = col - su * CPU;
      const uint8_t* src = wtile + ((uint32_t)min(row, rmax) * row_bytes + (uint32_t)min(ub + su, u1 - 1) * UB + 16 * within);
      // aux = 2: nontemporal — a launch reads every weight byte once (one workgroup column covers the batch); with the weights
      // streamed from HBM 9.49 -> 9.03 us at batch 8, 11.3 -> 10.6 at 16 (replayed on one resident tensor 6.90 -> 7.49)
      __builtin_amdgcn_global_load_lds((t16_gptr)src, (t16_lptr)(wl + L::W + i * 1024), 16, 0, 2);
    }
  };

  v4f acc[NTT];
#pragma unroll
  for (int jj = 0; jj < NTT; ++jj) acc[jj] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t m_lo = c < 2 ? 0x0F0F0F0Fu : 0u, m_hi = c < 2 ? 0u : 0x0F0F0F0Fu;   // lanes of the unit half's first / second group pair
  const int sel = c >> 1;   // the lane's group pair within a unit half

  // one unit: slot s of the round, fragments Fr.  Every LDS read of the unit is issued before the first use (one wait
  // instead of a read -> wait -> use chain per operand), the min term has its own accumulator (its MFMA chain does not
  // serialise with the scaling FMAs).
  v4f accm[NTT];   // min-term accumulators (f32 MFMA chain)
#pragma unroll
  for (int jj = 0; jj < NTT; ++jj) accm[jj] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
  auto compute_unit = [&](int s, const Frags& Fr) {
    if constexpr (F::sub16) {
      // ---- Q6_K {ql[128]; qh[64]; int8 scales[16]; fp16 d}: element 128 ip + 32 jq + l = nibble (jq >> 1) of ql[64 ip + 32 (jq & 1) + l]
      //      | bits 2 jq .. of qh[32 ip + l] << 4, minus 32 (dequantize.cuh:236-253).  Lane (row j, chunk c) of half ip reads
      //      ql[64 ip + 16 c .. + 15] and qh[32 ip + 16 (c & 1) .. + 15]: its low nibbles are the 16-element sub-block 8 ip + c, its high
      //      nibbles sub-block 8 ip + 4 + c — operand f = 2 ip + hi holds sub-block 4 f + c in chunk c, the identity order of the
      //      activation fragments.
      //      Q3_K {hmask[32]; qs[64]; scales[12]; fp16 d}: element 128 ip + 32 jq + l = ((qs[32 ip + l] >> 2 jq) & 3) - (bit 4 ip + jq of
      //      hmask[l] ? 0 : 4) (dequantize.cuh:123-152).  The lane reads qs[32 ip + 16 (c & 1) ..] and hmask[16 (c & 1) ..]; operand
      //      f = 2 ip + m takes jq = 2 m + (c >> 1) out of them: again sub-block 4 f + c in chunk c.
      //      Every sub-block has its own scale, so a K = 64 MFMA must not mix chunks: M8 splits an operand in two MFMAs (token rows
      //      0-7 take chunk 0 | 2, rows 8-15 chunk 1 | 3), the 16-token form in four with the other chunks of B zeroed.
      //      float(C) d8 (d sc) per sub-block (mmq.cuh:1726-1732, :51-72 re-associated). ----
      typedef unsigned v4u_a2 __attribute__((ext_vector_type(4), aligned(2)));
      const uint8_t* blk = wl + L::W + j * SB + s * UB + (k6_last_row ? k6_shift : 0);
      const float d6 = bits_h_f32(*(const uint16_t*)(blk + (F::k6 ? off::Q6_K_D : off::Q3_K_D)));
      const v4i zero = {0, 0, 0, 0};
      auto sub_off = [](uint32_t x) {   // per byte x - 32 (Q6_K, x <= 63) / x - 4 (Q3_K, x <= 7) as int8
        constexpr uint32_t o = F::k6 ? 0x20202020u : 0x04040404u;
        return ((x | 0x80808080u) - o) ^ 0x80808080u;
      };
      struct __attribute__((packed, aligned(2))) u32x3_l { uint32_t v[3]; };
      u32x3_l s3 = {};
      if constexpr (F::k3) s3 = *(const u32x3_l*)(blk + off::Q3_K_SC);
      auto sub_scale = [&](int sb) -> float {   // the sub-block's integer scale
        if constexpr (F::k6) return (float)(int8_t)blk[off::Q6_K_SC + sb];
        else return (float)q3k_scale(s3.v[0], s3.v[1], s3.v[2], sb);
      };
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int jl = c >> 1;
        v4i B[2];
        if constexpr (F::k6) {
          const v4u_a2 ql = *(const v4u_a2*)(blk + off::Q6_K_QL + 64 * ip + 16 * c);
          const v4u_a2 qh = *(const v4u_a2*)(blk + off::Q6_K_QH + 32 * ip + 16 * (c & 1));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            B[0][i] = (int)sub_off((ql[i] & 0x0F0F0F0Fu) | (((qh[i] >> (2 * jl)) & 0x03030303u) << 4));
            B[1][i] = (int)sub_off(((ql[i] >> 4) & 0x0F0F0F0Fu) | (((qh[i] >> (2 * jl + 4)) & 0x03030303u) << 4));
          }
        } else {
          const v4u_a2 qs = *(const v4u_a2*)(blk + off::Q3_K_QS + 32 * ip + 16 * (c & 1));
          const v4u_a2 hm = *(const v4u_a2*)(blk + off::Q3_K_HM + 16 * (c & 1));
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              B[m][i] = (int)sub_off(((qs[i] >> (2 * (2 * m + jl))) & 0x03030303u) | (((hm[i] >> (4 * ip + 2 * m + jl)) & 0x01010101u) << 2));
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int f = 2 * ip + m;
          if constexpr (M8) {
            const v4i A = Fr.a[ip][0][m];
            const bool ka = j < 8 ? c == 0 : c == 1, kb = j < 8 ? c == 2 : c == 3;
            v4i Aa, Ab;
#pragma unroll
            for (int i = 0; i < 4; ++i) { Aa[i] = ka ? A[i] : 0; Ab[i] = kb ? A[i] : 0; }
            const v4i Ca = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aa, B[m], zero, 0, 0, 0);   // lane quad c: sub-block 4 f + (c >> 1)
            const v4i Cb = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ab, B[m], zero, 0, 0, 0);   //              sub-block 4 f + 2 + (c >> 1)
            const float dwa = d6 * sub_scale(4 * f + sel), dwb = d6 * sub_scale(4 * f + 2 + sel);
            const uint8_t* tq = wl + L::TAB + s * 256 + ip * 128 + (c & 1) * 64;   // [token quad][group of the half][token] fp32 d8
            const v4u_t da = *(const v4u_t*)(tq + (2 * m) * 16), db = *(const v4u_t*)(tq + (2 * m + 1) * 16);
            const uint32_t ta[4] = {da[0], da[1], da[2], da[3]}, tb[4] = {db[0], db[1], db[2], db[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[0][r] = __builtin_fmaf((float)Ca[r] * as_f32((int)ta[r]), dwa, acc[0][r]);
              acc[0][r] = __builtin_fmaf((float)Cb[r] * as_f32((int)tb[r]), dwb, acc[0][r]);
            }
          } else {
            v4i Bm[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
              for (int i = 0; i < 4; ++i) Bm[cc][i] = c == cc ? B[m][i] : 0;
            float dw[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) dw[cc] = d6 * sub_scale(4 * f + cc);
#pragma unroll
            for (int jj = 0; jj < NTT; ++jj) {
              const uint8_t* tq = wl + L::TAB + (jj * ((MAXU + 1) / 2) * 2 + s) * 512 + ip * 256 + c * 64;
              const v4u_t d0 = *(const v4u_t*)(tq + (2 * m) * 16), d1 = *(const v4u_t*)(tq + (2 * m + 1) * 16);   // tokens 4 c .. + 3, groups 2 m, 2 m + 1 of the half
              const uint32_t tw[2][4] = {{d0[0], d0[1], d0[2], d0[3]}, {d1[0], d1[1], d1[2], d1[3]}};
#pragma unroll
              for (int cc = 0; cc < 4; ++cc) {
                const v4i C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[ip][jj][m], Bm[cc], zero, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[jj][r] = __builtin_fmaf((float)C[r] * as_f32((int)tw[cc >> 1][r]), dw[cc], acc[jj][r]);
              }
            }
          }
        }
      }
      return;
    } else if constexpr (F::legacy) {
      // ---- 32-element blocks: the unit is eight blocks of row j.  Lane (row j, chunk c) of MFMA m of half q works on block
      //      4 q + 2 m + (c >> 1), h = c & 1:
      //        Q8_0   qs[16 h .. + 15] (a 2-byte aligned ds_read_b128): K order = element order;
      //        nibble formats   qs[8 h .. + 7] (ds_read_b64): low nibbles = elements 8 h .., high nibbles = 16 + 8 h ..
      //                         (+ the fifth bits of qh), which is the order the activation fragments are written in
      //                         (quantize.hip, GGQ_T16_RUN8).
      //      One MFMA covers two blocks (chunks 0-1 | 2-3), which M8 separates by its doubled token rows and the 16-token
      //      form by zeroing the other block's lanes.  Factors as in the reference's tensor-core bodies: Q4_0 / Q5_0 / Q8_0
      //      d d8 C with the offset subtracted from the weights (mmq.cuh:359, 561, 971), Q4_1 / Q5_1 fp16(d d8) C + fp16(m s8)
      //      (:527-529). ----
      typedef unsigned v4u_a2 __attribute__((ext_vector_type(4), aligned(2)));
      typedef unsigned v2u_a2 __attribute__((ext_vector_type(2), aligned(2)));
      struct __attribute__((packed, aligned(2))) u32_a2 { uint32_t v; };
      const uint8_t* blk = wl + L::W + j * SB + s * UB;
      const v4i zero = {0, 0, 0, 0};
      auto sub_bytes = [](uint32_t x) {   // per byte x - sub as int8 (x <= 31): no borrow leaves a byte whose top bit is set
        if constexpr (F::sub == 0) return x;
        else return ((x | 0x80808080u) - 0x01010101u * (uint32_t)F::sub) ^ 0x80808080u;
      };
      auto operand = [&](int bi) {   // the lane's 16 K-values of block bi
        const uint8_t* bp = blk + bi * F::BS;
        if constexpr (!F::nib) {
          const v4u_a2 t = *(const v4u_a2*)(bp + F::BQS + 16 * (c & 1));
          return v4i{(int)t[0], (int)t[1], (int)t[2], (int)t[3]};
        } else {
          const v2u_a2 x = *(const v2u_a2*)(bp + F::BQS + 8 * (c & 1));
          uint32_t v[4] = {x[0] & 0x0F0F0F0Fu, x[1] & 0x0F0F0F0Fu, (x[0] >> 4) & 0x0F0F0F0Fu, (x[1] >> 4) & 0x0F0F0F0Fu};
          if constexpr (F::blk_qh) {   // bit e of qh = fifth bit of element e: elements 8 h .. (byte h), 16 + 8 h .. (byte 2 + h)
            const uint32_t qh = ((const u32_a2*)(bp + F::BQH))->v >> (8 * (c & 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t nb = (qh >> (16 * (i >> 1) + 4 * (i & 1))) & 0xF;
              v[i] |= ((nb * 0x00204081u) & 0x01010101u) << 4;   // bit k -> bit 0 of byte k
            }
          }
          return v4i{(int)sub_bytes(v[0]), (int)sub_bytes(v[1]), (int)sub_bytes(v[2]), (int)sub_bytes(v[3])};
        }
      };
      // acc += the block's term for the token whose scale word is `tw` (fp32 d8 | half2(d8, s8)), C the integer dot,
      // `bw` the block's scale word (fp16 d, or half2(d, m) read as one dword)
      auto apply = [&](float a, int C, uint32_t tw, uint32_t bw) -> float {
        if constexpr (F::blk_m) {
          const h2 pr = __builtin_bit_cast(h2, bw) * __builtin_bit_cast(h2, tw);   // (d d8, m s8), each rounded to fp16
          float fr;
          asm("v_fma_mix_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(fr) : "v"(__builtin_bit_cast(uint32_t, pr)), "v"((float)C));
          return a + fr;
        } else if constexpr (F::d8_half) {
          float t;
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t) : "v"((float)C), "v"(tw));
          return __builtin_fmaf(t, bits_h_f32(bw & 0xFFFF), a);
        } else {
          return __builtin_fmaf((float)C * as_f32((int)tw), bits_h_f32(bw & 0xFFFF), a);
        }
      };
      auto scale_word = [&](int bi) -> uint32_t {
        if constexpr (F::blk_m) return ((const u32_a2*)(blk + bi * F::BS))->v;
        else return *(const uint16_t*)(blk + bi * F::BS);
      };
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        v4i b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) b[m] = operand(4 * q + 2 * m + (c >> 1));
        if constexpr (M8) {
          const uint8_t* tq = wl + L::TAB + s * 256 + q * 128;   // [token quad][group][token]
          const v4u_t d0 = *(const v4u_t*)(tq + (c & 1) * 64 + sel * 16);         // block 4 q + sel, tokens 4 (c & 1) .. + 3
          const v4u_t d1 = *(const v4u_t*)(tq + (c & 1) * 64 + (2 + sel) * 16);   // block 4 q + 2 + sel
          const uint32_t bw0 = scale_word(4 * q + sel), bw1 = scale_word(4 * q + 2 + sel);
          const v4i C0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][0][0], b[0], zero, 0, 0, 0);
          const v4i C1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][0][1], b[1], zero, 0, 0, 0);
          const uint32_t t0[4] = {d0[0], d0[1], d0[2], d0[3]}, t1[4] = {d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0][r] = apply(acc[0][r], C0[r], t0[r], bw0);
            acc[0][r] = apply(acc[0][r], C1[r], t1[r], bw1);
          }
        } else {
          v4i bm[4];   // block 4 q + g: chunks 0-1 of MFMA g >> 1 for even g, chunks 2-3 for odd g
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bm[0][i] = c < 2 ? b[0][i] : 0; bm[1][i] = c < 2 ? 0 : b[0][i];
            bm[2][i] = c < 2 ? b[1][i] : 0; bm[3][i] = c < 2 ? 0 : b[1][i];
          }
          uint32_t bw[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) bw[g] = scale_word(4 * q + g);
#pragma unroll
          for (int jj = 0; jj < NTT; ++jj) {
            const uint8_t* tq = wl + L::TAB + (jj * ((MAXU + 1) / 2) * 2 + s) * 512 + q * 256;
            uint32_t tw[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const v4u_t t = *(const v4u_t*)(tq + c * 64 + g * 16);   // tokens 4 c .. 4 c + 3, block 4 q + g
              tw[g][0] = t[0]; tw[g][1] = t[1]; tw[g][2] = t[2]; tw[g][3] = t[3];
            }
            v4i C[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) C[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][jj][g >> 1], bm[g], zero, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[jj][r] = apply(acc[jj][r], C[g][r], tw[g][r], bw[g]);
          }
        }
      }
      return;
    } else {
    const uint8_t* blk = wl + L::W + j * SB + s * UB;   // row j's block of this unit
    const v4u_t hd = *(const v4u_t*)blk;   // {d | dmin << 16, scales[0..3], scales[4..7], scales[8..11]}
    v4u_t qh = {0, 0, 0, 0};
    if constexpr (F::has_qh) qh = *(const v4u_t*)(blk + off::Q5_K_QH + 16 * (c & 1));
    v4u_t rawq[2];
    rawq[0] = *(const v4u_t*)(blk + F::QS + 16 * c);
    rawq[1] = *(const v4u_t*)(blk + F::QS + 64 + 16 * c);
    // token scales: M8 [unit][half][token quad][group][token], else [token tile][unit][half][token quad][group][token]
    constexpr bool HOIST = false;   // (hoisting every scale word above the first MFMA measured no gain: 86 -> 118 registers)
    uint32_t sw[2][NTT];
    v4u_t dsv[2][NTT][4];
    auto read_scales = [&](int q, int jj) {
      if constexpr (M8) {
        const uint8_t* tq = wl + L::TAB + s * 256 + q * 128;
        sw[q][jj] = *(const uint32_t*)(tq + ((j >> 2) & 1) * 64 + c * 16 + (j & 3) * 4);
        dsv[q][jj][0] = *(const v4u_t*)(tq + (c & 1) * 64 + (2 * sel) * 16);
        dsv[q][jj][1] = *(const v4u_t*)(tq + (c & 1) * 64 + (2 * sel + 1) * 16);
      } else {
        const uint8_t* tq = wl + L::TAB + (jj * ((MAXU + 1) / 2) * 2 + s) * 512 + q * 256;
        sw[q][jj] = *(const uint32_t*)(tq + (j >> 2) * 64 + c * 16 + (j & 3) * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) dsv[q][jj][g] = *(const v4u_t*)(tq + c * 64 + g * 16);   // tokens 4c .. 4c+3, group 4q + g
      }
    };
    if constexpr (HOIST) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < NTT; ++jj) read_scales(q, jj);
    }
    // ---- row scales of the unit's 8 groups: d * sc_g and -(dmin * m_g)  (get_scale_min_k4, dequantize.cuh:154-161) ----
    const uint32_t s0 = hd[1], s1 = hd[2], s2 = hd[3];
    const uint32_t sc4[2] = {s0 & 0x3F3F3F3Fu, (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u)};
    const uint32_t mn4[2] = {s1 & 0x3F3F3F3Fu, ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u)};
    const float dall = bits_h_f32(hd[0] & 0xFFFF), dmin = bits_h_f32(hd[0] >> 16);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const v4u_t raw = rawq[q];
      v4i b_lo, b_hi;   // the lane's 16 elements of group 4q + 2 sel (low nibbles) and 4q + 2 sel + 1 (high nibbles)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t lo = raw[i] & 0x0F0F0F0Fu, hi = (raw[i] >> 4) & 0x0F0F0F0Fu;
        if constexpr (F::has_qh) {   // high bit of element b of group g: bit g of qh[b]
          const uint32_t hsel = qh[i] >> (4 * q + 2 * sel);
          lo |= (hsel & 0x01010101u) << 4;
          hi |= ((hsel >> 1) & 0x01010101u) << 4;
        }
        b_lo[i] = (int)lo; b_hi[i] = (int)hi;
      }
      // min term, B operand of the f32 MFMA below: lane (row j, k = c) holds -(dmin * m) of group 4 q + c
      const float mwc = -(dmin * (float)((mn4[q] >> (8 * c)) & 0xFF));
      const v4i zero = {0, 0, 0, 0};
      if constexpr (M8) {
        if constexpr (!HOIST) read_scales(q, 0);
        const float dw_lo = dall * (float)((sc4[q] >> (16 * sel)) & 0xFF), dw_hi = dall * (float)((sc4[q] >> (16 * sel + 8)) & 0xFF);
        const v4i C_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][0][0], b_lo, zero, 0, 0, 0);
        const v4i C_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][0][1], b_hi, zero, 0, 0, 0);
        // A rows 8-15 are the second group pair's copy of tokens 0-7: their min term is already in rows 0-7
        // min term on the vector ALU (every lane's four registers are real tokens here, so 8 mixed-precision FMAs per half
        // beat the f32 MFMA of the 16-token form: 7.06 -> 6.99 us warm, 9.73 -> 9.49 cold); s8 read as fp16 in place
        const float mw_lo = -(dmin * (float)((mn4[q] >> (16 * sel)) & 0xFF)), mw_hi = -(dmin * (float)((mn4[q] >> (16 * sel + 8)) & 0xFF));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(accm[0][r]) : "v"(dsv[q][0][0][r]), "v"(mw_lo));
          asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(accm[0][r]) : "v"(dsv[q][0][1][r]), "v"(mw_hi));
        }

#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // float(C) * d8 rounded once (v_fma_mix_f32 reads the fp16 d8 in place), then the row scale (mmq.cuh:1274-1363 factors)
          float t0, t1;
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t0) : "v"((float)C_lo[r]), "v"(dsv[q][0][0][r]));
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t1) : "v"((float)C_hi[r]), "v"(dsv[q][0][1][r]));
          acc[0][r] = __builtin_fmaf(t0, dw_lo, acc[0][r]);
          acc[0][r] = __builtin_fmaf(t1, dw_hi, acc[0][r]);
        }
      } else {
        v4i b_a, b_b, b_c, b_d;   // group 4q (low nibbles, lanes c < 2), 4q+1 (high, c < 2), 4q+2 (low, c >= 2), 4q+3 (high, c >= 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!F::has_qh) {   // the lane masks ARE the nibble masks (all ones or zero per lane): no selects
            const uint32_t hi4 = raw[i] >> 4;
            b_a[i] = (int)(raw[i] & m_lo); b_b[i] = (int)(hi4 & m_lo);
            b_c[i] = (int)(raw[i] & m_hi); b_d[i] = (int)(hi4 & m_hi);
          } else {
            b_a[i] = c < 2 ? b_lo[i] : 0; b_b[i] = c < 2 ? b_hi[i] : 0;
            b_c[i] = c < 2 ? 0 : b_lo[i]; b_d[i] = c < 2 ? 0 : b_hi[i];
          }
        }
        float dw[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) dw[g] = dall * (float)((sc4[q] >> (8 * g)) & 0xFF);
#pragma unroll
        for (int jj = 0; jj < NTT; ++jj) {
          if constexpr (!HOIST) read_scales(q, jj);
          v4i C[4];
          C[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][jj][0], b_a, zero, 0, 0, 0);
          C[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][jj][1], b_b, zero, 0, 0, 0);
          C[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][jj][0], b_c, zero, 0, 0, 0);
          C[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fr.a[q][jj][1], b_d, zero, 0, 0, 0);
          accm[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(bits_h_f32(sw[q][jj] >> 16), mwc, accm[jj], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t;
              asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(t) : "v"((float)C[g][r]), "v"(dsv[q][jj][g][r]));
              acc[jj][r] = __builtin_fmaf(t, dw[g], acc[jj][r]);
            }
          }
        }
      }
    }
    }
  };

  // fragments: double-buffered one unit ahead; a single buffer when two token tiles make a unit's fragments 64 registers
  constexpr bool DBUF = NTT == 1;
  Frags FA, FB;
  request_frags(FA, u0);
  if constexpr (DBUF) { if (u0 + 1 < u1) request_frags(FB, u0 + 1); }
  for (int ub = u0; ub < u1; ub += MAXU) {
    request_round(ub);   // (slices longer than MAXU units: the previous round's LDS reads have all returned — their results were used)
    T16_STAMP(1);
    // vmcnt counts loads and LDS-DMA together in issue order: everything requested so far has landed after this wait.
    // (A counted wait per unit would need the image in unit-major order, which costs 1.7x the cache lines per KB.)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), through the builtin so that hipcc's own wait insertion knows
    __builtin_amdgcn_wave_barrier();
    T16_STAMP(2);
    const int ns = min(MAXU, u1 - ub);
    if constexpr (DBUF) {
#pragma unroll 1
      for (int s = 0; s < ns; s += 2) {
        compute_unit(s, FA);
        if (s == 0) T16_STAMP(3);
        if (ub + s + 2 < u1) request_frags(FA, ub + s + 2);   // (wave-uniform: nothing is requested past the slice)
        if (s + 1 < ns) compute_unit(s + 1, FB);
        if (ub + s + 3 < u1) request_frags(FB, ub + s + 3);
      }
    } else {
#pragma unroll 1
      for (int s = 0; s < ns; ++s) {
        compute_unit(s, FA);
        if (s == 0) T16_STAMP(3);
        if (ub + s + 1 < u1) request_frags(FA, ub + s + 1);
      }
    }
  }

  T16_STAMP(4);
#pragma unroll
  for (int jj = 0; jj < NTT; ++jj) acc[jj] += accm[jj];
  // ---- K-slice reduction: sums in slice order, the (token tile, register) pairs dealt round-robin to the waves ----
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the last (unused) fragment prefetch
  __syncthreads();
  float* red = (float*)lds;   // [K-slice][NTT][4][64]
  void* otile = lds + KS * NTT * 1024;   // behind the partial sums: the output tile for the wide stores to peers (<= 2 KB)
#pragma unroll
  for (int jj = 0; jj < NTT; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((ks * NTT + jj) * 4 + r) * 64 + lane] = acc[jj][r];
  __syncthreads();
  T16_STAMP(5);
  const int row = n0 + j;
  for (int p = ks; p < NTT * 4; p += KS) {
    const int jj = p >> 2, r = p & 3;
    float v = 0.0f;
    int t;
    if constexpr (M8) {   // lanes 0-31: token 4 c + r; its second group pair's partial sums sit 32 lanes up
      const int l2 = lane & 31;
      v = red[((0 * NTT + jj) * 4 + r) * 64 + l2] + red[((0 * NTT + jj) * 4 + r) * 64 + l2 + 32];
      for (int sl = 1; sl < KS; ++sl) v += red[((sl * NTT + jj) * 4 + r) * 64 + l2] + red[((sl * NTT + jj) * 4 + r) * 64 + l2 + 32];
      t = lane < 32 ? 4 * c + r : batch;
    } else {
      v = red[((0 * NTT + jj) * 4 + r) * 64 + lane];
      for (int sl = 1; sl < KS; ++sl) v += red[((sl * NTT + jj) * 4 + r) * 64 + lane];
      t = 16 * (tt0 + jj) + 4 * c + r;
    }
    if (t < batch && row < n_rows) {
      const int64_t yi = (int64_t)t * ldy + row;
      const float o = t16_epilogue<DT>(v, epi, aux, yi, row);
      Elem<DT>::st(y, yi, o);
      // peers (go.n_dst > 1): the tile is staged in LDS [token of the tile][16 rows] and leaves in 16-byte stores below — a
      // peer's buffer is uncached memory behind xGMI, where 64 separate 2-byte stores per wave cost 4 x the whole kernel
      if (go.n_dst > 1) Elem<DT>::st(otile, (M8 ? 4 * c + r : 16 * jj + 4 * c + r) * 16 + j, o);
    }
  }
  T16_STAMP(6);
  if (go.n_dst > 1) {   // (kernel-uniform)
    __syncthreads();
    constexpr int ESZ = (int)sizeof(typename Elem<DT>::type), RPC = 16 / ESZ, CPT = 16 / RPC;   // rows per 16-byte chunk, chunks per token
    const int n_tok = M8 ? 8 : 16 * NTT;
    for (int idx = threadIdx.x; idx < n_tok * CPT; idx += blockDim.x) {
      const int tl = idx / CPT, h = idx - tl * CPT;
      const int t = (M8 ? 0 : 16 * tt0) + tl, r0 = n0 + h * RPC;
      if (t >= batch || r0 >= n_rows) continue;
      const int64_t e0 = (int64_t)t * ldy + r0;
      if (r0 + RPC <= n_rows) {
        const v4i v16 = *(const v4i*)((const uint8_t*)otile + (tl * 16 + h * RPC) * ESZ);
        for (int d = 1; d < go.n_dst; ++d) {   // system-coherent write-through store: visible to the peer without a cache write-back
          uint8_t* p = (uint8_t*)go.dst[d] + e0 * ESZ;
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v16) : "memory");
        }
      } else {   // a row tile cut by the end of the matrix: element by element
        for (int e = 0; r0 + e < n_rows; ++e) {
          const float o = Elem<DT>::ld(otile, tl * 16 + h * RPC + e);
          for (int d = 1; d < go.n_dst; ++d) {
            if constexpr (ESZ == 4) __hip_atomic_store((uint32_t*)go.dst[d] + e0 + e, __builtin_bit_cast(uint32_t, o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else {
              uint16_t bits;
              Elem<DT>::st(&bits, 0, o);
              __hip_atomic_store((uint16_t*)go.dst[d] + e0 + e, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
    }
  }
  if (go.n_flag > 0) {   // (kernel-uniform) publish.  Unlike peer_scatter_kernel (64 workgroups, a system-scope release fence each)
    // the ~700 workgroups of a GEMM cannot each write back their XCD's L2 (measured: 170 instead of 40 us): the peer stores above are
    // system-coherent write-through stores, complete when vmcnt reaches zero, so arriving needs no fence; the last workgroup's flag
    // stores carry the one release.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t before = __hip_atomic_fetch_add(go.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == gridDim.x * gridDim.y - 1) {        // last workgroup: every workgroup's release precedes its arrival
        __hip_atomic_store(go.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = 0; d < go.n_flag; ++d) __hip_atomic_store(go.flag[d], go.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

template <int T, int DT, bool M8, int NTT, int MAXU>
static int launch_t16_u(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy, hipStream_t s,
                        Epi16 ep, int64_t ks) {
  constexpr int MAXKS = M8 ? 16 : 12;   // waves per workgroup the kernel is compiled for (M8 fits 128 registers)
  using L = T16Lds<T, M8, NTT, MAXU>;
  const int64_t n_units = k / 256;
  const int64_t n_tt = M8 ? 1 : (batch + 15) / 16;
  constexpr size_t red = (size_t)NTT * 4 * 64 * 4;
  constexpr size_t wave = (size_t)L::WAVE > red ? (size_t)L::WAVE : red;
  constexpr int64_t LDS_KS = (160 * 1024) / wave;   // waves whose LDS areas fit one CU
  if (ks > MAXKS) ks = MAXKS;
  if (ks > LDS_KS) ks = LDS_KS;
  const int64_t ups = (n_units + ks - 1) / ks;   // > MAXU: slices of several LDS rounds
  ks = (n_units + ups - 1) / ups;                // no empty slice
  const size_t lds = (size_t)ks * wave;
  const int64_t gy = (n_tt + NTT - 1) / NTT;
  const int64_t gx = (n + 15) / 16;
  if (gx > 0x7fffffffLL || gy > 65535) return GGQ_ERR_SHAPE;
  if ((uint64_t)n_units * T16Fmt<T>::UB * 16 >= (1ull << 32)) return GGQ_ERR_SHAPE;   // 32-bit byte offsets inside a 16-row tile
  auto kern = mmq_t16_kernel<T, DT, M8, NTT, MAXU, MAXKS>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GGQ_ERR_LAUNCH;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3((unsigned)(64 * ks)), lds, s, (const uint8_t*)w, (const uint8_t*)q8, y,
                     (int)n_units, (int)ups, (int)n, (int)batch, ldy, (int)n_tt, ep.kind, ep.aux, ep.go);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

// K-slices: the fewest units per wave (2, 3 or 4 held in LDS at once) for which every workgroup of the grid is resident
// at once — short slices mean a short per-wave chain of request -> land -> compute -> reduce and more waves to hide it
// (stamps: one wave alone needs 0.84 us to issue a 4-unit slice's memory instructions and 0.45 us per unit), but a second
// round of workgroups costs more than that saves (11008 x 4096, batch 8: 8.1 us with 2-unit slices in two rounds against
// 6.9 us with 4-unit slices in one).  Splitting a resident 4-unit slice into two request -> land -> compute rounds of 2 units
// (half the LDS image, the second round's requests issued after the first round's compute) measured -2 .. -5 % on the kernel at
// batch 16 / 32 and +3 % at batch 8, +-0.2 us on the op: not taken (profiles/r03_t16_two_rounds.txt).
template <int T, int DT, bool M8, int NTT>
static int launch_t16(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy, hipStream_t s,
                      Epi16 ep) {
  constexpr int MAXKS = M8 ? 16 : 12;
  constexpr int64_t WAVES_PER_CU = M8 ? 16 : 12;   // what the instances' register counts admit (4 / 3 waves per SIMD; compiling the
                                                   // two-unit M8 instance for 6 waves per SIMD spills 12 registers: 17.6 instead of 7.0 us)
  const int64_t n_units = k / 256;
  const int64_t n_wg = ((n + 15) / 16) * (M8 ? 1 : ((batch + 15) / 16 + NTT - 1) / NTT);
  // waves of one CU by LDS: the wave-private image of MAXU units + scale tables (the reduction scratch aliases it)
  auto lds_waves = [&](int maxu) -> int64_t {
    const int64_t w2 = T16Lds<T, M8, NTT, 2>::WAVE, w3 = T16Lds<T, M8, NTT, 3>::WAVE, w4 = T16Lds<T, M8, NTT, 4>::WAVE;
    const int64_t wave = maxu == 2 ? w2 : maxu == 3 ? w3 : w4, red = (int64_t)NTT * 4 * 64 * 4;
    return (160 * 1024) / (wave > red ? wave : red);
  };
  // every workgroup resident at once with `ups` units per wave, `maxu` of them in LDS at a time (ups > maxu: several
  // request -> land -> compute rounds per wave — the formats whose weights do not fit the LDS of the chip in one go)
  auto resident = [&](int64_t ups, int maxu) {
    const int64_t ks = (n_units + ups - 1) / ups;
    const int64_t wpc = WAVES_PER_CU < lds_waves(maxu) ? WAVES_PER_CU : lds_waves(maxu);
    if (ups > maxu && (maxu & 1)) return false;   // several rounds: the two fragment buffers alternate by unit parity across rounds
    return ks <= MAXKS && ks <= wpc && n_wg <= 256 * (wpc / ks);
  };
  for (int64_t ups = 2; ups <= 8; ++ups)
    for (int maxu = ups < 4 ? (int)ups : 4; maxu >= 2; --maxu)
      if (resident(ups, maxu)) {
        const int64_t ks = (n_units + ups - 1) / ups;
        if (maxu == 2) return launch_t16_u<T, DT, M8, NTT, 2>(w, q8, y, batch, k, n, ldy, s, ep, ks);
        if (maxu == 3) return launch_t16_u<T, DT, M8, NTT, 3>(w, q8, y, batch, k, n, ldy, s, ep, ks);
        return launch_t16_u<T, DT, M8, NTT, 4>(w, q8, y, batch, k, n, ldy, s, ep, ks);
      }
  // not resident in one go whatever the slicing (very many rows): four-unit slices, as many waves as fit
  if (lds_waves(4) >= 4) return launch_t16_u<T, DT, M8, NTT, 4>(w, q8, y, batch, k, n, ldy, s, ep, (n_units + 3) / 4);
  return launch_t16_u<T, DT, M8, NTT, 2>(w, q8, y, batch, k, n, ldy, s, ep, (n_units + 3) / 4);
}

template <int T, int DT>
static int launch_t16_dt(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy, hipStream_t s,
                         Epi16 ep) {
  if (batch <= 8) return launch_t16<T, DT, true, 1>(w, q8, y, batch, k, n, ldy, s, ep);   // (the scratch layout follows the batch too: quantize.hip)
  if (batch <= 16) return launch_t16<T, DT, false, 1>(w, q8, y, batch, k, n, ldy, s, ep);
  // two token tiles per wave: the K-quants only (the 32-element-block instances spill 70 - 90 registers; ggq_mmq_t16_supported
  // keeps their batches at 16)
  if constexpr (T16Fmt<T>::legacy || T16Fmt<T>::sub16) return GGQ_ERR_SHAPE;
  else return launch_t16<T, DT, false, 2>(w, q8, y, batch, k, n, ldy, s, ep);
}

template <int T>
static int launch_t16_t(const void* w, const void* q8, void* y, int dt, int64_t batch, int64_t k, int64_t n, int64_t ldy,
                        hipStream_t s, Epi16 ep) {
  switch (dt) {
    case GGQ_F32: return launch_t16_dt<T, GGQ_F32>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_F16: return launch_t16_dt<T, GGQ_F16>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_BF16: return launch_t16_dt<T, GGQ_BF16>(w, q8, y, batch, k, n, ldy, s, ep);
    default: return GGQ_ERR_DTYPE;
  }
}

}  // namespace ggq

namespace ggq {
int mul_mat_q_stream_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                          int64_t ldy, int epilogue, const void* aux, void* stream, const void* go);   // mmq.hip
int mul_mat_q_x64_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                       int64_t ldy, int epilogue, const void* aux, void* stream, const void* go);   // mmq_x64.hip
int mul_mat_q_t16_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k,
                       int64_t n_rows, int64_t ldy, int epilogue, const void* aux, void* stream, const void* go);
}
extern "C" int ggq_mul_mat_q_t16(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k,
                                 int64_t n_rows, int64_t ldy, int epilogue, const void* aux, void* stream) {
  return ggq::mul_mat_q_t16_impl(w, q, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream, nullptr);
}

// The 16-token-tile GEMM with the multi-destination write-back (GatherOut): quantise + kernel, for the (type, batch, shape) that
// ggq_mmq_route sends to this kernel; GGQ_ERR_SHAPE otherwise (the caller then runs ggq_mul_mat_q_ld + ggq_peer_scatter).
extern "C" int ggq_mul_mat_q_gather(const void* w, const void* x, void* const* dsts, int n_dst, void* const* flags, int n_flag,
                                    uint32_t generation, void* arrivals, int type, int dtype, int64_t batch, int64_t k,
                                    int64_t n_rows, int64_t ldy, void* scratch, void* stream) {
  using namespace ggq;
  if (n_dst < 1 || n_dst > 8 || n_flag < 0 || n_flag > 8 || !dsts || (n_flag && (!flags || !arrivals))) return GGQ_ERR_ARG;
  if (batch < 0 || k <= 0 || n_rows < 0) return GGQ_ERR_ARG;
  if (batch == 0 || n_rows == 0) return GGQ_ERR_SHAPE;   // nothing would publish the flags
  const int route = ggq_mmq_route(type, batch, k, n_rows);
  if (route != GGQ_MMQ_ROUTE_T16 && route != GGQ_MMQ_ROUTE_STREAM && route != GGQ_MMQ_ROUTE_X64) return ggq_mmq_type_supported(type) ? GGQ_ERR_SHAPE : GGQ_ERR_TYPE;
  GatherOut go{};
  for (int d = 0; d < n_dst; ++d) {
    if (!dsts[d]) return GGQ_ERR_ARG;
    go.dst[d] = dsts[d];
  }
  for (int d = 0; d < n_flag; ++d) {
    if (!flags[d] || ((uintptr_t)flags[d] & 3)) return GGQ_ERR_ARG;
    go.flag[d] = (uint32_t*)flags[d];
  }
  {   // 16-byte stores into the peers' slots: row pitch and bases at 16-byte multiples
    const int64_t esz = dtype == GGQ_F32 ? 4 : 2;
    if ((ldy * esz) % 16) return GGQ_ERR_ALIGN;
    for (int d = 1; d < n_dst; ++d)
      if ((uintptr_t)dsts[d] & 15) return GGQ_ERR_ALIGN;
  }
  go.arrivals = (uint32_t*)arrivals;
  go.generation = generation;
  go.n_dst = n_dst;
  go.n_flag = n_flag;
  if (!w || !x || !scratch) return GGQ_ERR_ARG;
  if (route == GGQ_MMQ_ROUTE_X64) {   // the 64 x 64 wave-tile kernel: a workgroup arrives once its stores have drained
    const int rc = ggq_quantize_q8_1_x64(x, dtype, scratch, batch, k, type, stream);
    if (rc != GGQ_OK) return rc;
    return mul_mat_q_x64_impl(w, scratch, dsts[0], type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream, &go);
  }
  if (route == GGQ_MMQ_ROUTE_STREAM) {   // the streamed kernel: every wave that stores arrives, the last arrival publishes
    const int rc = ggq_quantize_q8_1_tiled(x, dtype, scratch, batch, k, type, stream);
    if (rc != GGQ_OK) return rc;
    return mul_mat_q_stream_impl(w, scratch, dsts[0], type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream, &go);
  }
  const int rc = ggq_quantize_q8_1_t16(x, dtype, scratch, batch, k, type, stream);
  if (rc != GGQ_OK) return rc;
  return mul_mat_q_t16_impl(w, scratch, dsts[0], type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream, &go);
}

int ggq::mul_mat_q_t16_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k,
                            int64_t n_rows, int64_t ldy, int epilogue, const void* aux, void* stream, const void* go) {
  using namespace ggq;
  if (epilogue < GGQ_EPI_NONE || epilogue > GGQ_EPI_SILU_MUL || (epilogue != GGQ_EPI_NONE && !aux)) return GGQ_ERR_ARG;
  if (k <= 0 || n_rows < 0 || batch < 0 || ldy < n_rows) return GGQ_ERR_ARG;
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0 || batch == 0) return GGQ_OK;
  if (!ggq_mmq_t16_supported(type, k, batch)) return ggq_mmq_t16_type_supported(type) ? GGQ_ERR_SHAPE : GGQ_ERR_TYPE;
  if (n_rows > 0x7fffffffLL - 64) return GGQ_ERR_SHAPE;
  if ((type == GGQ_TYPE_Q6_K || type == GGQ_TYPE_Q3_K) && (n_rows * ggq_row_bytes(type, k) < 1024 || n_rows * ggq_row_bytes(type, k) >= (1ll << 32)))
    return GGQ_ERR_SHAPE;   // the shifted copy of the last row starts inside the tensor; 32-bit offsets into the whole tensor
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 15)) return GGQ_ERR_ALIGN;
  const Epi16 ep{epilogue, aux, go ? *(const GatherOut*)go : GatherOut{}};
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
    case GGQ_TYPE_Q4_K: return launch_t16_t<GGQ_TYPE_Q4_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_K: return launch_t16_t<GGQ_TYPE_Q5_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q8_0: return launch_t16_t<GGQ_TYPE_Q8_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q4_0: return launch_t16_t<GGQ_TYPE_Q4_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q4_1: return launch_t16_t<GGQ_TYPE_Q4_1>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_0: return launch_t16_t<GGQ_TYPE_Q5_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_1: return launch_t16_t<GGQ_TYPE_Q5_1>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q6_K: return launch_t16_t<GGQ_TYPE_Q6_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q3_K: return launch_t16_t<GGQ_TYPE_Q3_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    default: return GGQ_ERR_TYPE;
  }
}
