// mmq_x64.hip — quantised GEMM  Y[B,N] = X[B,K] (Q8_1) · W[N,K]^T for batches from 17 tokens up: 64-row x 64-token wave tiles (32 x 32 up to 32 tokens),
// K loop in hand-scheduled gfx950 assembly (generated: scripts/gen_mmq_x64.py -> mmq_x64_loops.inc).
//
// Replaces, for the formats it serves, mul_mat_q's large-tile instances (HK/ggml/mmq.cuh:1917-1986 with mmq_x = 64 ... 128,
// kernel_instances/mmq_kernel.cuh:21-32) and their vec_dot_*_q8_1_mma bodies (mmq.cuh:1247-1363 for Q4_K).  Same numerical contract as
// mmq.hip ("MMQ canon"): exact integer contraction per 32-group, fp16 d8 / s8 for the need_sum formats, s8 (not d8·Σq8)
// in the Q4_K min term; the fp32 accumulation order is this kernel's.
//
// Structure:
//   workgroup = 4 waves = 4 K-slices of one unit (64 weight rows x 64 tokens); no workgroup barrier in the K loop;
//   wave tile = 2 x 2 MFMA tiles (v_mfma_i32_32x32x32_i8), lane = weight row, accumulator register = token: the row scale d·sc is a lane
//     scalar decoded from the row's own header, the token scales come from the scratch as fp32 in accumulator-register order —
//     no scale ever crosses lanes, and both FMA stages are v_pk_fma_f32;
//   two MFMA result sets: the 8 + 16 FMAs of tile n-1 are issued under the MFMA of tile n, unpack / decode / address work sits in
//     the MFMA's issue shadow (scripts/ubench_tile.hip: 50-52 ns per tile per SIMD against 100+ in mmq_stream_kernel);
//   weights: raw super-block bytes, each byte once, row-major by LDS-DMA into a wave-private two-stage ring; the ring's rows are
//     144 bytes apart = an odd number of 16-byte units: every ds_read_b128 of the loop is conflict-free;
//   activations: the x64 scratch layout (quantize.hip LAYOUT 5): per (super-block, 32-token tile) 8 fragments of 1 KB in lane
//     order (straight into registers, one 32-group ahead), the 32 token scales of every group as fp32 in accumulator-register
//     order (by LDS-DMA into a 2 KB table beside the ring, read back as broadcast ds_read_b128: as global loads they made the CU's
//     one texture addresser the limit), the s8 operand of the min-term MFMA;
//   min term Σ_g (-dmin·m_g)[row]·s8_g[token]: ONE v_mfma_f32_32x32x16_f16 per super-block and tile (exact hi + lo fp16 split, as in
//     mmq_stream_kernel), rows with |dmin| > 1024 through a 2^-8-scaled cold pass;
//   K-slice partial sums meet once in LDS and are added in slice order; the write-back reads them back transposed, so that a
//     thread stores 16 consecutive rows of one token (32 contiguous bytes, a token's 64 rows = one 128-byte line);
//   K-slices are interleaved (slice s: super-blocks s, s + KS, ...): the waves of a workgroup read adjacent super-blocks at every step;
//   unit shapes (UR / TT template parameters; ggq_mmq_x64_unit_rows): 64 rows x 64 tokens as above; 96 rows (four two-row-tile waves + four
//     one-row-tile waves) where that makes the launch one even round; 32 rows (one-row-tile waves only) for small launches and for Q5_K,
//     whose 176-byte super-blocks fit the LDS as 32-row stages only; 32 rows x 32 tokens (one MFMA tile per wave and group) up to 32 tokens.
//     Eleven generated loops: {Q4_K, Q8_0, Q4_0} x {2 x 2, 1 x 2, 1 x 1 tiles} and Q5_K x {1 x 2, 1 x 1}; bit-identical per (format, KS).
#include "ggq_common.h"

namespace ggq {

typedef float v32f __attribute__((ext_vector_type(32)));

#ifndef GGQ_X64_LOOPS_INC
#define GGQ_X64_LOOPS_INC "mmq_x64_loops.inc"   // (scripts/build_variant.sh points experiments at another generated file)
#endif
#include GGQ_X64_LOOPS_INC

struct X64Epilogue { int kind; const void* aux; GatherOut go; };
}  // namespace ggq

#ifndef GGQ_X64_TBLK
#define GGQ_X64_TBLK 8          // token tiles per block of the blocked unit order, and the token-tile count it starts at (mmq_x64_kernel; the
#define GGQ_X64_TBLK_FROM 16    // sweep behind both: profiles/r04c_x64_k_slices_large_batch.txt)
#endif
#ifndef GGQ_X64_STAMP
#define GGQ_X64_STAMP 0   // 1: per-wave timestamps (scripts/stamps_x64.py); 0 in every shipped build
#endif
#if GGQ_X64_STAMP
__device__ unsigned long long g_x64_stamps[2048 * 8 * 16];
extern "C" int ggq_debug_read_x64_stamps(void* dst, long long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_x64_stamps), n * 8);
}
// slot i: 100 MHz wall clock; slot 8 + i: the shader clock (s_memtime); slot 7: hardware id (XCC / SE / CU / SIMD)
#define X64_STAMP(i)                                                                          \
  do {                                                                                        \
    if (lane == 0 && blockIdx.x < 2048) {                                                     \
      g_x64_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = wall_clock64();                        \
      g_x64_stamps[(blockIdx.x * 8 + wave) * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime();      \
    }                                                                                         \
  } while (0)
#else
#define X64_STAMP(i) do {} while (0)
#endif

namespace ggq {

template <int DT>
__device__ __forceinline__ float x64_apply_epilogue(float v, int epi, const void* aux, int64_t yi, int row) {
  if (epi == GGQ_EPI_BIAS) return v + Elem<DT>::ld(aux, row);
  if (epi == GGQ_EPI_SILU_MUL) {
    const float g = Elem<DT>::ld(aux, yi);
    return v * (g / (1.0f + expf(-g)));
  }
  return v;
}

constexpr int X64_REC = 10240;          // bytes of one (super-block, 32-token tile) record of the x64 scratch layout
constexpr int X64_STAGE = 64 * 144;     // one ring stage = 64 rows at a 144-byte pitch: a Q4_K super-block, or 128 elements of Q8_0 (136 bytes + 8 of overrun)
constexpr int X64_WAVE_LDS = 2 * X64_STAGE + 2048;   // a wave's weight ring + the fp32 token scales of its two token tiles for one K step
// four K-slices: 80 KB, two workgroups fill a CU's 160 KB exactly; eight K-slices (few units: one workgroup per CU, half the K loop per
// wave): all 160 KB.  The K-slice reduction (16 KB per slice) aliases the rings.
template <int KS> struct X64Lds { static constexpr int BYTES = KS * X64_WAVE_LDS; };
// 96-row units (R3): four waves as above on rows 0-63 + four one-row-tile waves (32 rows x 64 tokens, the same K quarter each) on rows
// 64-95, one workgroup per CU: a launch of 257 .. 512 units of 64 rows that has at most 256 units of 96 is one even round instead of
// "some CUs carry two workgroups" (ggq_mmq_x64_unit_rows; profiles/r04_x64_stamps.txt)
constexpr int X64_STAGE_R1 = 32 * 144;
constexpr int X64_WAVE_LDS_R1 = 2 * X64_STAGE_R1 + 2048;
constexpr int X64_LDS_R3 = 4 * X64_WAVE_LDS + 4 * X64_WAVE_LDS_R1;
// Q5_K: 176-byte super-blocks at a 176-byte pitch (11 chunks), one-row-tile waves only (two stages of 64 rows would be 196 KB for eight waves)
template <int T> struct X64Fmt {
  static constexpr int PITCH = T == GGQ_TYPE_Q5_K ? 176 : 144;      // LDS bytes per row of a stage
  static constexpr int CPR = PITCH / 16, RPI = 63 / CPR;            // 16-byte chunks per row; rows one DMA instruction covers
  static constexpr int STAGE_R1 = 32 * PITCH, WAVE_LDS_R1 = 2 * STAGE_R1 + 2048;
  static constexpr bool ONLY_32 = T == GGQ_TYPE_Q5_K;
  static constexpr bool HAS_T1 = true;   // a one-tile loop (32 rows x 32 tokens per wave) exists for every format of the kernel
};

// UR = weight rows per unit: 64; 96 (R3, above); 32 (U32: every wave a one-row-tile wave — the form for launches with too few 64-row
// units to fill the chip: twice the units, half the work each; ggq_mmq_x64_unit_rows)
// TT = token tiles of 32 per wave: 2; 1 with 32-row units for batches up to 32 tokens (the one-tile loops: no padding to 64 tokens)
template <int T, int DT, int KS, int UR, int TT = 2>
__global__ void __launch_bounds__(UR == 96 ? 512 : 64 * KS, UR == 96 ? 1 : 2) mmq_x64_kernel(const uint8_t* __restrict__ w, const uint8_t* __restrict__ q8,
                                                         void* __restrict__ y, int k, int n_rows, int batch, int64_t ldy,
                                                         int n_tok_tiles, int n_units, int per_xcd, int epi,
                                                         const void* __restrict__ aux, GatherOut go) {
  constexpr bool R3 = UR == 96, U32 = UR == 32;
  static_assert(UR == 32 || UR == 64 || UR == 96, "unit rows");
  static_assert(UR == 32 || !X64Fmt<T>::ONLY_32, "this format has the one-row-tile loop only");
  static_assert(!R3 || KS == 4, "96-row units: four K-slices");
  static_assert(TT == 2 || (TT == 1 && U32 && X64Fmt<T>::HAS_T1), "one token tile: 32-row units, formats with a one-tile loop");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // the units of one XCD are consecutive: a weight tile lives in one L2
  if (unit >= n_units) return;
  constexpr int UROWS = UR;
  // unit -> (row tile, token tile).  Few token tiles: token tiles fastest (the workgroups that share a weight tile run together).  From 16 token
  // tiles on (1024 tokens with 64-token tiles) the order is BLOCKED: eight token tiles at a time, all row tiles under them, token tile fastest
  // inside — the units in flight on an XCD then re-read 8 x 327 KB of activation records (K = 4096) that stay in its 4 MB L2 while the weights
  // stream past once per block, instead of streaming the whole activation scratch (21 MB at 4096 tokens) through L2 once per row tile.
  int row_tile, tok_tile;
  if (n_tok_tiles < GGQ_X64_TBLK_FROM) {
    row_tile = unit / n_tok_tiles;
    tok_tile = unit % n_tok_tiles;
  } else {
    constexpr int TBLK = GGQ_X64_TBLK;
    const int nrt = n_units / n_tok_tiles;
    const int tb = unit / (nrt * TBLK), rem = unit - tb * nrt * TBLK;
    const int width = min(TBLK, n_tok_tiles - tb * TBLK);
    row_tile = rem / width;
    tok_tile = tb * TBLK + rem - row_tile * width;
  }
  const int t0 = tok_tile * (32 * TT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool one_tile = U32 || (R3 && wave >= 4);         // the wave kind with one row tile (R3: rows 64-95 of the unit)
  const int ks = R3 ? (wave & 3) : wave;
  const int n0 = row_tile * UROWS, nb = n0 + (R3 && one_tile ? 64 : 0);
  const int r = lane & 31, h = lane >> 5;
  const int n_sb = k / 256;
  // K-slices are INTERLEAVED: slice ks takes the super-blocks ks, ks + KS, ks + 2 KS ... — at every K step the waves of a workgroup
  // read KS adjacent super-blocks of each row (576 contiguous bytes with four slices of Q4_K: whole 64-byte lines, each fetched once;
  // with blocked slices the 144-byte pieces 2304 bytes apart cost 1.19 x the algorithmic HBM traffic, profiles/r04_traffic.json)
  const int sb_begin = ks, n_own = ks < n_sb ? (n_sb - ks + KS - 1) / KS : 0;
  const uint32_t row_bytes = (uint32_t)(k / Fmt<T>::QK) * Fmt<T>::BS;
  const int valid_rows = min(one_tile ? 32 : 64, n_rows - nb);
  X64_STAMP(0);
#if GGQ_X64_STAMP
  if (lane == 0 && blockIdx.x < 2048) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_x64_stamps[(blockIdx.x * 8 + wave) * 16 + 7] = ((unsigned long long)xcc << 32) | hw;
  }
#endif

  v32f acc0, acc1;
#pragma unroll
  for (int i = 0; i < 32; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }

  if (n_own > 0 && valid_rows > 0) {   // wave-uniform
    v16i magic;
#pragma unroll
    for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
    // weights: the tile's valid rows only — rows past the tensor read as zeros (d = 0: they contribute nothing and are never stored)
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(w + (int64_t)nb * row_bytes), 0, (int)((uint32_t)valid_rows * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q8, 0, (int)0xFFFFFFFFu, 0x00020000);
    // 32-token records per super-block (the scratch layout pads the batch to a multiple of 64 tokens)
    const uint32_t n_tt32 = TT == 2 ? 2u * (uint32_t)n_tok_tiles : 2u * (uint32_t)((batch + 63) / 64);
    const uint32_t sbstride = (uint32_t)KS * n_tt32 * X64_REC;          // scratch bytes from one of this wave's super-blocks to its next
    const uint32_t f0 = ((uint32_t)sb_begin * n_tt32 + (uint32_t)TT * (uint32_t)tok_tile) * X64_REC;
    using XF = X64Fmt<T>;
    const uint32_t ring = (uint32_t)(uintptr_t)lds + (U32 ? (uint32_t)ks * XF::WAVE_LDS_R1
                                                           : one_tile ? 4u * X64_WAVE_LDS + (uint32_t)ks * X64_WAVE_LDS_R1 : (uint32_t)ks * X64_WAVE_LDS);
    // LDS-DMA source offset of this lane inside the seven rows one instruction copies: row lane / 9, 16-byte chunk lane % 9
    // (lane 63 = chunk 0 of the next instruction's first row: both write the same bytes)
    const uint32_t dmaoff = (uint32_t)(lane / XF::CPR) * row_bytes + (uint32_t)(lane % XF::CPR) * 16u;
    const uint32_t ldsd = ring + 2 * (one_tile ? XF::STAGE_R1 : X64_STAGE) + (uint32_t)h * 64u;
    const uint32_t nsb = (uint32_t)n_own;
    X64_STAMP(1);
    if constexpr (T == GGQ_TYPE_Q8_0) {   // 272 bytes of a row per 256 elements, in two 128-element stages at a 144-byte LDS pitch
      const uint32_t hoff = 16u * (uint32_t)h;
      if constexpr (TT == 1)
        x64_loop_q80_t1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 272u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 272u);
      else if (one_tile)
        x64_loop_q80_r1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 272u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 272u);
      else
      x64_loop_q80(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                   sbstride, (uint32_t)sb_begin * 272u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 272u);
    } else if constexpr (T == GGQ_TYPE_Q5_K) {   // one-row-tile waves only; header 16 bytes, qh 32, quants from byte 48
      const uint32_t hoff = 48u + 16u * (uint32_t)h;
      if constexpr (TT == 1)
        x64_loop_q5k_t1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 176u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 176u, 5u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 176u);
      else
      x64_loop_q5k_r1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 176u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                      sbstride, (uint32_t)sb_begin * 176u, 5u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 176u);
    } else if constexpr (T == GGQ_TYPE_Q4_0) {   // 144 bytes of a row per 256 elements (eight 18-byte blocks): one stage; both lane halves read the same bytes
      const uint32_t hoff = 16u * (uint32_t)h;
      if constexpr (TT == 1)
        x64_loop_q40_t1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
      else if (one_tile)
        x64_loop_q40_r1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
      else
        x64_loop_q40(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                     sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
    } else {
      const uint32_t hoff = 16u + 16u * (uint32_t)h;
      if constexpr (TT == 1)
        x64_loop_q4k_t1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
      else if (one_tile)
        x64_loop_q4k_r1(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                        sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
      else
        x64_loop_q4k(acc0, acc1, magic, (uint32_t)lane * 16u, ldsd, ring + (uint32_t)r * 144u + hoff, hoff, dmaoff, wrsrc, arsrc, ring, nsb,
                     sbstride, (uint32_t)sb_begin * 144u, 7u * row_bytes, f0, f0 + X64_REC, f0 + 8192u, f0 + X64_REC + 8192u, f0 + 9216u, (uint32_t)KS * 144u);
    }
  }

  X64_STAMP(2);
  // ---- K-slice reduction: red[slice][tile][register][lane] (R3: the one-tile waves' red1[slice][token tile][register][lane] behind it);
  // the rings are dead once every wave is past its last ds_read ----
  __syncthreads();
  X64_STAMP(3);
  float* red = (float*)lds;
  float* red1 = U32 ? red : red + KS * 4096;
  if (!one_tile) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      red[((ks * 4 + (i >> 4)) * 16 + (i & 15)) * 64 + lane] = acc0[i];
      red[((ks * 4 + 2 + (i >> 4)) * 16 + (i & 15)) * 64 + lane] = acc1[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      red1[((ks * 2 + 0) * 16 + i) * 64 + lane] = acc0[i];
      red1[((ks * 2 + 1) * 16 + i) * 64 + lane] = acc1[i];
    }
  }
  __syncthreads();
  X64_STAMP(4);
  // thread -> (token tl64 of the unit, RPT consecutive rows): lane (row r, half h) of tile (rt, tt) holds token 32 tt + 8 (i >> 2) + 4 h + (i & 3)
  // of row 32 rt + r in register i, so four consecutive rows of one token are four consecutive floats of red[]
  constexpr int NTHR = R3 ? 512 : 64 * KS;
  constexpr int RPT = UROWS * 64 / NTHR;   // rows per thread: 16 (four waves), 8 (eight), 12 (96-row units), 8 / 4 (32-row units)
  const int tl64 = tid / (UROWS / RPT), rb = (tid % (UROWS / RPT)) * RPT;
  const int t = t0 + tl64;
  if (t < batch && (TT == 2 || tl64 < 32)) {
  const int tt = tl64 >> 5, tl = tl64 & 31;
  const int i_reg = 4 * (tl >> 3) + (tl & 3), hh = (tl >> 2) & 1;
  float v[RPT];
#pragma unroll
  for (int j = 0; j < RPT / 4; ++j) {
    const int R = rb + 4 * j, rt = R >> 5, rr = R & 31;
    v4f s;
    if (U32 || (R3 && rt == 2)) {
      s = *(const v4f*)(red1 + (((0 * 2 + tt) * 16 + i_reg) * 64 + 32 * hh + rr));
#pragma unroll
      for (int sl = 1; sl < KS; ++sl) s += *(const v4f*)(red1 + (((sl * 2 + tt) * 16 + i_reg) * 64 + 32 * hh + rr));
    } else {
      s = *(const v4f*)(red + (((0 * 4 + 2 * tt + rt) * 16 + i_reg) * 64 + 32 * hh + rr));
#pragma unroll
      for (int sl = 1; sl < KS; ++sl) {
        const v4f p = *(const v4f*)(red + (((sl * 4 + 2 * tt + rt) * 16 + i_reg) * 64 + 32 * hh + rr));
        s += p;   // fixed slice order ((0 + 1) + 2) + 3 ...
      }
    }
    v[4 * j] = s[0]; v[4 * j + 1] = s[1]; v[4 * j + 2] = s[2]; v[4 * j + 3] = s[3];
  }
  const int row0 = n0 + rb;
  const int64_t yi0 = (int64_t)t * ldy + row0;
  if (epi != GGQ_EPI_NONE) {   // wave-uniform
#pragma unroll
    for (int e = 0; e < RPT; ++e)
      if (row0 + e < n_rows) v[e] = x64_apply_epilogue<DT>(v[e], epi, aux, yi0 + e, row0 + e);
  }
  // 16-byte stores of eight fp16 / bf16 rows; the 12- and 4-row threads (96-row units; 32-row units with eight slices) store 8 bytes at a time
  const bool vec_ok = DT != GGQ_F32 && (ldy & 7) == 0 && ((uintptr_t)y & 15) == 0 && row0 + RPT <= n_rows;
  if (vec_ok) {
    uint32_t pk[RPT / 2];
#pragma unroll
    for (int e = 0; e < RPT / 2; ++e) {
      uint16_t lo, hi;
      if (DT == GGQ_F16) {
        lo = __builtin_bit_cast(uint16_t, (_Float16)v[2 * e]);
        hi = __builtin_bit_cast(uint16_t, (_Float16)v[2 * e + 1]);
      } else {
        lo = Elem<GGQ_BF16>::cvt(v[2 * e]);
        hi = Elem<GGQ_BF16>::cvt(v[2 * e + 1]);
      }
      pk[e] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    if constexpr (RPT == 12 || RPT == 4) {
      v2i* dst = (v2i*)((uint16_t*)y + yi0);
#pragma unroll
      for (int c = 0; c < RPT / 4; ++c) dst[c] = v2i{(int)pk[2 * c], (int)pk[2 * c + 1]};
      for (int d = 1; d < go.n_dst; ++d)   // (kernel-uniform) the peers' slots: system-coherent write-through stores
#pragma unroll
        for (int c = 0; c < RPT / 4; ++c) {
          const v2i val{(int)pk[2 * c], (int)pk[2 * c + 1]};
          asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"((uint16_t*)go.dst[d] + yi0 + 4 * c), "v"(val) : "memory");
        }
    } else {
      v4i* dst = (v4i*)((uint16_t*)y + yi0);
#pragma unroll
      for (int c = 0; c < RPT / 8; ++c) dst[c] = v4i{(int)pk[4 * c], (int)pk[4 * c + 1], (int)pk[4 * c + 2], (int)pk[4 * c + 3]};
      for (int d = 1; d < go.n_dst; ++d)
#pragma unroll
        for (int c = 0; c < RPT / 8; ++c) {
          const v4i val{(int)pk[4 * c], (int)pk[4 * c + 1], (int)pk[4 * c + 2], (int)pk[4 * c + 3]};
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"((uint16_t*)go.dst[d] + yi0 + 8 * c), "v"(val) : "memory");
        }
    }
  } else {
#pragma unroll
    for (int e = 0; e < RPT; ++e)
      if (row0 + e < n_rows) {
        Elem<DT>::st(y, yi0 + e, v[e]);
        gather_store<DT>(go, yi0 + e, v[e]);
      }
  }
  }   // t < batch
  X64_STAMP(5);
  if (go.n_flag > 0) {   // (kernel-uniform) multi-destination launch: the workgroup's stores drain (write-through: complete at vmcnt 0), it
    // arrives; the last of the launch's n_units arrivals writes the flags — the one release at system scope (as mmq_t16.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t before = __hip_atomic_fetch_add(go.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (uint32_t)n_units - 1u) {
        __hip_atomic_store(go.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = 0; d < go.n_flag; ++d) __hip_atomic_store(go.flag[d], go.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

template <int T, int DT, int KS, int UR, int TT = 2>
static int launch_x64_inst(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy, hipStream_t s,
                           const X64Epilogue& ep, int64_t n_tok_tiles, int64_t n_units) {
  constexpr int LDS = UR == 96 ? X64_LDS_R3 : UR == 32 ? KS * X64Fmt<T>::WAVE_LDS_R1 : X64Lds<KS>::BYTES;
  constexpr int NTHR = UR == 96 ? 512 : 64 * KS;
  const int64_t per_xcd = (n_units + 7) / 8;
  auto kern = mmq_x64_kernel<T, DT, KS, UR, TT>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return GGQ_ERR_LAUNCH;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(NTHR), LDS, s, (const uint8_t*)w, (const uint8_t*)q8, y, (int)k, (int)n,
                     (int)batch, ldy, (int)n_tok_tiles, (int)n_units, (int)per_xcd, ep.kind, ep.aux, ep.go);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T, int DT>
static int launch_x64(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy, hipStream_t s,
                      X64Epilogue ep) {
  // -DGGQ_TUNING builds only (scripts/sweep_x64.py): GGQ_X64_KS forces 4 or 8 K-slices, GGQ_X64_ROWS 32- / 64- / 96-row units
  static const char* e = GGQ_TUNING_ENV("GGQ_X64_KS");
  static const char* er = GGQ_TUNING_ENV("GGQ_X64_ROWS");
  if constexpr (X64Fmt<T>::HAS_T1) {
    static const char* et = GGQ_TUNING_ENV("GGQ_X64_TT1");   // tuning builds: the one-tile units up to this many tokens
    if (et ? batch <= atoi(et) : ggq_mmq_x64_tile_tokens(T, batch, k, n) == 32) {   // 32-token tiles: 32-row units of one MFMA tile per wave and group
      const int64_t n_tt = (batch + 31) / 32;
      const int64_t n_units = ((n + 31) / 32) * n_tt;
      if (n_units > 0x7fffffffLL - 8) return GGQ_ERR_SHAPE;
      const int ks = e && (e[0] == '4' || e[0] == '8') ? e[0] - '0' : (n_units <= 256 && k >= 8 * 256 ? 8 : 4);
      return ks == 8 ? launch_x64_inst<T, DT, 8, 32, 1>(w, q8, y, batch, k, n, ldy, s, ep, n_tt, n_units)
                     : launch_x64_inst<T, DT, 4, 32, 1>(w, q8, y, batch, k, n, ldy, s, ep, n_tt, n_units);
    }
  }
  const int64_t n_tok_tiles = (batch + 63) / 64;
  const int unit_rows = er && (er[0] == '3' || er[0] == '6' || er[0] == '9') ? (er[0] == '9' ? 96 : er[0] == '3' ? 32 : 64)
                                                                             : ggq_mmq_x64_unit_rows(T, batch, k, n);
  const int64_t n_units = ((n + unit_rows - 1) / unit_rows) * n_tok_tiles;
  if (n_units > 0x7fffffffLL - 8) return GGQ_ERR_SHAPE;
  if constexpr (X64Fmt<T>::ONLY_32) {
    if (unit_rows != 32) return GGQ_ERR_SHAPE;   // (ggq_mmq_x64_unit_rows returns 32 for these formats; a tuning override must too)
  } else {
    if (unit_rows == 96) return launch_x64_inst<T, DT, 4, 96>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units);
  }
  // at most one unit per CU: eight K-slices per unit (two waves per SIMD either way, half the K loop per wave); otherwise four, two
  // workgroups per CU (ggq_mmq_x64_k_slices, csrc/core/traits.cpp, host-testable; for 32-row units the same rule on their count)
  // From 2048 units (one full round of the chip's 2048 wave slots — eight one-wave workgroups per CU): ONE K-slice, i.e. a wave walks the whole
  // K of its unit and nothing is reduced across waves.  A unit's fixed part (first stage, K-slice reduction, write-back) is then paid once per
  // 16 super-blocks instead of once per 4: 11008 x 4096 at 2048 / 4096 tokens 322 / 546 -> 274 / 492 us warm, 290 / 549 -> 257 / 484 cold;
  // 4096 x 11008 294 / 556 -> 266 / 502; below a full round it loses (1024 units: 155 -> 203) — profiles/r04c_x64_k_slices_large_batch.txt.
  // 64-row units (the 2 x 2-tile loop) only: with 32-row units — eight one-wave workgroups of the one-row-tile loop per CU — single 32 x 32 tiles
  // differed from run to run (Q4_K and Q5_K alike, 4 - 11 of 11 repeats; four-wave workgroups of the same loop: 0 of 11; the 2 x 2 loop with one
  // slice: 0 of 198 — scripts/stress_x64_repro.py, profiles/r04c_x64_k_slices_large_batch.txt): unexplained, so Q5_K keeps four slices.
  const int ks = e && (e[0] == '4' || e[0] == '8' || e[0] == '1') ? e[0] - '0'
                 : n_units >= 2048 && unit_rows == 64 ? 1 : (n_units <= 256 && k >= 8 * 256 ? 8 : 4);
  if (unit_rows == 32)
    return ks == 8 ? launch_x64_inst<T, DT, 8, 32>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units)
#ifdef GGQ_TUNING
         : ks == 1 ? launch_x64_inst<T, DT, 1, 32>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units)   // (the experiment above)
#endif
                   : launch_x64_inst<T, DT, 4, 32>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units);
  if constexpr (!X64Fmt<T>::ONLY_32) {
    if (ks == 1) return launch_x64_inst<T, DT, 1, 64>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units);
    return ks == 8 && k >= 8 * 256 ? launch_x64_inst<T, DT, 8, 64>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units)
                                   : launch_x64_inst<T, DT, 4, 64>(w, q8, y, batch, k, n, ldy, s, ep, n_tok_tiles, n_units);
  }
  return GGQ_ERR_SHAPE;
}

template <int T>
static int launch_x64_dt(const void* w, const void* q8, void* y, int dt, int64_t batch, int64_t k, int64_t n, int64_t ldy,
                         hipStream_t s, X64Epilogue ep) {
  switch (dt) {
    case GGQ_F32: return launch_x64<T, GGQ_F32>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_F16: return launch_x64<T, GGQ_F16>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_BF16: return launch_x64<T, GGQ_BF16>(w, q8, y, batch, k, n, ldy, s, ep);
    default: return GGQ_ERR_DTYPE;
  }
}

}  // namespace ggq

namespace ggq {
int mul_mat_q_x64_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                       int64_t ldy, int epilogue, const void* aux, void* stream, const void* go);
}

extern "C" int ggq_mul_mat_q_x64(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k,
                                 int64_t n_rows, int64_t ldy, int epilogue, const void* aux, void* stream) {
  return ggq::mul_mat_q_x64_impl(w, q, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream, nullptr);
}

// go != nullptr: with the multi-destination write-back (ggq_mul_mat_q_gather)
int ggq::mul_mat_q_x64_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                            int64_t ldy, int epilogue, const void* aux, void* stream, const void* go) {
  using namespace ggq;
  if (epilogue < GGQ_EPI_NONE || epilogue > GGQ_EPI_SILU_MUL || (epilogue != GGQ_EPI_NONE && !aux)) return GGQ_ERR_ARG;
  if (k <= 0 || n_rows < 0 || batch < 0 || ldy < n_rows) return GGQ_ERR_ARG;
  if (!ggq_mmq_x64_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % 256 || n_rows > 0x7fffffffLL - 64) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0 || batch == 0) return GGQ_OK;
  if (!ggq_mmq_x64_supported(type, k, batch)) return GGQ_ERR_SHAPE;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 15)) return GGQ_ERR_ALIGN;
  const X64Epilogue ep{epilogue, aux, go ? *(const GatherOut*)go : GatherOut{}};
  switch (type) {
    case GGQ_TYPE_Q4_K: return launch_x64_dt<GGQ_TYPE_Q4_K>(w, q, y, dtype, batch, k, n_rows, ldy, (hipStream_t)stream, ep);
    case GGQ_TYPE_Q8_0: return launch_x64_dt<GGQ_TYPE_Q8_0>(w, q, y, dtype, batch, k, n_rows, ldy, (hipStream_t)stream, ep);
    case GGQ_TYPE_Q4_0: return launch_x64_dt<GGQ_TYPE_Q4_0>(w, q, y, dtype, batch, k, n_rows, ldy, (hipStream_t)stream, ep);
    case GGQ_TYPE_Q5_K: return launch_x64_dt<GGQ_TYPE_Q5_K>(w, q, y, dtype, batch, k, n_rows, ldy, (hipStream_t)stream, ep);
    default: return GGQ_ERR_TYPE;
  }
}
