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
// Three kernels, all VALU-issue-bound rather than MFMA- or HBM-bound (every (row, token, 32-group) triple needs
// its own float scale: 2 vector ops per triple, 3 with an fp32 d8):
//   mmq_stream_kernel  what ggq_mul_mat_q runs from batch 5 (Q6_K: 33, Q8_0: 65) on the fragment-major scratch:
//                      no workgroup barrier in the K loop (see its header comment);
//   mmq_kernel         barrier-coupled LDS-tile kernel on the reference-layout scratch (ggq_mul_mat_q_prequant,
//                      and the mid-size batches of Q6_K / Q8_0): 8 waves = TBn token blocks x (8/TBn) K-slices per
//                      256-element slab; weights global -> registers -> unpacked int8 LDS tile ([row][256+16]:
//                      conflict-free ds_read_b128), activations by LDS-DMA, one barrier per slab;
//   mmq_small_kernel   batch <= 8, HBM-bound regime: dot4 against LDS-resident activations, no MFMA tiles.
// Shared by all three:
//   * per group: v_mfma_i32_32x32x32_i8 with the accumulator input preset to 0x4B400000, so the
//     int32 result read as a float is 12582912 + C exactly — no v_cvt; for fp16 activation scales
//     (need_sum formats) one fma(Df, d8, -12582912·d8) yields float(C)·d8 bit-exactly, a second
//     fma applies the row scale;
//   * Q4_K/Q5_K min term Σ m·s8 runs on the matrix pipe: one v_mfma_f32_32x32x2_f32 per group pair
//     accumulates straight into the fp32 accumulators;
//   * unpack_raw<T>: one definition of the integer operands and float scales per format (MMQ canon).
#include "ggq_common.h"

#ifndef GGQ_NO_SBMIN
#define GGQ_SBMIN 1   // Q4_K / Q5_K: scales + min term once per super-block (fp16 MFMA), see mmq_stream_kernel
#else
#define GGQ_SBMIN 0
#endif
#ifndef GGQ_NO_XT
#define GGQ_XT 1   // fp32-d8 formats: lane = weight row / register = token at every batch (2 instead of 3 vector ops per triple)
#else
#define GGQ_XT 0
#endif
#ifndef GGQ_SMALL_RW
#define GGQ_SMALL_RW 3   // rows in flight per wave in the dot4 kernel (batch <= 4): 3 = a wave's share of 11008 rows; 2 measured slower than 1
#endif
#ifndef GGQ_ABL
#define GGQ_ABL 0   // 32: per-wave timestamps in the streamed kernel (scripts/stamps_mmq.py); 0 in every shipped build
#endif

#if GGQ_ABL & 32
__device__ unsigned long long g_stamps[8192 * 8];
extern "C" int ggq_debug_read_stamps(void* dst, long long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n * 8);
}
#define GGQ_STAMP(i)                                                                             \
  do {                                                                                           \
    if (lane == 0 && blockIdx.x < 1024) g_stamps[(blockIdx.x * 8 + ks) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define GGQ_STAMP(i) do {} while (0)
#endif

#include "mmq_unpack.h"
#include <type_traits>

namespace ggq {


// LDS carve-up of one stage buffer (bytes); all offsets multiples of 16
template <int T, int TBn> struct MmqLds {
  static constexpr int TT = 32 * TBn;                      // tokens per workgroup tile
  static constexpr int W_TILE = 32 * WROW;                 // one int8 weight tile (32 rows)
  static constexpr int W_BYTES = W_TILE * (MmqTraits<T>::two_tiles ? 2 : 1);
  static constexpr int S_BYTES = MmqTraits<T>::n_scale * 8 * 32 * 4;
  static constexpr int A_BYTES = 2 * TT * 144;             // one 256-element K slab of activations
  static constexpr int W_OFF = 0;
  static constexpr int S_OFF = W_OFF + W_BYTES;
  static constexpr int A_OFF = S_OFF + S_BYTES;
  static constexpr int STAGE = A_OFF + A_BYTES;
  static constexpr int RED_BYTES = 8 * 16 * 64 * 4;        // K-slice reduction (aliases the stages)
  static constexpr int BYTES = 2 * STAGE > RED_BYTES ? 2 * STAGE : RED_BYTES;
  static constexpr int A_CHUNKS = 2 * TT * 9;              // 16-byte chunks of one activation slab
  static constexpr int N_DMA = (A_CHUNKS + 255) / 256;     // LDS-DMA instructions per copy wave per slab
};

#define GGQ_LDS_BARRIER()                                   \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

// Workgroup = 8 waves.  Waves 0-3 stage the weights (global -> registers one slab ahead ->
// unpack -> LDS), waves 4-7 stream the activation slab with LDS-DMA (global_load_lds_dwordx4:
// no VGPRs, no ds_write); all 8 waves compute.  Wave (tb, kq): 32-token block tb, K-slice kq.
template <int T, int DT, int TBn>
__global__ void __launch_bounds__(512, MmqTraits<T>::light && TBn <= 2 ? 4 : 2) mmq_kernel(const uint8_t* __restrict__ w,
                                                  const uint8_t* __restrict__ q8,
                                                  void* __restrict__ y, int k, int n_rows, int batch,
                                                  int64_t ldy, int n_tok_tiles) {
  using L = MmqLds<T, TBn>;
  using TR = MmqTraits<T>;
  constexpr int TT = L::TT;
  constexpr int KQ = 8 / TBn;   // K-slice waves per token block
  constexpr int GPW = TBn;      // groups per wave per slab
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tb = wave % TBn, kq = wave / TBn;
  const int unit = blockIdx.x;
  const int n0 = (unit / n_tok_tiles) * 32;   // first weight row of the tile
  const int t0 = (unit % n_tok_tiles) * TT;   // first token of the tile
  const int n_valid_tok = min(TT, batch - t0);
  const int64_t row_bytes = (int64_t)(k / Fmt<T>::QK) * Fmt<T>::BS;
  const int n_groups = k / 32;          // 32-element groups along K
  const int n_slabs = (k + 255) / 256;  // 256-element K slabs
  const bool w_wave = wave < 4;

  // ---- weight staging role (waves 0-3): row sr, group sg of a slab ----
  const int sr = (tid >> 3) & 31, sg = tid & 7;
  const uint8_t* srow = w + (int64_t)min(n0 + sr, n_rows - 1) * row_bytes;
  auto load_w = [&](Raw& R, int s) {
    load_raw<T>(srow, min(min(s, n_slabs - 1) * 8 + sg, n_groups - 1), R);  // clamped, never predicated
  };
  auto write_w = [&](const Raw& R, int s, uint8_t* st) {
    uint32_t wq[8], wq2[8];
    float s0 = 0.0f, s1 = 0.0f;
    unpack_raw<T>(R, s * 8 + sg, wq, wq2, s0, s1);
    if (s * 8 + sg >= n_groups) {  // K tail of a legacy format: contributes nothing
      s0 = 0.0f; s1 = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { wq[i] = 0; wq2[i] = 0; }
    }
    uint8_t* dst = st + L::W_OFF + sr * WROW + 32 * sg;
    *(v4i*)dst = v4i{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
    *(v4i*)(dst + 16) = v4i{(int)wq[4], (int)wq[5], (int)wq[6], (int)wq[7]};
    if constexpr (TR::two_tiles) {
      *(v4i*)(dst + L::W_TILE) = v4i{(int)wq2[0], (int)wq2[1], (int)wq2[2], (int)wq2[3]};
      *(v4i*)(dst + L::W_TILE + 16) = v4i{(int)wq2[4], (int)wq2[5], (int)wq2[6], (int)wq2[7]};
    }
    float* sdst = (float*)(st + L::S_OFF) + sg * 32 + sr;
    sdst[0] = s0;
    if constexpr (TR::n_scale == 2) sdst[256] = s1;
  };

  // ---- activation streaming role (waves 4-7): DMA instruction j of a slab moves chunks 64j..64j+63 ----
  int64_t dma_src[L::N_DMA];   // byte offset of this lane's chunk relative to slab 0, or -1
#pragma unroll
  for (int i = 0; i < L::N_DMA; ++i) {
    const int j = i * 4 + (wave - 4);
    const int c = 64 * j + lane;
    const int kb = c >= TT * 9 ? 1 : 0;
    int within = c - kb * TT * 9;
    if (within >= n_valid_tok * 9) within = within % 9;  // token outside the batch: any valid bytes do
    dma_src[i] = (w_wave || c >= L::A_CHUNKS) ? -1 : ((int64_t)kb * batch + t0) * 144 + 16 * within;
  }
  const int64_t slab_stride = (int64_t)2 * batch * 144;
  auto dma_acts = [&](int s, uint8_t* st) {
#pragma unroll
    for (int i = 0; i < L::N_DMA; ++i) {
      const int j = i * 4 + (wave - 4);
      if (dma_src[i] >= 0)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(q8 + s * slab_stride + dma_src[i]),
            (__attribute__((address_space(3))) void*)(st + L::A_OFF + 1024 * j), 16, 0, 0);
    }
  };

  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = (int)MAGIC_I;

  const int r = lane & 31, h = lane >> 5;
  const int tl = tb * 32 + r;  // token within the tile

  auto compute = [&](const uint8_t* st) {
    const uint8_t* wt = st + L::W_OFF;
    const float* sc = (const float*)(st + L::S_OFF);
    const uint8_t* at = st + L::A_OFF;
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
      const int g = kq * GPW + j;
      const uint8_t* ablk = at + ((g >> 2) * TT + tl) * 144;
      const uint32_t dsw = *(const uint32_t*)(ablk + 4 * (g & 3));
      float bs, bm = 0.0f;
      if constexpr (TR::need_sum) { bs = bits_h_f32(dsw & 0xFFFF); bm = bits_h_f32(dsw >> 16); }
      else bs = as_f32((int)dsw);
      const float nmbs = -(MAGIC_F * bs);  // exact when bs is an fp16 value (need_sum formats)

      v16i c0, c1 = magic;
      if constexpr (TR::half_scales) {
        const long a0 = *(const long*)(wt + r * WROW + 32 * g + 8 * h);
        const long a1 = *(const long*)(wt + r * WROW + 32 * g + 16 + 8 * h);
        const long b0 = *(const long*)(ablk + 16 + 32 * (g & 3) + 8 * h);
        const long b1 = *(const long*)(ablk + 16 + 32 * (g & 3) + 16 + 8 * h);
        c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a0, b0, magic, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, b1, magic, 0, 0, 0);
      } else {
        const v4i a = *(const v4i*)(wt + r * WROW + 32 * g + 16 * h);
        const v4i b = *(const v4i*)(ablk + 16 + 32 * (g & 3) + 16 * h);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, magic, 0, 0, 0);
        if constexpr (TR::two_tiles) {
          const v4i a2 = *(const v4i*)(wt + L::W_TILE + r * WROW + 32 * g + 16 * h);
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b, magic, 0, 0, 0);
        }
      }
      if constexpr (TR::mfma_min) {
        // Σ_g (-dmin·m)[row,g] · s8[token,g] for the group pair (g & ~1, g | 1): K = 2 outer products
        if ((j & 1) == 0) {
          const int gp = GPW >= 2 ? g + h : g;
          const float am = (GPW >= 2 || h == 0) ? sc[256 + gp * 32 + r] : 0.0f;
          const uint32_t dsp = *(const uint32_t*)(at + ((gp >> 2) * TT + tl) * 144 + 4 * (gp & 3));
          const float bmp = (GPW >= 2 || h == 0) ? bits_h_f32(dsp >> 16) : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bmp, acc, 0, 0, 0);
        }
      }
      // accumulator register i <-> tile row (i&3) + 8(i>>2) + 4h : 4 runs of 4 consecutive rows
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const v4f sa = *(const v4f*)(sc + g * 32 + 8 * qd + 4 * h);
        v4f sb = {0, 0, 0, 0};
        if constexpr (TR::n_scale == 2 && !TR::mfma_min) sb = *(const v4f*)(sc + 256 + g * 32 + 8 * qd + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * qd + e;
          const float df0 = as_f32(c0[i]);  // = 12582912 + C exactly
          // explicit fma + -ffp-contract=off: every accumulator register sees the same
          // instruction sequence, so a row's result does not depend on its position in the tile
          if constexpr (TR::fp16_prod) {  // Q4_1/Q5_1, mmq.cuh:527-529 / :840-842
            const float lo = (float)((_Float16)sa[e] * (_Float16)bs);
            const float hi = (float)((_Float16)sb[e] * (_Float16)bm);
            acc[i] += __builtin_fmaf(lo, df0 - MAGIC_F, hi);
          } else if constexpr (TR::two_tiles) {  // Q2_K: d8 (dall·Σsc q q8 − dmin·Σ m q8), mmq.cuh:47
            const float df1 = as_f32(c1[i]);
            acc[i] = __builtin_fmaf(bs, __builtin_fmaf(sa[e], df0 - MAGIC_F, -(sb[e] * (df1 - MAGIC_F))), acc[i]);
          } else if constexpr (TR::half_scales) {  // Q6_K (fp32 d8): mmq.cuh:1726-1732
            const float df1 = as_f32(c1[i]);
            acc[i] = __builtin_fmaf((df0 - MAGIC_F) * bs, sa[e], acc[i]);
            acc[i] = __builtin_fmaf((df1 - MAGIC_F) * bs, sb[e], acc[i]);
          } else if constexpr (TR::need_sum) {  // Q4_0, Q4_K, Q5_K (fp16 d8): float(C)·d8 in one exact fma
            acc[i] = __builtin_fmaf(__builtin_fmaf(df0, bs, nmbs), sa[e], acc[i]);
          } else {  // Q5_0 / Q8_0 / Q3_K (fp32 d8): d_w d8 C
            acc[i] = __builtin_fmaf((df0 - MAGIC_F) * bs, sa[e], acc[i]);
          }
        }
      }
    }
  };

  // ---- K loop: slab s in stage (s&1); during compute(s) the next slab is written / DMA'd into the
  //      other stage, whose previous contents (slab s-1) every wave finished before the last barrier ----
  uint8_t* st0 = lds;
  uint8_t* st1 = lds + L::STAGE;
  Raw R;
  if (w_wave) {
    load_w(R, 0);
    write_w(R, 0, st0);
    load_w(R, 1);
  } else {
    dma_acts(0, st0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  GGQ_LDS_BARRIER();
  for (int s = 0; s < n_slabs; ++s) {
    uint8_t* cur = (s & 1) ? st1 : st0;
    uint8_t* nxt = (s & 1) ? st0 : st1;
    const bool more = s + 1 < n_slabs;
    if (w_wave) {
      if (more) write_w(R, s + 1, nxt);   // registers were loaded during the previous slab
      load_w(R, s + 2);
    } else if (more) {
      dma_acts(s + 1, nxt);
    }
    compute(cur);
    if (!w_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GGQ_LDS_BARRIER();
  }

  // ---- K-slice reduction through LDS ----
  if constexpr (KQ > 1) {
    float* red = (float*)lds;  // [wave][16][64]; the stage buffers are dead after the last barrier
    if (kq > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    }
    GGQ_LDS_BARRIER();
    if (kq == 0) {
#pragma unroll
      for (int s = 1; s < KQ; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[((s * TBn + tb) * 16 + i) * 64 + lane];
    }
  }

  // ---- write back: lane = token, register i = row (i&3) + 8(i>>2) + 4h ----
  if (kq == 0) {
    const int t = t0 + tl;
    if (t < batch) {
      const bool vec_ok = DT != GGQ_F32 && (ldy & 3) == 0 && ((uintptr_t)y & 7) == 0 && n0 + 32 <= n_rows;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int row = n0 + 8 * qd + 4 * h;
        if (vec_ok) {
          uint16_t hv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (DT == GGQ_F16) hv[e] = __builtin_bit_cast(uint16_t, (_Float16)acc[4 * qd + e]);
            else hv[e] = Elem<GGQ_BF16>::cvt(acc[4 * qd + e]);
          }
          uint2 pk;
          pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
          pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
          *(uint2*)((uint16_t*)y + (int64_t)t * ldy + row) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < n_rows) Elem<DT>::st(y, (int64_t)t * ldy + row + e, acc[4 * qd + e]);
        }
      }
    }
  }
}

template <int T, int DT, int TBn>
static int launch_mmq_cfg(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                          int64_t ldy, hipStream_t s) {
  using L = MmqLds<T, TBn>;
  auto kern = mmq_kernel<T, DT, TBn>;
  // per call, like the reference (mmq.cuh:2022): the attribute is per device, and the op may run on any of them
  if (L::BYTES > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES) != hipSuccess)
    return GGQ_ERR_LAUNCH;
  const int64_t n_tok_tiles = (batch + L::TT - 1) / L::TT;
  const int64_t n_units = ((n + 31) / 32) * n_tok_tiles;
  if (n_units > 0x7fffffffLL) return GGQ_ERR_SHAPE;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)n_units), dim3(512), L::BYTES, s, (const uint8_t*)w,
                     (const uint8_t*)q8, y, (int)k, (int)n, (int)batch, ldy, (int)n_tok_tiles);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

// ---------------------------------------------------------------------------------------------
// Small-batch kernel (batch <= 16): HBM-bound, so no MFMA tiles — the weight stream is read
// exactly once with coalesced per-lane block loads (lane = one 32-element group of a row,
// consecutive lanes = consecutive blocks), each group is unpacked once with the same unpack_raw
// as the MFMA path (identical integer operands and scales: MMQ canon) and dotted against every
// token with v_dot4_i32_i8; the block_q8_1_mmq activations of all tokens sit in LDS.
// A 64-lane "folding" butterfly reduces the NTOK per-lane sums with NTOK/2+... shuffles.
// ---------------------------------------------------------------------------------------------
template <int T, int DT, int NTOK>
__global__ void __launch_bounds__(1024) mmq_small_kernel(const uint8_t* __restrict__ w,
                                                        const uint8_t* __restrict__ q8,
                                                        void* __restrict__ y, int k, int n_rows, int batch,
                                                        int64_t ldy, int rows_per_wave) {
  using TR = MmqTraits<T>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int n_groups = k / 32;
  constexpr int GP = 48;  // LDS pitch of one 32-byte activation group: conflict-free 64-lane ds_read_b128
  int8_t* xq = (int8_t*)lds;                                  // [NTOK][n_groups][GP]
  uint32_t* xds = (uint32_t*)(lds + (size_t)NTOK * n_groups * GP);  // [NTOK][n_groups] (d,s) / float d words

  // ---- stage activations: block (kb, t) of 144 B -> xq[t][4 kb + i][..], xds[t][4 kb + i] ----
  const int n_kb = (k + 127) / 128;
  for (int c = threadIdx.x; c < n_kb * NTOK * 9; c += 1024) {
    const int blk = c / 9, part = c - blk * 9;
    const int kb = blk / NTOK, t = blk - kb * NTOK;
    const int ts = min(t, batch - 1);  // tokens beyond the batch: duplicate (results discarded)
    const v4i v = *(const v4i*)(q8 + ((int64_t)kb * batch + ts) * 144 + 16 * part);
    if (part == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * kb + i < n_groups) xds[t * n_groups + 4 * kb + i] = (uint32_t)v[i];
    } else {
      const int g = 4 * kb + ((part - 1) >> 1);
      if (g < n_groups) *(v4i*)(xq + ((size_t)t * n_groups + g) * GP + 16 * ((part - 1) & 1)) = v;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 16 + (threadIdx.x >> 6);
  const int64_t row_bytes = (int64_t)(k / Fmt<T>::QK) * Fmt<T>::BS;
  const int row_end = min(n_rows, (wave + 1) * rows_per_wave);

  // RW rows of the wave in flight at once: the raw bytes of all of them are requested before the first dot product
  // (streamed from HBM a wave with one 16-34-byte load per lane in flight is bound by the round trip, not by bandwidth).
  // Q4_K 11008 x 4096, op incl. the quantise launch: batch 2 10.2 -> 9.0 us warm / 12.6 -> 11.8 cold, batch 4 13.1 -> 10.9 / 15.0 -> 12.9.
  constexpr int RW = GGQ_SMALL_RW;
  const int nsteps = (n_groups + 63) / 64;
  const int row_begin = wave * rows_per_wave;
  for (int r0 = row_begin; r0 < row_end; r0 += RW) {
    float acc[RW][NTOK];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int t = 0; t < NTOK; ++t) acc[r][t] = 0.0f;
    for (int us = 0; us < nsteps; ++us) {
      const int G = lane + 64 * us;
      const int Gc = min(G, n_groups - 1);   // clamped, never predicated
      Raw R[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) load_raw<T>(w + (int64_t)min(r0 + r, row_end - 1) * row_bytes, Gc, R[r]);
      if (G < n_groups) {
        uint32_t wq[RW][8], wq2[RW][8];
        float s0[RW], s1[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) unpack_raw<T>(R[r], G, wq[r], wq2[r], s0[r], s1[r]);
#pragma unroll
        for (int t = 0; t < NTOK; ++t) {
          const v4i a0 = *(const v4i*)(xq + ((size_t)t * n_groups + G) * GP);
          const v4i a1 = *(const v4i*)(xq + ((size_t)t * n_groups + G) * GP + 16);
          const uint32_t dsw = xds[t * n_groups + G];
          float bs, bm = 0.0f;
          if constexpr (TR::need_sum) { bs = bits_h_f32(dsw & 0xFFFF); bm = bits_h_f32(dsw >> 16); }
          else bs = as_f32((int)dsw);
#pragma unroll
          for (int r = 0; r < RW; ++r) {
            int c0 = 0, c1 = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              c0 = __builtin_amdgcn_sdot4((int)wq[r][i], a0[i], c0, false);
              if constexpr (TR::half_scales) c1 = __builtin_amdgcn_sdot4((int)wq[r][4 + i], a1[i], c1, false);
              else c0 = __builtin_amdgcn_sdot4((int)wq[r][4 + i], a1[i], c0, false);
              if constexpr (TR::two_tiles) {
                c1 = __builtin_amdgcn_sdot4((int)wq2[r][i], a0[i], c1, false);
                c1 = __builtin_amdgcn_sdot4((int)wq2[r][4 + i], a1[i], c1, false);
              }
            }
            // same float combinations as the MFMA kernels (explicit FMAs, -ffp-contract=off)
            if constexpr (TR::fp16_prod) {
              const float lo = (float)((_Float16)s0[r] * (_Float16)bs);
              const float hi = (float)((_Float16)s1[r] * (_Float16)bm);
              acc[r][t] += __builtin_fmaf(lo, (float)c0, hi);
            } else if constexpr (TR::two_tiles) {
              acc[r][t] = __builtin_fmaf(bs, __builtin_fmaf(s0[r], (float)c0, -(s1[r] * (float)c1)), acc[r][t]);
            } else if constexpr (TR::half_scales) {
              acc[r][t] = __builtin_fmaf((float)c0 * bs, s0[r], acc[r][t]);
              acc[r][t] = __builtin_fmaf((float)c1 * bs, s1[r], acc[r][t]);
            } else if constexpr (TR::mfma_min) {
              acc[r][t] = __builtin_fmaf((float)c0 * bs, s0[r], acc[r][t]);
              acc[r][t] = __builtin_fmaf(s1[r], bm, acc[r][t]);
            } else {
              acc[r][t] = __builtin_fmaf((float)c0 * bs, s0[r], acc[r][t]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      if (r0 + rr >= row_end) break;   // wave-uniform
      const int row = r0 + rr;
      // ---- folding butterfly: after the steps with n > 1 values, lane l holds token (l >> shift) ----
      float v[NTOK];
#pragma unroll
      for (int t = 0; t < NTOK; ++t) v[t] = acc[rr][t];
      int n = NTOK, m = 32;
#pragma unroll
      for (; n > 1; n >>= 1, m >>= 1) {
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < NTOK / 2; ++j) {
          if (j < n / 2) {
            const float keep = upper ? v[j + n / 2] : v[j];
            const float give = upper ? v[j] : v[j + n / 2];
            v[j] = keep + __shfl_xor(give, m, 64);
          }
        }
      }
      float r = v[0];
#pragma unroll
      for (; m > 0; m >>= 1) r += __shfl_xor(r, m, 64);
      // the token kept at each fold is selected by the lane bit, MSB first => token = lane / (64 / NTOK)
      constexpr int LPT = 64 / NTOK;
      const int t = lane / LPT;
      if ((lane % LPT) == 0 && t < batch) Elem<DT>::st(y, (int64_t)t * ldy + row, r);
    }
  }
}

template <int T, int DT, int NTOK>
static int launch_mmq_small_n(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                              int64_t ldy, hipStream_t s) {
  const size_t lds = (size_t)NTOK * (k / 32) * 48 + (size_t)NTOK * (k / 32) * 4;
  if (lds > 160 * 1024) return -100;  // caller falls back to the tiled kernel
  auto kern = mmq_small_kernel<T, DT, NTOK>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GGQ_ERR_LAUNCH;
  }
  // one 16-wave workgroup per CU: the activation staging (tens of KB) is paid once per CU
  int rpw = (int)((n + 256 * 16 - 1) / (256 * 16));
  rpw = rpw <= 1 ? 1 : (rpw + GGQ_SMALL_RW - 1) / GGQ_SMALL_RW * GGQ_SMALL_RW;   // whole groups of rows in flight
  const int64_t waves = (n + rpw - 1) / rpw;
  const int64_t grid = (waves + 15) / 16;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), lds, s, (const uint8_t*)w, (const uint8_t*)q8, y,
                     (int)k, (int)n, (int)batch, ldy, rpw);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T, int DT>
static int launch_mmq_small(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                            int64_t ldy, hipStream_t s) {
  if (batch <= 2) return launch_mmq_small_n<T, DT, 2>(w, q8, y, batch, k, n, ldy, s);
  if (batch <= 4) return launch_mmq_small_n<T, DT, 4>(w, q8, y, batch, k, n, ldy, s);
  return launch_mmq_small_n<T, DT, 8>(w, q8, y, batch, k, n, ldy, s);
}

template <int T, int DT>
static int launch_mmq_t(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                        int64_t ldy, hipStream_t s) {
  if (batch <= 8) {  // HBM-bound regime: stream the weights once, dot4 against every token
    static const char* e = GGQ_TUNING_ENV("GGQ_MMQ_SMALL");
    if (!e || e[0] != '0') {
      const int rc = launch_mmq_small<T, DT>(w, q8, y, batch, k, n, ldy, s);
      if (rc != -100) return rc;
    }
  }
  if (batch <= 32) return launch_mmq_cfg<T, DT, 1>(w, q8, y, batch, k, n, ldy, s);
  // 64-token units give the dispatcher 2x more, smaller units to balance (688 vs 344 at the headline
  // shape); 128-token units halve the weight re-staging once there are plenty of units anyway
  const int64_t units128 = ((n + 31) / 32) * ((batch + 127) / 128);
  if (batch <= 64 || units128 < 2048 || !MmqTraits<T>::light) {
    return launch_mmq_cfg<T, DT, 2>(w, q8, y, batch, k, n, ldy, s);
  }
  return launch_mmq_cfg<T, DT, 4>(w, q8, y, batch, k, n, ldy, s);
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
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (k > (1 << 30) || n_rows > 0x7fffffffLL - 64 || batch > 0x7fffffffLL / 256) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0 || batch == 0) return GGQ_OK;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 15)) return GGQ_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
#ifdef GGQ_DEV_ONLY   // development builds (scripts/build_variant.sh -DGGQ_DEV_ONLY=<type id>): one instantiation, seconds instead of minutes
  return GGQ_ERR_TYPE;
#else
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
#endif
}

namespace ggq {
// ---------------------------------------------------------------------------------------------
// Streamed kernel (all formats, fragment-major activations).  No workgroup barrier in the K loop:
//   workgroup = 4 waves = 4 K-slices of one unit (32 weight rows x 32·TB tokens); every wave is an
//   independent instruction stream, so a SIMD interleaves its waves freely (the barrier-coupled kernel
//   above spends more time at barriers than computing).  The K loop is VALU-issue bound: per
//   (row, token, 32-group) triple two FMAs (three with an fp32 d8) — DESIGN.md §5.4.
//   * weights: a wave copies the raw bytes of its 32 rows x one K stage (256 elements, 128 for the wide
//     legacy blocks) as 16-byte chunks, consecutive lanes = consecutive chunks of a row (per-lane row loads
//     would touch one 128-byte L1 line per 16 useful bytes: measured, they cost as much L1 time as all
//     activations), into a wave-private two-stage LDS ring, one stage ahead: the HBM latency of the
//     weight stream is off the critical path.  (Plain loads + ds_write, not LDS-DMA: with DMA operations
//     in flight the compiler's waitcnt pass degrades every later vector-memory wait to vmcnt(0).)
//   * A fragments.  Q4_K/Q5_K: lane (r, h) reads bytes 16h..16h+15 of the pair's 32 nibble bytes from its
//     row in the ring: low nibbles are its fragment of group 2p, high nibbles of group 2p+1.  Other
//     formats: the lane unpacks the whole group 2p+h (the same unpack_raw as the kernels above: MMQ canon)
//     and one v_permlane32_swap per dword hands the two lane halves their K-halves of both groups.
//   * the per-(row, group) scale must be seen per accumulator register: 64 floats per pair go through a
//     wave-private LDS line (one ds_write_b32, broadcast ds_read_b128) — same wave, no barrier.
//   * activations: fragment-major tiles (LAYOUT 2 of quantize.hip): one B fragment = 1 KB contiguous in
//     lane order; a fragment's registers are reloaded for the next pair right after the MFMA that consumed
//     them was issued (a full iteration of lead, no second register set).
//   * K-slice partial sums meet once per unit in LDS.  blockIdx -> unit is XCD-aware: the units of one XCD
//     are consecutive, so a weight row tile is fetched into one L2 only.
// ---------------------------------------------------------------------------------------------
// Fused epilogues of the streamed kernel's write-back (ggq_mul_mat_q_epi; the reference's op has none):
//   GGQ_EPI_BIAS      y = acc + bias[row]                          aux: bias, n_rows elements of the output dtype
//   GGQ_EPI_SILU_MUL  y = silu(gate[token, row]) * acc             aux: gate, same [batch, ldy] layout as y
// applied to the fp32 accumulator before the one rounding to the output dtype.
struct Epilogue { int kind = GGQ_EPI_NONE; const void* aux = nullptr; GatherOut go = GatherOut{}; };
template <int DT>
__device__ __forceinline__ float apply_epilogue(float v, int epi, const void* aux, int64_t yi, int row) {
  if (epi == GGQ_EPI_BIAS) return v + Elem<DT>::ld(aux, row);
  if (epi == GGQ_EPI_SILU_MUL) {
    const float g = Elem<DT>::ld(aux, yi);
    return v * (g / (1.0f + expf(-g)));
  }
  return v;
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int T> struct StreamCfg {
  using TR = MmqTraits<T>;
  static constexpr bool direct = T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K;
  static constexpr int QK = Fmt<T>::QK, BS = Fmt<T>::BS;
#ifdef GGQ_NO_SPLIT_RING
  static constexpr bool split = false;
#else
  // Legacy 32-element formats: the ring holds a row's stage DE-INTERLEAVED — the quant bytes of its blocks back to
  // back at 16-byte multiples, the block headers (d / dm / d + qh) behind them — instead of the raw 18/20/22/24/34-
  // byte blocks: every fragment read of the K loop is then an aligned ds_read_b128 (a misaligned one costs 56
  // instead of 16 cycles, scripts/ubench_lds_align.hip, and Q8_0 was bound by them).  The copy pays for it with
  // 2-byte-aligned global loads (one 16-byte piece of one block's quants per lane) and one small header load.
  static constexpr bool split = QK == 32 && T != GGQ_TYPE_Q4_0;   // (Q4_0: measured neutral, 31.1 vs 30.6 us: its 18-byte blocks keep the raw ring)
#endif
  static constexpr int QSB = T == GGQ_TYPE_Q8_0 ? 32 : 16;                 // quant bytes per block
  static constexpr int QS_OFF = BS - QSB;                                  // ... which end the block
  static constexpr int HB = (T == GGQ_TYPE_Q5_0 || T == GGQ_TYPE_Q5_1) ? 8 : 4;   // header bytes kept per block (a dword or two from the block start)
  static constexpr int SE = QK == 256 ? 256 : split ? ((HB == 8 || QSB == 32) ? 128 : 256) : (8 * BS <= 176 ? 256 : 128);   // elements of K per ring stage
  static constexpr int NB = SE / 32;                                       // blocks of a row in one stage (split form)
  static constexpr int TPR = NB * (QSB / 16);                              // 16-byte copy tasks per row
  static constexpr int ROW = NB * QSB + NB * HB;                           // de-interleaved row: quants, then headers
  static constexpr int SPITCH = (ROW / 16) % 2 ? ROW : ROW + 16;           // odd number of 16-byte units: conflict-free ds_read_b128
  static constexpr int SNW = 32 * TPR / 64;                                // copy windows (64 tasks each) per stage
  static constexpr int SEG = SE / QK * BS;               // bytes of one row in one stage
  static constexpr int IPS = SE / 64;                    // pair-iterations per stage
  static constexpr int CPR = (SEG + 15) / 16;            // 16-byte chunks per row (the last may overlap)
  // LDS row pitch.  A misaligned ds_read/write_b128 costs 3.4x an aligned one (measured: 56 vs 16 cycles), so
  // super-block formats (one block per row per stage, fields at 16-byte multiples) get a 16-byte-multiple
  // pitch; the 2-4 tail bytes of a 210/110/84-byte block are parked with a narrow store.  The legacy
  // formats keep their 18/20/22/24/34-byte blocks back to back (their quant bytes are misaligned anyway).
  static constexpr int TAIL = QK == 256 ? SEG % 16 : 0;  // bytes of the last, partial chunk (0: none)
  static constexpr int PITCH = split ? SPITCH : TAIL ? 16 * CPR : SEG;
  // copy windows: one 64-lane load moves up to 64 / CPR rows; the window count is rounded up to a power of two
  // so that the windows tile the 32 rows exactly (no overrun rows: a stage is 32 x PITCH bytes, which lets two
  // eight-wave workgroups of 32-token units share a CU's 160 KB)
  static constexpr int NW0 = (32 + 64 / CPR - 1) / (64 / CPR);
  static constexpr int NWP = NW0 <= 1 ? 1 : NW0 <= 2 ? 2 : NW0 <= 4 ? 4 : NW0 <= 8 ? 8 : NW0 <= 16 ? 16 : 32;
  // ... unless that costs an extra load per iteration (Q8_0: 5 windows over 2 iterations = 3 per iteration, 8
  // would be 4: measured 41 -> 47 us); then the windows keep their natural size and the stage holds the overrun rows
  static constexpr bool TILING = split || (NWP + IPS - 1) / IPS == (NW0 + IPS - 1) / IPS;
  static constexpr int NW = split ? SNW : TILING ? NWP : NW0;
  static constexpr int RPW = split ? 64 / TPR : TILING ? 32 / NW : 64 / CPR;   // rows per copy window
  static constexpr int WPI = (NW + IPS - 1) / IPS;          // windows per iteration
  static constexpr int STAGE = (((TILING ? 32 : NW * RPW) * PITCH + (TILING ? 0 : 16) + 15) / 16) * 16;
  static constexpr int SBUF = 1024;                      // two copies (pair parity) of [s0 | s1][group of the pair][row] floats
  static constexpr int WAVE = 2 * STAGE + SBUF;
  // three workgroups per CU (168 VGPRs, <= 53 KB LDS) except Q6_K: two result tiles per group and a 210-byte row
  static constexpr int OCC = TR::half_scales ? 2 : 3;
};

// dynamic LDS of a workgroup with KS K-slices (the K-slice reduction aliases the rings) and how many such
// workgroups the kernel is compiled to co-reside per CU
template <int T, int TB, int KS> struct StreamLaunch {
  static constexpr int RED = KS * TB * 16 * 64 * 4;   // (transposed small-batch variant: fewer registers, same bound)
  // per-super-block scales + fp16 min term (SBMIN, see the kernel): needs a 512-byte s8 stash per token block and wave;
  // not for the eight-slice / 32-token instance, whose two workgroups fill the CU's 160 KB exactly
  static constexpr bool SBMIN_OK = GGQ_SBMIN && StreamCfg<T>::direct && !(KS == 8 && TB == 1);
  static constexpr int WAVE = StreamCfg<T>::WAVE + (SBMIN_OK ? TB * 512 : 0);
  static constexpr int LDS = KS * WAVE > RED ? KS * WAVE : RED;
  // eight slices: 64-token units need ~235 VGPRs (one workgroup per CU); 32-token units stay under 128 and two
  // workgroups share a CU when their rings fit
  static constexpr int WG_PER_CU = KS == 4 ? StreamCfg<T>::OCC
                                           : (TB == 1 && StreamCfg<T>::OCC == 3 && 2 * LDS <= 160 * 1024 ? 2 : 1);
  static_assert(LDS * WG_PER_CU <= 160 * 1024, "LDS of the resident workgroups");
  // (the second __launch_bounds__ argument of hipcc is waves per SIMD: WG_PER_CU * KS / 4)
};

// KS = K-slices = waves per workgroup: 4 normally; 8 when there are too few units to give every SIMD three waves
// NR = 0: lane = token, accumulator register = weight row (all 16 registers live).
// NR = 4 / 8 (batch <= 8 / 16, TB = 1): the MFMA operands are swapped — accumulator register = token, lane = weight
//   row — so only the NR registers that hold real tokens are scaled: the vector work per 32-group drops from 32 FMAs
//   to 2·NR.  The weight scale is then a lane scalar (one v_permlane32_swap hands both lane halves the scales of both
//   groups of the pair: no LDS exchange), the token scales are per register and come from a wave-private LDS line the
//   token lanes fill (already converted to fp32).
template <int T, int DT, int TB, int KS, int NR>
__global__ void __launch_bounds__(64 * KS, (StreamLaunch<T, TB, KS>::WG_PER_CU * KS / 4)) mmq_stream_kernel(const uint8_t* __restrict__ w,
                                                            const uint8_t* __restrict__ q8,
                                                            void* __restrict__ y, int k, int n_rows, int batch,
                                                            int64_t ldy, int n_tok_tiles, int n_units, int per_xcd,
                                                            int epi, const void* __restrict__ aux, GatherOut go) {
  using C = StreamCfg<T>;
  using TR = MmqTraits<T>;
  constexpr int SEG = C::SEG, STAGE = C::STAGE, IPS = C::IPS;
  static_assert(NR == 0 || (TB == 1 && (NR == 4 || NR == 8)), "transposed variant: one token block, 4 or 8 live registers");
  // Q4_K / Q5_K, lane = token: everything that is per super-block is done once per stage (= super-block) instead of
  // once per pair — the 8 (scale, min) pairs are decoded with packed byte arithmetic, the row scales of all 8 groups go
  // to the wave's LDS line in one go, and the min term  Σ_g (-dmin·m_g)[row] · s8_g[token]  is ONE
  // v_mfma_f32_32x32x16_f16 per token block (K = 8 groups x {hi, lo}: dmin·m_g has 17 significant bits and is split
  // exactly into two fp16 values; s8 is an fp16 value already; fp16 subnormals are not flushed by the MFMA — checked in
  // scripts/ubench_f16mfma.hip) instead of four v_mfma_f32_32x32x2_f32: 17 ns instead of 108 ns of matrix pipe per
  // super-block and token block, and the SIMD does not overlap MFMAs with vector ops (scripts/ubench_overlap.hip).
  constexpr bool SBMIN = StreamLaunch<T, TB, KS>::SBMIN_OK && NR == 0;
  // Formats whose token scale d8 is a full fp32 value (Q5_0, Q8_0, Q3_K, Q6_K: need_sum = false) cannot use the exact
  // two-op apply with lane = token: 12582912·d8 is not representable, so float(C)·d8 costs a subtract and a multiply.
  // With the MFMA operands swapped (lane = weight row, register = token, as in the small-batch variant but for all 16
  // registers and every token block) the LANE scalar is the row scale d·sc, which has at most 19 significant bits:
  // fma(12582912 + C, s, -12582912·s) = RN(C·s) exactly, then one fma with the token's d8 (a per-register value read
  // from the wave's LDS line): 2 vector ops per (row, token, 32-group) triple instead of 3 (Q6_K: 4 instead of 6).
  // Measured at batch 128, 11008 x 4096: Q6_K 58.8 -> 56.2 us, Q8_0 42.7 -> 41.7; Q5_0 unchanged and Q3_K 6 % slower (their
  // loops are bound by the weight copy, not by the apply), so only the first two take it.
  constexpr bool XT = GGQ_XT && NR == 0 && KS == 4 && (T == GGQ_TYPE_Q8_0 || T == GGQ_TYPE_Q6_K);   // (the eight-slice instances spill with it)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [wave]{ ring[2][STAGE]; float sb[2][2][32] }

  const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (unit >= n_units) return;
  // unit -> (row tile, token tile): token tile fastest; from 16 token tiles on BLOCKED, eight token tiles at a time under all row tiles, so that
  // the activation tiles of an XCD's units in flight stay in its L2 instead of the whole scratch streaming through once per 32-row tile
  // (as mmq_x64.hip; profiles/r04c_x64_k_slices_large_batch.txt)
  int row_tile, tok_tile;
  if (n_tok_tiles < 16) {
    row_tile = unit / n_tok_tiles;
    tok_tile = unit % n_tok_tiles;
  } else {
    const int nrt = n_units / n_tok_tiles;
    const int tb = unit / (nrt * 8), rem = unit - tb * nrt * 8;
    const int width = min(8, n_tok_tiles - tb * 8);
    row_tile = rem / width;
    tok_tile = tb * 8 + rem - row_tile * width;
  }
  const int n0 = row_tile * 32;
  const int t0 = tok_tile * 32 * TB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t row_bytes = (uint32_t)(k / C::QK) * C::BS;
  const int n_groups = k / 32;
  const int n_st = (k + C::SE - 1) / C::SE;
  const int st_begin = (int)((int64_t)ks * n_st / KS), st_end = (int)((int64_t)(ks + 1) * n_st / KS);
  uint8_t* ring = lds + ks * StreamLaunch<T, TB, KS>::WAVE;
  float* sb0 = (float*)(ring + 2 * STAGE);   // scale exchange line, double-buffered by pair parity
  uint32_t* s8l = (uint32_t*)(ring + C::WAVE);   // SBMIN: s8 stash [token block][token][pair of the super-block]
  const int n_tt32 = (batch + 31) / 32;
  const uint32_t lane16 = lane * 16;
  const uint8_t* wtile = w + (int64_t)n0 * row_bytes;
  GGQ_STAMP(0);

  // ---- weight stage copy: window m = rows [m·RPW, (m+1)·RPW) of the stage, lane = chunk lane % CPR of row
  //      lane / CPR (lanes past RPW·CPR and rows past 31 repeat valid bytes into unused ring space).  The last
  //      chunk of a row ends exactly at the row's stage bytes (it may overlap its neighbour), so no load
  //      ever leaves the weight tensor. ----
  const int lrow = min(lane / C::CPR, C::RPW - 1), lchunk = lane % C::CPR;
  const int rmax = min(31, n_rows - 1 - n0);
  // byte offset of this lane's chunk inside the row's stage bytes; super-block formats have no K tail
  const uint32_t lcoff = (uint32_t)min(16 * lchunk, (int)SEG - 16);
  auto chunk_off = [&](int st_src) {
    if constexpr (C::QK == 256) return lcoff;
    else return (uint32_t)min(16 * lchunk, min((int)SEG, (int)(row_bytes - (uint32_t)st_src * SEG)) - 16);
  };
  // per-lane constants + scalar (window, stage) terms: no vector multiply in the loop
  const uint32_t lrow_off = (uint32_t)lrow * row_bytes, rmax_off = (uint32_t)rmax * row_bytes;
  const uint32_t lds_lane = (uint32_t)lrow * C::PITCH;
  const uint32_t w_lane = lrow_off + lcoff;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wtile, 0, (int)((uint32_t)(rmax + 1) * row_bytes), 0x00020000);
  // split form: task lane -> (row lane / TPR of the window, block, 16-byte piece of its quants)
  const int s_lrow = lane / C::TPR, s_blk = (lane % C::TPR) / (C::QSB / 16), s_part = lane % (C::QSB / 16);
  const int n_blk_row = k / 32;
  struct Win { u32x4_a2 q; u32x2_a2 h; };
  auto load_window = [&](int st_src, int m) {
    if constexpr (C::split) {
      Win wv;
      const int row = min(m * C::RPW + s_lrow, rmax);                       // clamped to the tile's last valid row
      const int blk = min(st_src * C::NB + s_blk, n_blk_row - 1);            // K tail: a valid block (zeroed by the n_groups test)
      const uint8_t* b = wtile + ((uint32_t)row * row_bytes + (uint32_t)blk * C::BS);
      wv.q = ld_u32x4(b + C::QS_OFF + 16 * s_part);
      if constexpr (C::HB == 8) wv.h = ld_u32x2(b);
      else { wv.h.v[0] = ld_u32(b); wv.h.v[1] = 0; }
      return wv;
    } else {
    Win wv;
    if constexpr (C::QK == 256) {
      // descriptor over the tile's valid rows: rows past the tensor's last row read as zeros (never stored)
      const v4u t = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)w_lane, (int)((uint32_t)(m * C::RPW) * row_bytes + (uint32_t)st_src * SEG), 0);
      wv.q = u32x4_a2{{t[0], t[1], t[2], t[3]}};
      return wv;
    }
    const uint32_t sbase = (uint32_t)(m * C::RPW) * row_bytes + (uint32_t)st_src * SEG;   // scalar
    const uint32_t srmax = rmax_off + (uint32_t)st_src * SEG;                             // scalar
    // row m·RPW + lrow, clamped to the tile's last valid row
    const uint32_t roff = lrow <= rmax - m * C::RPW ? sbase + lrow_off : srmax;
    wv.q = ld_u32x4(wtile + (roff + chunk_off(st_src)));
    return wv;
    }
  };
  auto store_window = [&](const Win& wv, int st_src, int buf, int m) {
    if constexpr (C::split) {
      uint8_t* drow = ring + (uint32_t)(buf * STAGE + (m * C::RPW + s_lrow) * C::PITCH);
      *(v4i*)(drow + s_blk * C::QSB + 16 * s_part) = v4i{(int)wv.q.v[0], (int)wv.q.v[1], (int)wv.q.v[2], (int)wv.q.v[3]};
      if (s_part == 0) {
        if constexpr (C::HB == 8) *(v2i*)(drow + C::NB * C::QSB + 8 * s_blk) = v2i{(int)wv.h.v[0], (int)wv.h.v[1]};
        else *(uint32_t*)(drow + C::NB * C::QSB + 4 * s_blk) = wv.h.v[0];
      }
      return;
    }
    const u32x4_a2& v = wv.q;
    uint8_t* dst = ring + (uint32_t)(buf * STAGE + m * C::RPW * C::PITCH) + lds_lane;
    if constexpr (C::TAIL != 0) {
      // aligned 16-byte chunks; the last chunk was loaded ending at the block end: its final TAIL bytes
      // (fp16 d / half2 dm / Q3_K scales + d) go to offset 16 (CPR - 1) with narrow stores
      static_assert(C::TAIL == 2 || C::TAIL == 4 || C::TAIL == 14, "tail store widths");
      if (lchunk < C::CPR - 1) *(v4i*)(dst + 16 * lchunk) = v4i{(int)v.v[0], (int)v.v[1], (int)v.v[2], (int)v.v[3]};
      else if constexpr (C::TAIL == 2) *(uint16_t*)(dst + 16 * (C::CPR - 1)) = (uint16_t)(v.v[3] >> 16);
      else if constexpr (C::TAIL == 4) *(uint32_t*)(dst + 16 * (C::CPR - 1)) = v.v[3];
      else {   // Q3_K: 14 bytes = scales (12) + d (2): bytes 2..15 of the chunk loaded at SEG - 16
        *(uint16_t*)(dst + 16 * (C::CPR - 1)) = (uint16_t)(v.v[0] >> 16);
        *(uint32_t*)(dst + 16 * (C::CPR - 1) + 2) = v.v[1];   // 2-byte aligned dword stores: rare lanes only
        *(uint32_t*)(dst + 16 * (C::CPR - 1) + 6) = v.v[2];
        *(uint32_t*)(dst + 16 * (C::CPR - 1) + 10) = v.v[3];
      }
    } else {
      *(u32x4_a2*)(dst + chunk_off(st_src)) = v;
    }
  };

  // ---- activations (LAYOUT 2 tiles of 4608 bytes per 128 elements x 32 tokens): abase[] points at the
  //      current pair's 2 KB of fragments; its half2(d, sum) / float d pairs sit 4096 (even pair) or
  //      2304 (odd pair) bytes further ----
  typedef const __attribute__((address_space(1))) uint8_t* gptr;   // keeps the loads global_load (not flat)
  typedef const __attribute__((address_space(1))) v4i* gptr_v4i;
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) v2u* gptr_v2u;
  const int64_t kb_stride = (int64_t)n_tt32 * 4608;
  gptr abase[TB];
#pragma unroll
  for (int jj = 0; jj < TB; ++jj) abase[jj] = (gptr)q8 + (int64_t)min((t0 >> 5) + jj, n_tt32 - 1) * 4608;
  // All activation loads are raw buffer loads: descriptor = the wave-uniform tile pointer (SGPRs, stepped with two scalar
  // adds per pair), vector offset = a per-lane constant, scalar offset = the uniform remainder.  No 64-bit vector
  // address arithmetic in the loop and no 64-bit per-lane pointers held in registers.
  auto arsrc = [&](int jj) { return __builtin_amdgcn_make_buffer_rsrc((void*)abase[jj], 0, (int)0xFFFFFFFFu, 0x00020000); };
  auto ld_b128 = [&](int jj, uint32_t voff, uint32_t soff) {
    const v4u t = __builtin_amdgcn_raw_buffer_load_b128(arsrc(jj), (int)voff, (int)soff, 0);
    return v4i{(int)t[0], (int)t[1], (int)t[2], (int)t[3]};
  };
  auto ld_b64 = [&](int jj, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b64(arsrc(jj), (int)voff, (int)soff, 0); };
  const uint32_t r8 = r * 8;

  v16f acc[TB];
#pragma unroll
  for (int jj = 0; jj < TB; ++jj)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[jj][i] = 0.0f;
  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = (int)MAGIC_I;

  v4i B[TB][2];
  uint32_t ds0[TB], ds1[TB];   // d8 (+ sum) words of groups 2p / 2p+1 (separate scalars: never indexed by a lane value)
  v2u dsn[TB];
  Win wq[C::WPI];              // windows of the next stage in flight
  const int p_begin = IPS * st_begin, p_end = IPS * st_end;
  if (p_begin < p_end) {
#pragma unroll
    for (int m = 0; m < C::NW; ++m) store_window(load_window(st_begin, m), st_begin, st_begin & 1, m);
    const int st1 = min(st_begin + 1, st_end - 1);
#pragma unroll
    for (int i = 0; i < C::WPI; ++i) wq[i] = load_window(st1, min(i, C::NW - 1));
#pragma unroll
    for (int jj = 0; jj < TB; ++jj) {
      abase[jj] += (int64_t)(p_begin >> 1) * kb_stride + (p_begin & 1) * 2048;
      B[jj][0] = ld_b128(jj, lane16, 0);
      B[jj][1] = ld_b128(jj, lane16 + 1024, 0);
      dsn[jj] = ld_b64(jj, r8, (p_begin & 1) ? 2304 : 4096);
    }
  }
#if GGQ_ABL & 32
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GGQ_STAMP(1);
#endif

  for (int p = p_begin; p < p_end; ++p) {
    asm volatile("" : "+v"(magic));   // keep the accumulator preset in registers: hipcc otherwise re-creates it from SGPRs every pair (-2 %)
    const int st = (int)((unsigned)p / IPS), q = (int)((unsigned)p % IPS);
    // abase[] points at pair p; step to the next pair (the last iteration re-reads its own pair)
    const bool more = p + 1 < p_end;
    const int64_t step = !more ? 0 : (p & 1) ? kb_stride - 2048 : 2048;
    const int ds_off = (((p & 1) != 0) == more) ? 4096 : 2304;   // next pair even -> +4096, odd -> +2304
    const uint8_t* stage = ring + (st & 1) * STAGE + r * C::PITCH;
    float* sb = sb0 + (p & 1) * 128;

    // ---- raw weight bytes of this lane from the ring (before the ring is written below) ----
    Raw R;
    v4i qs16 = {}, hdr = {}, qh16 = {};
    if constexpr (C::direct) {
      constexpr int QS = T == GGQ_TYPE_Q4_K ? off::Q4_K_QS : off::Q5_K_QS;
      qs16 = *(const v4i*)(stage + QS + 32 * q + 16 * h);
      if constexpr (!SBMIN) hdr = *(const v4i*)stage;
      if constexpr (T == GGQ_TYPE_Q5_K) qh16 = *(const v4i*)(stage + off::Q5_K_QH + 16 * h);
    } else if constexpr (C::split) {
      // de-interleaved stage: aligned 16-byte quant pieces + the block's header dword(s)
      const int blk = 2 * q + h;
      const v4i q0 = *(const v4i*)(stage + blk * C::QSB);
      R.q[0] = u32x4_a2{{(uint32_t)q0[0], (uint32_t)q0[1], (uint32_t)q0[2], (uint32_t)q0[3]}};
      if constexpr (C::QSB == 32) {
        const v4i q1 = *(const v4i*)(stage + blk * C::QSB + 16);
        R.q[1] = u32x4_a2{{(uint32_t)q1[0], (uint32_t)q1[1], (uint32_t)q1[2], (uint32_t)q1[3]}};
      }
      if constexpr (C::HB == 8) {
        const v2i hd = *(const v2i*)(stage + C::NB * C::QSB + 8 * blk);
        if constexpr (T == GGQ_TYPE_Q5_0) { R.s[0] = (uint32_t)hd[0] & 0xFFFF; R.s[1] = ((uint32_t)hd[0] >> 16) | ((uint32_t)hd[1] << 16); }
        else { R.s[0] = (uint32_t)hd[0]; R.s[1] = (uint32_t)hd[1]; }   // Q5_1: dm, qh
      } else {
        const uint32_t hd = *(const uint32_t*)(stage + C::NB * C::QSB + 4 * blk);
        R.s[0] = (T == GGQ_TYPE_Q4_1) ? hd : (hd & 0xFFFF);
      }
    } else {
      load_raw<T>(stage, 2 * q + h, R);
    }
    if constexpr (SBMIN) {
      if (q == 0) {   // wave-uniform: first pair of a super-block — row scales of its 8 groups to the wave's LDS line
        const v4i hd = *(const v4i*)stage;   // {d | dmin << 16, scales[0..3], scales[4..7], scales[8..11]}
        const uint32_t w0 = (uint32_t)hd[1], w2 = (uint32_t)hd[3];
        // get_scale_min_k4 for the four groups 4h .. 4h+3 of row r at once, one byte each
        const uint32_t hm = 0u - (uint32_t)h;   // all ones in the upper lane half: bit-select instead of a branch
        const uint32_t sc4 = (((w2 & 0x0F0F0F0Fu) | ((w0 >> 2) & 0x30303030u)) & hm) | (w0 & 0x3F3F3F3Fu & ~hm);
        const float dall = bits_h_f32((uint32_t)hd[0] & 0xFFFF);
#pragma unroll
        for (int j = 0; j < 4; ++j) sb0[(4 * h + j) * 32 + r] = dall * (float)((sc4 >> (8 * j)) & 0xFF);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // park the windows requested during the previous iteration (slot q of the stage after this one) in the
    // other ring buffer, then request the windows iteration p + 1 will park.  Past the wave's last stage the
    // source stage is clamped: valid bytes land in the dead buffer.
    {
      const int sw = min(st + 1, st_end - 1);
#pragma unroll
      for (int i = 0; i < C::WPI; ++i) store_window(wq[i], sw, (st + 1) & 1, min(q * C::WPI + i, C::NW - 1));
      const int sn = min((int)((unsigned)(p + 1) / IPS) + 1, st_end - 1), qn = (int)((unsigned)(p + 1) % IPS);
#pragma unroll
      for (int i = 0; i < C::WPI; ++i) wq[i] = load_window(sn, min(qn * C::WPI + i, C::NW - 1));
    }
#pragma unroll
    for (int jj = 0; jj < TB; ++jj) { ds0[jj] = dsn[jj][0]; ds1[jj] = dsn[jj][1]; }
    if constexpr (SBMIN) {
      // every pair leaves its two s8 values (high halves of the half2(d8, s8) words) in the wave's stash [jj][token][q];
      // the last pair of the super-block turns the eight of them into the min term
#pragma unroll
      for (int jj = 0; jj < TB; ++jj) s8l[(jj * 32 + r) * 4 + q] = __builtin_amdgcn_perm(ds1[jj], ds0[jj], 0x07060302u);
      __builtin_amdgcn_wave_barrier();
      if (q == IPS - 1) {   // wave-uniform
        const v4i hd = *(const v4i*)stage;
        const uint32_t w1 = (uint32_t)hd[2], w2 = (uint32_t)hd[3];
        const uint32_t hm = 0u - (uint32_t)h;
        const uint32_t m4 = ((((w2 >> 4) & 0x0F0F0F0Fu) | ((w1 >> 2) & 0x30303030u)) & hm) | (w1 & 0x3F3F3F3Fu & ~hm);
        const float dmin = bits_h_f32((uint32_t)hd[0] >> 16);
        // -dmin·m_g = hi + lo exactly, both fp16 (RTZ for hi: any rounding leaves a representable remainder).  Only
        // |dmin·m| beyond the fp16 range cannot be split: such a ROW (lane) contributes nothing to the MFMA below and goes
        // through the 2^-8-scaled pass of the cold branch instead, in which the other rows contribute nothing — so a row's
        // result does not depend on which rows share its tile.
        const bool lane_big = !(__builtin_fabsf(dmin) <= 1024.0f);
        const bool big = __builtin_amdgcn_ballot_w64(lane_big) != 0;
        auto split4 = [&](float dm) {   // k order of the lane half: hi(g0) hi(g1) lo(g0) lo(g1) hi(g2) hi(g3) lo(g2) lo(g3), g = group - 4h
          v4i av;
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const float p0 = -(dm * (float)((m4 >> (8 * j)) & 0xFF)), p1 = -(dm * (float)((m4 >> (8 * j + 8)) & 0xFF));
            const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            av[j] = (int)hb;
            av[j + 1] = (int)__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(p0 - bits_h_f32(hb & 0xFFFF), p1 - bits_h_f32(hb >> 16)));
          }
          return av;
        };
        const v4i av = split4(lane_big ? 0.0f : dmin);
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
          const v2u x = *(const v2u*)(s8l + (jj * 32 + r) * 4 + 2 * h);   // s8 of groups 4h .. 4h+3 of token r
          const v4i bv = {(int)x[0], (int)x[0], (int)x[1], (int)x[1]};   // s8(g0) s8(g1) twice, s8(g2) s8(g3) twice: matches av
          const h8 ah = __builtin_bit_cast(h8, av), bh = __builtin_bit_cast(h8, bv);
          acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[jj], 0, 0, 0);
          if (__builtin_expect(big, 0)) {   // cold: the rows with |dmin| > 1024 at 2^-8 of their scale, times 256 (exact) afterwards
            const v4i avs = split4(lane_big ? dmin * 0.00390625f : 0.0f);
            v16f z = {};
            z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, avs), bh, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[jj][i] = __builtin_fmaf(z[i], 256.0f, acc[jj][i]);
          }
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < TB; ++jj) {
      abase[jj] += step;
      dsn[jj] = ld_b64(jj, r8, ds_off);
    }

    // ---- A fragments of groups 2p, 2p+1 and their scales ----
    v4i a[2], a2[2];
    float s0, s1;
    if constexpr (C::direct) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[0][i] = (int)((uint32_t)qs16[i] & 0x0F0F0F0Fu);
        a[1][i] = (int)(((uint32_t)qs16[i] >> 4) & 0x0F0F0F0Fu);
      }
      if constexpr (T == GGQ_TYPE_Q5_K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[0][i] |= (int)((((uint32_t)qh16[i] >> (2 * q)) & 0x01010101u) << 4);
          a[1][i] |= (int)((((uint32_t)qh16[i] >> (2 * q + 1)) & 0x01010101u) << 4);
        }
      }
      int sc = 0, mn = 0;   // scale / min of group 2q + h (get_scale_min_k4): q < 2 <=> group < 4
      if constexpr (SBMIN) {
      } else if (q < 2) {
        sc = ((uint32_t)hdr[1] >> (16 * q + 8 * h)) & 63;
        mn = ((uint32_t)hdr[2] >> (16 * q + 8 * h)) & 63;
      } else {
        const int sh = 16 * (q - 2) + 8 * h;
        const uint32_t bb = ((uint32_t)hdr[3] >> sh) & 0xFF;
        sc = (bb & 0xF) | ((((uint32_t)hdr[1] >> (sh + 6)) & 3) << 4);
        mn = (bb >> 4) | ((((uint32_t)hdr[2] >> (sh + 6)) & 3) << 4);
      }
      if constexpr (SBMIN) { s0 = 0.0f; s1 = 0.0f; }
      else {
        s0 = bits_h_f32((uint32_t)hdr[0] & 0xFFFF) * (float)sc;
        s1 = -(bits_h_f32((uint32_t)hdr[0] >> 16) * (float)mn);
      }
    } else {
      uint32_t wv[8], wv2[8];
      unpack_raw<T>(R, 2 * q + h, wv, wv2, s0, s1);
      if (2 * p + h >= n_groups) {   // K tail of a legacy format: contributes nothing
        s0 = 0.0f; s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { wv[i] = 0; wv2[i] = 0; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // lanes 32-63 of wv[i] <-> lanes 0-31 of wv[4+i]: afterwards [0] = fragment of group 2p, [1] = of 2p+1
        const auto sw = __builtin_amdgcn_permlane32_swap(wv[i], wv[4 + i], false, false);
        a[0][i] = (int)sw[0]; a[1][i] = (int)sw[1];
        if constexpr (TR::two_tiles) {
          const auto sw2 = __builtin_amdgcn_permlane32_swap(wv2[i], wv2[4 + i], false, false);
          a2[0][i] = (int)sw2[0]; a2[1][i] = (int)sw2[1];
        }
      }
    }
    if constexpr (NR != 0) {
      // ---- transposed: per-lane weight scales of both groups, per-register token scales through LDS ----
      float sal[2], sbl[2] = {0.0f, 0.0f};
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, s0), __builtin_bit_cast(uint32_t, s0), false, false);
        sal[0] = as_f32((int)sw[0]); sal[1] = as_f32((int)sw[1]);   // scales of groups 2p, 2p+1 of row r, in both halves
        if constexpr (TR::n_scale == 2 && !TR::mfma_min) {
          const auto sw1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s1), false, false);
          sbl[0] = as_f32((int)sw1[0]); sbl[1] = as_f32((int)sw1[1]);
        }
      }
      // token lanes (h = 0) publish {bs, aux} of both groups: aux = s8 for the fp16-product formats, else unused
      float bsl[2], auxl[2];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const uint32_t dsw = gg ? ds1[0] : ds0[0];
        if constexpr (TR::need_sum) {
          bsl[gg] = bits_h_f32(dsw & 0xFFFF);
          auxl[gg] = TR::fp16_prod ? bits_h_f32(dsw >> 16) : 0.0f;
        } else {
          bsl[gg] = as_f32((int)dsw); auxl[gg] = 0.0f;
        }
      }
      if (h == 0) *(v4f*)(sb + 4 * r) = v4f{bsl[0], auxl[0], bsl[1], auxl[1]};
      __builtin_amdgcn_wave_barrier();

      if constexpr (TR::mfma_min) {
        // min term, operands swapped: rows = tokens (s8 of token lane & 31, group 2p+h), columns = weight rows
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bits_h_f32((h ? ds1[0] : ds0[0]) >> 16), s1, acc[0], 0, 0, 0);
      }
      v4f tk[NR];   // {bs0, aux0, bs1, aux1} of the token of accumulator register i: token 8 (i >> 2) + 4 h + (i & 3)
#pragma unroll
      for (int i = 0; i < NR; ++i) tk[i] = *(const v4f*)(sb + 4 * (8 * (i >> 2) + 4 * h + (i & 3)));
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        v4i alo = a[gg], ahi = a[gg];
        if constexpr (TR::half_scales) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { alo[i] = h == 0 ? a[gg][i] : 0; ahi[i] = h == 1 ? a[gg][i] : 0; }
        }
        // (few live registers: a zero accumulator input + v_cvt_f32_i32 on the live registers is cheaper than
        //  keeping the 16-register 0x4B400000 constant alive; (float)C·d8 rounds exactly like the preset trick)
        const v16i zero = {};
        v16i c0, c1 = zero;
        if constexpr (TR::half_scales) {
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[0][gg], alo, zero, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[0][gg], ahi, zero, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[0][gg], a[gg], zero, 0, 0, 0);
          if constexpr (TR::two_tiles) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[0][gg], a2[gg], zero, 0, 0, 0);
        }
        B[0][gg] = ld_b128(0, lane16 + 1024 * gg, 0);
        const float sae = sal[gg], sbe = sbl[gg];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const float cf0 = (float)c0[i];
          const float bs = gg ? tk[i][2] : tk[i][0], aux = gg ? tk[i][3] : tk[i][1];
          if constexpr (TR::fp16_prod) {
            const float lo = (float)((_Float16)sae * (_Float16)bs);
            const float hi = (float)((_Float16)sbe * (_Float16)aux);
            acc[0][i] += __builtin_fmaf(lo, cf0, hi);
          } else if constexpr (TR::two_tiles) {
            const float cf1 = (float)c1[i];
            acc[0][i] = __builtin_fmaf(bs, __builtin_fmaf(sae, cf0, -(sbe * cf1)), acc[0][i]);
          } else if constexpr (TR::half_scales) {
            const float cf1 = (float)c1[i];
            acc[0][i] = __builtin_fmaf(cf0 * bs, sae, acc[0][i]);
            acc[0][i] = __builtin_fmaf(cf1 * bs, sbe, acc[0][i]);
          } else {
            acc[0][i] = __builtin_fmaf(cf0 * bs, sae, acc[0][i]);
          }
        }
      }
    } else if constexpr (XT) {
      float sal[2], sbl[2] = {0.0f, 0.0f};
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, s0), __builtin_bit_cast(uint32_t, s0), false, false);
        sal[0] = as_f32((int)sw[0]); sal[1] = as_f32((int)sw[1]);   // scales of groups 2p, 2p+1 of row r, in both halves
        if constexpr (TR::half_scales) {
          const auto sw1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s1), false, false);
          sbl[0] = as_f32((int)sw1[0]); sbl[1] = as_f32((int)sw1[1]);
        }
      }
      // the fp32 d8 of token r: lanes h = 0 publish group 2p, lanes h = 1 group 2p+1 -> line [token block][group][token]
#pragma unroll
      for (int jj = 0; jj < TB; ++jj) sb[jj * 64 + h * 32 + r] = as_f32((int)(h ? ds1[jj] : ds0[jj]));
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const float sae = sal[gg], sbe = sbl[gg];
        const float nma = -(MAGIC_F * sae), nmb = -(MAGIC_F * sbe);   // exact: the row scales have <= 19 significant bits
        v4i alo = a[gg], ahi = a[gg];
        if constexpr (TR::half_scales) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { alo[i] = h == 0 ? a[gg][i] : 0; ahi[i] = h == 1 ? a[gg][i] : 0; }
        }
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
          v4f dk[4];   // d8 of the tokens of registers 4 qd .. 4 qd + 3: token 8 qd + 4 h + e
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) dk[qd] = *(const v4f*)(sb + jj * 64 + gg * 32 + 8 * qd + 4 * h);
          v16i c0, c1 = magic;
          if constexpr (TR::half_scales) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[jj][gg], alo, magic, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[jj][gg], ahi, magic, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B[jj][gg], a[gg], magic, 0, 0, 0);
          }
          B[jj][gg] = ld_b128(jj, lane16 + 1024 * gg, 0);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float d8 = dk[i >> 2][i & 3];
            acc[jj][i] = __builtin_fmaf(__builtin_fmaf(as_f32(c0[i]), sae, nma), d8, acc[jj][i]);
            if constexpr (TR::half_scales) acc[jj][i] = __builtin_fmaf(__builtin_fmaf(as_f32(c1[i]), sbe, nmb), d8, acc[jj][i]);
          }
        }
      }
    } else {
      if constexpr (SBMIN) sb = sb0 + q * 64;   // the super-block's line [group][row]: groups 2q, 2q+1
      else if constexpr (TR::fp16_prod) {
        // Q4_1 / Q5_1: the line carries the block's half2(d, m) word itself (s0, s1 came from halves: the pack is exact),
        // so that the per-triple __hmul2(dm, ds8) of the reference is ONE v_pk_mul_f16 against the token's half2(d8, s8)
        sb[h * 32 + r] = as_f32((int)__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(s0, s1)));
        __builtin_amdgcn_wave_barrier();
      } else {
      sb[h * 32 + r] = s0;
      if constexpr (TR::n_scale == 2 && !TR::mfma_min) sb[64 + h * 32 + r] = s1;
      __builtin_amdgcn_wave_barrier();
      }

      if constexpr (TR::mfma_min && !SBMIN) {
        // min term: Σ (-dmin·m)[row, 2p+h] · s8[token, 2p+h] on the matrix pipe
  #pragma unroll
        for (int jj = 0; jj < TB; ++jj)
          acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, bits_h_f32((h ? ds1[jj] : ds0[jj]) >> 16), acc[jj], 0, 0, 0);
      }

  #pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        v4f sa[4], sbv[4];
  #pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          sa[qd] = *(const v4f*)(sb + gg * 32 + 8 * qd + 4 * h);
          if constexpr (TR::n_scale == 2 && !TR::mfma_min && !TR::fp16_prod) sbv[qd] = *(const v4f*)(sb + 64 + gg * 32 + 8 * qd + 4 * h);
        }
        v4i alo = a[gg], ahi = a[gg];
        if constexpr (TR::half_scales) {   // Q6_K: separate sums over k < 16 and k >= 16 — zero the other half's lanes
  #pragma unroll
          for (int i = 0; i < 4; ++i) { alo[i] = h == 0 ? a[gg][i] : 0; ahi[i] = h == 1 ? a[gg][i] : 0; }
        }
  #pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
          const uint32_t dsw = gg ? ds1[jj] : ds0[jj];
          float bs, bm = 0.0f;
          if constexpr (TR::need_sum) { bs = bits_h_f32(dsw & 0xFFFF); bm = bits_h_f32(dsw >> 16); }
          else bs = as_f32((int)dsw);
          const float nmbs = -(MAGIC_F * bs);   // exact when bs is an fp16 value (need_sum formats)
          v16i c0, c1 = magic;
          if constexpr (TR::half_scales) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, B[jj][gg], magic, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, B[jj][gg], magic, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[gg], B[jj][gg], magic, 0, 0, 0);
            if constexpr (TR::two_tiles) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[gg], B[jj][gg], magic, 0, 0, 0);
          }
          // refill in place (hipcc sinks these loads to the end of the loop body; pinning them with sched_barrier
          // spills at 168 VGPRs, and without them the kernel is only 1.2 us faster: not the limiter)
          B[jj][gg] = ld_b128(jj, lane16 + 1024 * gg, 0);
  #pragma unroll
          for (int qd = 0; qd < 4; ++qd)
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * qd + e;
              const float df0 = as_f32(c0[i]);   // = 12582912 + C exactly
              const float sae = sa[qd][e];
              // explicit fma + -ffp-contract=off: every accumulator register sees the same instruction
              // sequence, so a row's result does not depend on its position in the tile
              if constexpr (TR::fp16_prod) {   // Q4_1/Q5_1, mmq.cuh:527-529 / :840-842
                // half2(d d8, m s8) in one packed fp16 multiply, then fma(lo, C, hi) straight from the halves (v_fma_mix_f32)
                // (asm: hipcc otherwise converts both halves to fp32 first — two more vector ops per triple)
                const h2 pr = __builtin_bit_cast(h2, sae) * __builtin_bit_cast(h2, dsw);
                float fr;
                asm("v_fma_mix_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(fr) : "v"(__builtin_bit_cast(uint32_t, pr)), "v"(df0 - MAGIC_F));
                acc[jj][i] += fr;
              } else if constexpr (TR::two_tiles) {   // Q2_K: d8 (dall·Σsc q q8 − dmin·Σ m q8), mmq.cuh:47
                const float df1 = as_f32(c1[i]);
                acc[jj][i] = __builtin_fmaf(bs, __builtin_fmaf(sae, df0 - MAGIC_F, -(sbv[qd][e] * (df1 - MAGIC_F))), acc[jj][i]);
              } else if constexpr (TR::half_scales) {   // Q6_K (fp32 d8): mmq.cuh:1726-1732
                const float df1 = as_f32(c1[i]);
                acc[jj][i] = __builtin_fmaf((df0 - MAGIC_F) * bs, sae, acc[jj][i]);
                acc[jj][i] = __builtin_fmaf((df1 - MAGIC_F) * bs, sbv[qd][e], acc[jj][i]);
              } else if constexpr (TR::need_sum) {   // Q4_0, Q4_K, Q5_K (fp16 d8): float(C)·d8 in one exact fma
                acc[jj][i] = __builtin_fmaf(__builtin_fmaf(df0, bs, nmbs), sae, acc[jj][i]);
              } else {   // Q5_0 / Q8_0 / Q3_K (fp32 d8): d_w d8 C
                acc[jj][i] = __builtin_fmaf((df0 - MAGIC_F) * bs, sae, acc[jj][i]);
              }
            }
        }
      }
    }
    // (no barrier here: the scale line is double-buffered by pair parity, so the next iteration may start early)
  }

  GGQ_STAMP(2);
  // ---- K-slice reduction (the rings are dead once every wave has passed its last ds_read) ----
  __syncthreads();
  float* red = (float*)lds;   // [K-slice][TB][16][64]
  constexpr int NLIVE = NR ? NR : 16;
  // lane = token, register 4 qd + e = row 8 qd + 4 h + e: one 8-byte store of 4 consecutive rows per (token block, qd)
  const bool vec_ok = DT != GGQ_F32 && (ldy & 3) == 0 && ((uintptr_t)y & 7) == 0 && n0 + 32 <= n_rows;
  auto store_quarter = [&](int jj, int qd, float (&v)[4]) {
    const int t = t0 + 32 * jj + r;
    if (t >= batch) return;
    const int row = n0 + 8 * qd + 4 * h;
    if (epi != GGQ_EPI_NONE) {   // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row + e < n_rows) v[e] = apply_epilogue<DT>(v[e], epi, aux, (int64_t)t * ldy + row + e, row + e);
    }
    if (vec_ok) {
      uint16_t hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (DT == GGQ_F16) hv[e] = __builtin_bit_cast(uint16_t, (_Float16)v[e]);
        else hv[e] = Elem<GGQ_BF16>::cvt(v[e]);
      }
      uint2 pk;
      pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
      pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
      *(uint2*)((uint16_t*)y + (int64_t)t * ldy + row) = pk;
      for (int d = 1; d < go.n_dst; ++d) {   // (kernel-uniform) the peers' slots: system-coherent write-through stores
        uint16_t* p = (uint16_t*)go.dst[d] + (int64_t)t * ldy + row;
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(pk) : "memory");
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row + e < n_rows) {
          Elem<DT>::st(y, (int64_t)t * ldy + row + e, v[e]);
          gather_store<DT>(go, (int64_t)t * ldy + row + e, v[e]);
        }
    }
  };
  // publish (multi-destination launches): a wave that has stored drains its own stores (write-through: complete at vmcnt 0) and
  // arrives; the last of the launch's `n_units * waves_that_store` arrivals writes the flags — the one release at system scope
  auto gather_arrive = [&](int storing_waves) {
    if (go.n_flag <= 0) return;   // (kernel-uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const uint32_t before = __hip_atomic_fetch_add(go.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (uint32_t)n_units * (uint32_t)storing_waves - 1u) {
        __hip_atomic_store(go.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = 0; d < go.n_flag; ++d) __hip_atomic_store(go.flag[d], go.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  };

  if constexpr (NR == 0 && !XT && KS == 4 && T != GGQ_TYPE_Q5_0) {   // (Q5_0 sits at 168 VGPRs: the four specialised finishes push it into scratch, 34 -> 39 us)
    // All four waves finish the unit: wave q owns register quarter q (rows 8 q + 4 h .. + 3 of every token), publishes
    // the other three quarters of its partial sums and adds up its own quarter in the fixed order
    // ((slice 0 + slice 1) + slice 2) + slice 3 — the order the single-wave form below uses, so a result does not depend
    // on which wave produced it.  A quarter of the LDS reads, adds and stores per wave instead of all of them on wave 0.
    auto publish = [&](auto QC) {
      constexpr int Q = decltype(QC)::value;
#pragma unroll
      for (int jj = 0; jj < TB; ++jj)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if ((i >> 2) != Q) red[((Q * TB + jj) * 16 + i) * 64 + lane] = acc[jj][i];
    };
    auto finish = [&](auto QC) {
      constexpr int Q = decltype(QC)::value;
#pragma unroll
      for (int jj = 0; jj < TB; ++jj) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * Q + e;
          float sum = Q == 0 ? acc[jj][i] : red[((0 * TB + jj) * 16 + i) * 64 + lane];
#pragma unroll
          for (int sl = 1; sl < 4; ++sl) sum += sl == Q ? acc[jj][i] : red[((sl * TB + jj) * 16 + i) * 64 + lane];
          v[e] = sum;
        }
        store_quarter(jj, Q, v);
      }
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    switch (ks) { case 0: publish(Q0{}); break; case 1: publish(Q1{}); break; case 2: publish(Q2{}); break; default: publish(Q3{}); }
    __syncthreads();
    GGQ_STAMP(3);
    switch (ks) { case 0: finish(Q0{}); break; case 1: finish(Q1{}); break; case 2: finish(Q2{}); break; default: finish(Q3{}); }
    GGQ_STAMP(4);
    gather_arrive(4);
    return;
  }

  if (ks > 0) {
#pragma unroll
    for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int i = 0; i < NLIVE; ++i) red[(((ks - 1) * TB + jj) * 16 + i) * 64 + lane] = acc[jj][i];
  }
  __syncthreads();
  GGQ_STAMP(3);
  if (ks != 0) return;
#pragma unroll
  for (int s = 0; s < KS - 1; ++s)
#pragma unroll
    for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int i = 0; i < NLIVE; ++i) acc[jj][i] += red[((s * TB + jj) * 16 + i) * 64 + lane];

  if constexpr (NR != 0 || XT) {
    // register i = token t0 + 32 jj + 8 (i >> 2) + 4 h + (i & 3), lane & 31 = weight row: 32 consecutive rows per store
    if (n0 + r < n_rows) {
#pragma unroll
      for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int i = 0; i < NLIVE; ++i) {
        const int t = t0 + 32 * jj + 8 * (i >> 2) + 4 * h + (i & 3);
        if (t < batch) {
          const float o = apply_epilogue<DT>(acc[jj][i], epi, aux, (int64_t)t * ldy + n0 + r, n0 + r);
          Elem<DT>::st(y, (int64_t)t * ldy + n0 + r, o);
          gather_store<DT>(go, (int64_t)t * ldy + n0 + r, o);
        }
      }
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[4] = {acc[jj][4 * qd], acc[jj][4 * qd + 1], acc[jj][4 * qd + 2], acc[jj][4 * qd + 3]};
        store_quarter(jj, qd, v);
      }
  }
  GGQ_STAMP(4);
  gather_arrive(1);
}

template <int T, int DT, int TB, int KS, int NR>
static int launch_mmq_stream_ks(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                                int64_t ldy, int64_t n_tok_tiles, int64_t n_units, hipStream_t s, Epilogue ep) {
  constexpr int LDS = StreamLaunch<T, TB, KS>::LDS;
  auto kern = mmq_stream_kernel<T, DT, TB, KS, NR>;
  if (LDS > 64 * 1024) {
    // per call: the attribute is per device (a process may drive several GPUs), and it is a cheap host call
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return GGQ_ERR_LAUNCH;
  }
  const int64_t per_xcd = (n_units + 7) / 8;
  GGQ_HIP_PRE_LAUNCH();
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(64 * KS), LDS, s,
                     (const uint8_t*)w, (const uint8_t*)q8, y, (int)k, (int)n, (int)batch, ldy,
                     (int)n_tok_tiles, (int)n_units, (int)per_xcd, ep.kind, ep.aux, ep.go);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T, int DT, int TB>
static int launch_mmq_stream(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n,
                             int64_t ldy, hipStream_t s, Epilogue ep) {
  const int64_t n_tok_tiles = (batch + 32 * TB - 1) / (32 * TB);
  const int64_t n_units = ((n + 31) / 32) * n_tok_tiles;
  if (n_units > 0x7fffffffLL - 8) return GGQ_ERR_SHAPE;
  // Too few units to give every SIMD three waves: eight K-slices per unit instead of four.
  //   64-token units (235 VGPRs, one eight-wave workgroup per CU): only with at most one unit per CU (down-projection
  //   shapes, 4096 rows x 128 tokens = 256 units: 40.1 -> 32.5 us; with 344 units it would need a second round);
  //   32-token units (<= 128 VGPRs, two workgroups per CU): up to 512 units — the batch <= 32 case of the 11008-row
  //   shape is otherwise a latency chain of 16 pair-iterations per wave at 1.3 waves per SIMD.
  static const char* e = GGQ_TUNING_ENV("GGQ_MMQ_KS");
  const int64_t n_st = (k + StreamCfg<T>::SE - 1) / StreamCfg<T>::SE;
  const bool two_per_cu = StreamLaunch<T, TB, 8>::WG_PER_CU == 2;
  const bool ks8 = e ? e[0] == '8' : (n_units <= (two_per_cu ? 512 : 256) && n_st >= 16);
  if constexpr (TB == 1) {   // batch <= 16: transposed variant, only the registers that hold tokens are scaled
    static const char* et = GGQ_TUNING_ENV("GGQ_MMQ_TRANS");
    const bool trans = et ? et[0] == '1' : true;
    if (trans && batch <= 8)
      return ks8 ? launch_mmq_stream_ks<T, DT, 1, 8, 4>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep)
                 : launch_mmq_stream_ks<T, DT, 1, 4, 4>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep);
    if (trans && batch <= 16)
      return ks8 ? launch_mmq_stream_ks<T, DT, 1, 8, 8>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep)
                 : launch_mmq_stream_ks<T, DT, 1, 4, 8>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep);
  }
  if (ks8) return launch_mmq_stream_ks<T, DT, TB, 8, 0>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep);
  return launch_mmq_stream_ks<T, DT, TB, 4, 0>(w, q8, y, batch, k, n, ldy, n_tok_tiles, n_units, s, ep);
}
}  // namespace ggq

namespace ggq {
template <int T>
static int launch_mmq_tiled(const void* w, const void* q8, void* y, int dt, int64_t batch, int64_t k,
                            int64_t n, int64_t ldy, hipStream_t s, Epilogue ep) {
  // 32-token units while one token tile covers the batch (and up to batch 64 for four formats), 64-token units beyond
  // (measured r1, Q4_K 11008x4096: batch 32 14.4 vs 20.0 us, batch 128 38.8 vs 29.9 us; r3: ggq_mmq_stream_unit_tokens)
  static const char* e = GGQ_TUNING_ENV("GGQ_MMQ_TB");   // experiments: force 32- or 64-token units
  // (Q2_K's second int8 tile does not fit 168 VGPRs with two token blocks: 168 us spilled vs 54 us)
  const bool one = e ? e[0] == '1' : ggq_mmq_stream_unit_tokens(T, batch, n) == 32;   // (csrc/core/traits.cpp: per format, measured)
  switch (dt) {
    case GGQ_F32: return one ? launch_mmq_stream<T, GGQ_F32, 1>(w, q8, y, batch, k, n, ldy, s, ep) : launch_mmq_stream<T, GGQ_F32, 2>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_F16: return one ? launch_mmq_stream<T, GGQ_F16, 1>(w, q8, y, batch, k, n, ldy, s, ep) : launch_mmq_stream<T, GGQ_F16, 2>(w, q8, y, batch, k, n, ldy, s, ep);
    case GGQ_BF16: return one ? launch_mmq_stream<T, GGQ_BF16, 1>(w, q8, y, batch, k, n, ldy, s, ep) : launch_mmq_stream<T, GGQ_BF16, 2>(w, q8, y, batch, k, n, ldy, s, ep);
    default: return GGQ_ERR_DTYPE;
  }
}
}  // namespace ggq

namespace ggq {
int mul_mat_q_stream_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                          int64_t ldy, int epilogue, const void* aux, void* stream, const void* go);
}

extern "C" int ggq_mmq_tiled_supported(int type, int64_t k) {
  // the streamed kernel addresses a 32-row weight tile with 32-bit byte offsets: rows up to 32 MiB (K of a few
  // tens of millions); longer rows stay on the reference-layout kernel
  return ggq_mmq_type_supported(type) && k > 0 && k % ggq_block_elems(type) == 0 && ggq_row_bytes(type, k) <= (32 << 20);
}

extern "C" int ggq_mul_mat_q_pretiled(const void* w, const void* q, void* y, int type, int dtype,
                                      int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                                      void* stream) {
  return ggq_mul_mat_q_pretiled_epi(w, q, y, type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream);
}

extern "C" int ggq_mul_mat_q_pretiled_epi(const void* w, const void* q, void* y, int type, int dtype,
                                          int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                                          int epilogue, const void* aux, void* stream) {
  return ggq::mul_mat_q_stream_impl(w, q, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream, nullptr);
}

// the streamed kernel behind ggq_mul_mat_q_pretiled_epi; go != nullptr: with the multi-destination write-back (ggq_mul_mat_q_gather)
int ggq::mul_mat_q_stream_impl(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                               int64_t ldy, int epilogue, const void* aux, void* stream, const void* go) {
  using namespace ggq;
  if (epilogue < GGQ_EPI_NONE || epilogue > GGQ_EPI_SILU_MUL || (epilogue != GGQ_EPI_NONE && !aux)) return GGQ_ERR_ARG;
  const Epilogue ep{epilogue, aux, go ? *(const GatherOut*)go : GatherOut{}};
  if (k <= 0 || n_rows < 0 || batch < 0 || ldy < n_rows) return GGQ_ERR_ARG;
  if (!ggq_mmq_type_supported(type)) return GGQ_ERR_TYPE;
  if (k % ggq_block_elems(type)) return GGQ_ERR_SHAPE;
  if (!ggq_mmq_tiled_supported(type, k)) return GGQ_ERR_TYPE;
  if (k > (1 << 30) || n_rows > 0x7fffffffLL - 64 || batch > 0x7fffffffLL / 256) return GGQ_ERR_SHAPE;
  if (dtype < GGQ_F32 || dtype > GGQ_BF16) return GGQ_ERR_DTYPE;
  if (n_rows == 0 || batch == 0) return GGQ_OK;
  if (!w || !q || !y) return GGQ_ERR_ARG;
  if (((uintptr_t)w & 1) || ((uintptr_t)q & 15)) return GGQ_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
#ifdef GGQ_DEV_ONLY
  if (type != GGQ_DEV_ONLY || dtype != GGQ_F16 || batch <= 32) return GGQ_ERR_TYPE;
#ifndef GGQ_DEV_TB
#define GGQ_DEV_TB 2
#endif
  return launch_mmq_stream_ks<GGQ_DEV_ONLY, GGQ_F16, GGQ_DEV_TB, 4, 0>(w, q, y, batch, k, n_rows, ldy, (batch + 32 * GGQ_DEV_TB - 1) / (32 * GGQ_DEV_TB),
                                                                      ((n_rows + 31) / 32) * ((batch + 32 * GGQ_DEV_TB - 1) / (32 * GGQ_DEV_TB)), s, ep);
#else
  switch (type) {
    case GGQ_TYPE_Q4_0: return launch_mmq_tiled<GGQ_TYPE_Q4_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q4_1: return launch_mmq_tiled<GGQ_TYPE_Q4_1>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_0: return launch_mmq_tiled<GGQ_TYPE_Q5_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_1: return launch_mmq_tiled<GGQ_TYPE_Q5_1>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q8_0: return launch_mmq_tiled<GGQ_TYPE_Q8_0>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q2_K: return launch_mmq_tiled<GGQ_TYPE_Q2_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q3_K: return launch_mmq_tiled<GGQ_TYPE_Q3_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q4_K: return launch_mmq_tiled<GGQ_TYPE_Q4_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q5_K: return launch_mmq_tiled<GGQ_TYPE_Q5_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    case GGQ_TYPE_Q6_K: return launch_mmq_tiled<GGQ_TYPE_Q6_K>(w, q, y, dtype, batch, k, n_rows, ldy, s, ep);
    default: return GGQ_ERR_TYPE;
  }
#endif
}

extern "C" int ggq_mul_mat_q_ld(const void* w, const void* x, void* y, int type, int dtype,
                                int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                                void* scratch, void* stream) {
  if (!scratch) return GGQ_ERR_ARG;
  // kernel selection: ggq_mmq_route (csrc/core/traits.cpp) — format, batch and shape; the measurements behind every
  // threshold are recorded there
  switch (ggq_mmq_route(type, batch, k, n_rows)) {
    case GGQ_MMQ_ROUTE_T16: {
      const int rc = ggq_quantize_q8_1_t16(x, dtype, scratch, batch, k, type, stream);
      if (rc != GGQ_OK) return rc;
      return ggq_mul_mat_q_t16(w, scratch, y, type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream);
    }
    case GGQ_MMQ_ROUTE_X64: {
      const int rc = ggq_quantize_q8_1_x64(x, dtype, scratch, batch, k, type, stream);
      if (rc != GGQ_OK) return rc;
      return ggq_mul_mat_q_x64(w, scratch, y, type, dtype, batch, k, n_rows, ldy, GGQ_EPI_NONE, nullptr, stream);
    }
    case GGQ_MMQ_ROUTE_STREAM: {
      const int rc = ggq_quantize_q8_1_tiled(x, dtype, scratch, batch, k, type, stream);
      if (rc != GGQ_OK) return rc;
      return ggq_mul_mat_q_pretiled(w, scratch, y, type, dtype, batch, k, n_rows, ldy, stream);
    }
    default: break;   // dot4 / LDS-tile kernels on the reference layout; argument errors are reported by the callee
  }
  int rc = ggq_quantize_q8_1_mmq(x, dtype, scratch, batch, k, type, stream);
  if (rc != GGQ_OK) return rc;
  return ggq_mul_mat_q_prequant(w, scratch, y, type, dtype, batch, k, n_rows, ldy, stream);
}

extern "C" int ggq_mul_mat_q_epi(const void* w, const void* x, void* y, int type, int dtype, int64_t batch, int64_t k,
                                 int64_t n_rows, int64_t ldy, int epilogue, const void* aux, void* scratch,
                                 void* stream) {
  if (!scratch) return GGQ_ERR_ARG;
  const int route = ggq_mmq_route(type, batch, k, n_rows);
  if (route == GGQ_MMQ_ROUTE_T16) {
    const int rc = ggq_quantize_q8_1_t16(x, dtype, scratch, batch, k, type, stream);
    if (rc != GGQ_OK) return rc;
    return ggq_mul_mat_q_t16(w, scratch, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream);
  }
  if (route == GGQ_MMQ_ROUTE_X64) {
    const int rc = ggq_quantize_q8_1_x64(x, dtype, scratch, batch, k, type, stream);
    if (rc != GGQ_OK) return rc;
    return ggq_mul_mat_q_x64(w, scratch, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream);
  }
  if (!ggq_mmq_tiled_supported(type, k)) return ggq_mmq_type_supported(type) ? GGQ_ERR_SHAPE : GGQ_ERR_TYPE;
  const int rc = ggq_quantize_q8_1_tiled(x, dtype, scratch, batch, k, type, stream);
  if (rc != GGQ_OK) return rc;
  return ggq_mul_mat_q_pretiled_epi(w, scratch, y, type, dtype, batch, k, n_rows, ldy, epilogue, aux, stream);
}

extern "C" int ggq_mul_mat_q(const void* w, const void* x, void* y, int type, int dtype,
                             int64_t batch, int64_t k, int64_t n_rows, void* scratch, void* stream) {
  return ggq_mul_mat_q_ld(w, x, y, type, dtype, batch, k, n_rows, n_rows, scratch, stream);
}
