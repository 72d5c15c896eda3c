// mmq_sk.hip — persistent stream-K quantised GEMM  Y[B,N] = X[B,K] (Q8_1) · W[N,K]^T, int8 MFMA, gfx950.
//
// Same role and arithmetic as the streamed kernel of mmq.hip (mul_mat_q, HK/ggml/mmq.cuh:1917-1986, with the
// tensor-core bodies :1274-1553 as the numerical canon: exact int8 contraction per 32-group, fp16-valued d8 / s8,
// min term -dmin·m·s8), restructured around what round 1 measured (profiles/r01_mmq_pmc.md):
//   * the loop is bound by vector-instruction issue, and 688 units on 256 CUs left a 2.69-units-per-CU tail.
//     Here the (unit, K) space is cut into equal contiguous ranges, one per RESIDENT workgroup (stream-K): a
//     workgroup finishes the tail of one unit and starts the head of the next; a head is published as an fp32
//     partial tile (slot + flag in the scratch) and added by the workgroup that finishes the unit.  Heads are
//     computed FIRST, so no workgroup ever waits before it has published: no circular wait.
//   * per (row, token, 32-group) triple exactly two FMAs, issued as v_pk_fma_f32 (measured 1.0-1.2 ns per
//     FMA-equivalent per SIMD at >= 2 waves, scripts/ubench_occ.hip); everything else is amortised:
//       - token scales arrive as fp32 {d8 g0, d8 g1, s8} from the quantiser (LAYOUT 3): no conversions here;
//       - the 6-bit scales / mins of a super-block are decoded ONCE per super-block per row, four groups per
//         lane with packed byte arithmetic (lane (r, h) owns groups h, 2+h, 4+h, 6+h of row r), not per pair;
//       - the K loop is unrolled over the four pairs of a super-block: every shift and LDS offset is static;
//   * the int8 MFMA of tile-group t+1 is issued before the FMAs of tile-group t (two result buffers), the
//     fp32 min-term MFMA of a token block right after its first apply: the matrix pipe runs under the FMAs.
// Workgroup = 4 waves = 4 K-slices of its current range; wave-private LDS ring (weights one stage = one
// super-block ahead) exactly as in mmq.hip; no workgroup barrier inside the K loop.
#include "mmq_unpack.h"
#include "../core/sk_policy.h"

#ifdef GGQ_SK_STAMPS   // per-wave timeline (scripts/stamps_sk.py); never defined in a shipped build
__device__ unsigned long long g_sk_stamps[1024 * 4 * 16];
extern "C" int ggq_debug_read_sk_stamps(void* dst, long long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sk_stamps), n * 8);
}
#define SK_STAMP(i)                                                                              \
  do {                                                                                           \
    if (lane == 0 && wg < 1024 && (i) < 16) g_sk_stamps[(wg * 4 + ks) * 16 + (i)] = wall_clock64(); \
  } while (0)
#else
#define SK_STAMP(i) do {} while (0)
#endif

#ifndef SK_CBUF
#define SK_CBUF 1
#endif
#ifdef SK_DBG_SC128   // debug: fetch the 12-byte scale triple with a 16-byte load
#define SK_LD_SC(r, v, s) __builtin_shufflevector(__builtin_amdgcn_raw_buffer_load_b128(r, v, s, 0), __builtin_amdgcn_raw_buffer_load_b128(r, v, s, 0), 0, 1, 2)
#else
#define SK_LD_SC(r, v, s) __builtin_amdgcn_raw_buffer_load_b96(r, v, s, 0)
#endif
#ifndef SK_WPC2
#define SK_WPC2 3
#endif

namespace ggq {

template <int T> struct SkFmt {
  static_assert(T == GGQ_TYPE_Q4_K || T == GGQ_TYPE_Q5_K, "stream-K kernel: Q4_K / Q5_K");
  static constexpr int BS = Fmt<T>::BS;
  static constexpr int CPR = (BS + 15) / 16;         // 16-byte chunks of one row's super-block
  static_assert(BS % 16 == 0, "aligned chunks");
  static constexpr int PITCH = BS;
  static constexpr int STAGE = 32 * PITCH;           // one super-block of each of the 32 rows
  static constexpr int QS = T == GGQ_TYPE_Q4_K ? off::Q4_K_QS : off::Q5_K_QS;
  static constexpr int SB_BYTES = 2 * 4 * 2 * 32 * 4;   // row scales + mins of the current stage: [sa | sm][pair][group][row] fp32
  static constexpr int WAVE = 2 * STAGE + SB_BYTES;
};

template <int T, int TB> struct SkLaunch {
  static constexpr int RED = 4 * TB * 16 * 64 * 4;    // K-slice reduction (aliases the rings)
  static constexpr int RINGS = 4 * SkFmt<T>::WAVE;
  static constexpr int LDS = RINGS > RED ? RINGS : RED;
  static constexpr int WPC = TB >= 4 ? 2 : TB == 2 ? SK_WPC2 : 3;   // resident workgroups per CU (= waves per SIMD)
  static_assert(LDS * WPC <= 160 * 1024, "LDS of the resident workgroups");
};

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));

// raw buffer descriptor over [p, p + bytes): loads beyond it return 0 instead of faulting, and the scalar
// offset operand keeps all per-pair address arithmetic out of the vector ALU
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int T, int DT, int TB>
__global__ void __launch_bounds__(256, (SkLaunch<T, TB>::WPC)) mmq_sk_kernel(
    const uint8_t* __restrict__ w, const uint8_t* __restrict__ q8, void* __restrict__ y, int k, int n_rows, int batch,
    int64_t ldy, int n_tok_tiles, int n_units, int G, uint32_t q8_bytes, float* __restrict__ partials,
    int* __restrict__ flags) {
  using F = SkFmt<T>;
  constexpr int BS = F::BS, CPR = F::CPR, PITCH = F::PITCH, STAGE = F::STAGE;
  constexpr int NT = 2 * TB;   // tile-groups (32 rows x 32 tokens x 32 K) per pair
  constexpr int NQ = TB * 4;   // accumulator registers each wave finalises after the K-slice reduction
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int P = k >> 6, n_st = k >> 8;               // pairs / stages (super-blocks) per row
  const uint32_t row_bytes = (uint32_t)n_st * BS;
  const int n_tt32 = (batch + 31) >> 5;
  uint8_t* ring = lds + ks * F::WAVE;
  float* sbuf = (float*)(ring + 2 * STAGE);          // [sa | sm][pair q][group gg][row]

  // ---- this workgroup's range of the linear (unit, pair) space; boundaries are multiples of 4 pairs ----
  const int wg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // the workgroups of one XCD are consecutive
  const int64_t total4 = (int64_t)n_units * (P >> 2);
  auto bound = [&](int i) { return (int)((int64_t)i * total4 / G) * 4; };
  const int lo = bound(wg), hi = bound(wg + 1);

  const int lrow = min(lane / CPR, 3), lchunk = lane % CPR;   // ring copy: 4 rows per 64-lane window
  const uint32_t lane_src = (uint32_t)lrow * row_bytes + 16 * lchunk;
  uint8_t* const ring_dst = ring + lrow * PITCH + 16 * lchunk;
  const uint8_t* const ring_qs = ring + r * PITCH + F::QS + 16 * h;
  const uint32_t lane16 = lane * 16, lane12 = lane * 12 + 2048;
  const __amdgpu_buffer_rsrc_t rq = sk_rsrc(q8, q8_bytes);
  const __amdgpu_buffer_rsrc_t rp = sk_rsrc(partials, (uint32_t)sk::MAX_WG * sk::SLOT_FLOATS * 4);
  const uint32_t pstride = (uint32_t)n_tt32 * sk::TILE_BYTES;

  v16i magic;
#pragma unroll
  for (int i = 0; i < 16; ++i) magic[i] = (int)MAGIC_I;

  // segments of [lo, hi) by unit, LAST first: a range's last segment is the only one that can stop short of its
  // unit's end (a head: published, never waited for); every other segment ends a unit (this workgroup writes it)
  int seg_hi = hi;
  int seg_no = 0;
  SK_STAMP(0);
  while (seg_hi > lo) {
    SK_STAMP(1 + 5 * seg_no);
    const int u = (seg_hi - 1) / P, ubase = u * P;
    const int seg_lo = max(lo, ubase);
    const int p0 = seg_lo - ubase, p1 = seg_hi - ubase;
    seg_hi = seg_lo;

    const int n0 = (u / n_tok_tiles) * 32, t0 = (u % n_tok_tiles) * (32 * TB);
    const int rmax = min(31, n_rows - 1 - n0);
    // rows past the tile's last valid row and bytes past a row's end read as zeros (or as in-tile garbage
    // that only ever lands in ring space no pair reads)
    const __amdgpu_buffer_rsrc_t rw = sk_rsrc(w + (int64_t)n0 * row_bytes, (uint32_t)(rmax + 1) * row_bytes);
    const int len4 = (p1 - p0) >> 2;
    const int pa = p0 + ks * len4, pb = pa + len4;   // this wave's pairs of the segment

    v16f acc[TB];
#pragma unroll
    for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[jj][i] = 0.0f;

    {
      // window (stage s, slot m) = rows 4m..4m+3 of super-block s
      auto load_window = [&](uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b128(rw, lane_src, soff, 0); };
      const int st_first = pa >> 2, q_first = pa & 3;
      {
        uint8_t* dst = ring_dst + (st_first & 1) * STAGE;
        const uint32_t so = (uint32_t)st_first * BS;
#pragma unroll
        for (int m = 0; m < 8; ++m) *(v4u*)(dst + m * 4 * PITCH) = load_window(so + m * 4 * row_bytes);
        // a wave that starts inside a stage: the slots of the next stage its skipped pairs would have parked
        uint8_t* dst1 = ring_dst + ((st_first + 1) & 1) * STAGE;
        for (int m = 0; m < 2 * q_first; ++m) *(v4u*)(dst1 + m * 4 * PITCH) = load_window(so + BS + m * 4 * row_bytes);
      }
      v4u wq[2];   // the two windows in flight: slot q of the stage after the current pair's
      {
        const uint32_t so = (uint32_t)(st_first + 1) * BS + (uint32_t)(2 * q_first) * 4 * row_bytes;
        wq[0] = load_window(so); wq[1] = load_window(so + 4 * row_bytes);
      }
      // ---- activations of pair pa: operand fragments and fp32 scales ----
      const int tt0 = t0 >> 5;
      uint32_t joff[TB];   // tile of token block jj relative to the pair's first tile (clamped inside the batch)
#pragma unroll
      for (int jj = 0; jj < TB; ++jj) joff[jj] = (uint32_t)(min(tt0 + jj, n_tt32 - 1) * sk::TILE_BYTES);
      uint32_t soff_b = (uint32_t)pa * pstride;   // tile row of the current pair
      v4u B[TB][2];
      v3u sc0[TB], sc1[TB];
#pragma unroll
      for (int jj = 0; jj < TB; ++jj) {
        B[jj][0] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane16, soff_b + joff[jj], 0);
        B[jj][1] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane16 + 1024, soff_b + joff[jj], 0);
        sc0[jj] = SK_LD_SC(rq, lane12, soff_b + joff[jj]);
        sc1[jj] = sc0[jj];
      }
#ifdef GGQ_SK_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SK_STAMP(2 + 5 * seg_no);
#endif
      v4i qh16 = {};

      // one pair; scc = its scales, scn = where the next pair's go.  (Called for even and odd pairs in turn, so
      // the two register sets swap roles without copies.)
      auto pair_body = [&](int p, v3u (&scc)[TB], v3u (&scn)[TB]) __attribute__((always_inline)) {
        const int st = p >> 2, q = p & 3;
        const int cur = (st & 1) * STAGE;
        // keep the 0x4B400000 accumulator input resident and never rewritten: hipcc otherwise re-creates it from
        // SGPRs every pair and reuses its registers in between — e.g. for the -M·d8 product 7 wait states after the
        // int8 MFMA that reads them as SrcC, which on gfx950 is too early: the MFMA's late passes then read the
        // product instead of the constant in the last quarter-wave (observed: nondeterministic results, lanes 48-63)
        asm volatile("" : "+v"(magic));
#ifdef SK_DBG_WAIT
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
        if (q == 0 || p == pa) {
          // ---- scales / mins of super-block st of row r: lane (r, h) decodes groups h, 2+h, 4+h, 6+h
          //      (get_scale_min_k4, HK/ggml/dequantize.cuh:154-161, four bytes at a time) ----
          const uint8_t* stage = ring + cur + r * PITCH;
          const v4i hdr = *(const v4i*)stage;
          if constexpr (T == GGQ_TYPE_Q5_K) qh16 = *(const v4i*)(stage + off::Q5_K_QH + 16 * h);
          const uint32_t s0 = (uint32_t)hdr[1], s1 = (uint32_t)hdr[2], s2 = (uint32_t)hdr[3];
          const float dall = bits_h_f32((uint32_t)hdr[0] & 0xFFFF), dmin = bits_h_f32((uint32_t)hdr[0] >> 16);
          const uint32_t sh = 8 * h;
          const uint32_t scA = (s0 & 0x3F3F3F3Fu) >> sh, mA = (s1 & 0x3F3F3F3Fu) >> sh;
          const uint32_t scB = ((s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u)) >> sh;
          const uint32_t mB = (((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u)) >> sh;
          const float scf[4] = {(float)(scA & 0xFF), (float)((scA >> 16) & 0xFF), (float)(scB & 0xFF), (float)((scB >> 16) & 0xFF)};
          const float mf[4] = {(float)(mA & 0xFF), (float)((mA >> 16) & 0xFF), (float)(mB & 0xFF), (float)((mB >> 16) & 0xFF)};
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            sbuf[qq * 64 + lane] = dall * scf[qq];          // exact: 11 x 6 significant bits; [q][h][r] = [q][lane]
            sbuf[256 + qq * 64 + lane] = -(dmin * mf[qq]);
          }
          __builtin_amdgcn_wave_barrier();
        }
        // ring: park the two windows requested one pair ago (slot q of stage st + 1), request the next two
        {
          uint8_t* dst = ring_dst + ((STAGE - cur) + q * (8 * PITCH));
          *(v4u*)dst = wq[0];
          *(v4u*)(dst + 4 * PITCH) = wq[1];
          const int p1n = p + 1;
          const uint32_t so = (uint32_t)((p1n >> 2) + 1) * BS + (uint32_t)(p1n & 3) * (8 * row_bytes);
          wq[0] = load_window(so); wq[1] = load_window(so + 4 * row_bytes);
        }
        // the next pair's scales first (oldest loads of the iteration); past the K range the scratch holds the
        // quantised zero padding of ggml_mul_mat_a8 (>= 4 pairs), so the prefetch never leaves it
        soff_b += pstride;
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) scn[jj] = SK_LD_SC(rq, lane12, soff_b + joff[jj]);
        __builtin_amdgcn_sched_barrier(0);   // the prefetches above are issued here, not where the scheduler likes them
#ifdef SK_DBG_VMWAIT
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SK_DBG_VMWAIT == 1 ? 2 + TB : 0) : "memory");
#endif
        // ---- A fragments: lane (r, h) holds bytes 16h..16h+15 of the pair's 32 nibble bytes:
        //      low nibbles = its K-half of group 2p, high nibbles = of group 2p+1 ----
        const v4i qs16 = *(const v4i*)(ring_qs + (cur + 32 * q));
        const float smq = sbuf[256 + q * 64 + lane];
        v4i a[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[0][i] = (int)((uint32_t)qs16[i] & 0x0F0F0F0Fu);
          a[1][i] = (int)(((uint32_t)qs16[i] >> 4) & 0x0F0F0F0Fu);
          if constexpr (T == GGQ_TYPE_Q5_K) {
            a[0][i] |= (int)((((uint32_t)qh16[i] >> (2 * q)) & 0x01010101u) << 4);
            a[1][i] |= (int)((((uint32_t)qh16[i] >> (2 * q + 1)) & 0x01010101u) << 4);
          }
        }
        const float* sap = sbuf + (q * 64 + 4 * h);   // rows 8qd + 4h + e of group gg at sap[gg * 32 + 8 qd + e]
        v4f sa[4];   // row scales of the current group as the accumulator registers see them
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) sa[qd] = *(const v4f*)(sap + 8 * qd);
        constexpr int CB = SK_CBUF;   // result buffers: 2 = the MFMA of tile-group t+1 is issued before the FMAs of t
        v16i c[CB];
        if (CB == 2) c[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], (v4i)B[0][0], magic, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int gg = t / TB, jj = t % TB;
          if (CB == 2 && t + 1 < NT) {
            const int g1 = (t + 1) / TB, j1 = (t + 1) % TB;
            c[(t + 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g1], (v4i)B[j1][g1], magic, 0, 0, 0);
          }
          if (CB == 1) c[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[gg], (v4i)B[jj][gg], magic, 0, 0, 0);
#ifdef SK_DBG_NOP_M
          asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(c[0]) :: "memory");
#endif
#ifndef SK_DBG_LATE_REFILL
          B[jj][gg] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane16 + 1024 * gg, soff_b + joff[jj], 0);   // next pair's
#endif
          __builtin_amdgcn_sched_barrier(0);
          const float d8 = as_f32((int)scc[jj][gg]);   // (by value: see as_f32)
          const float nm = -(MAGIC_F * d8);   // exact: d8 is an fp16 value
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[jj][i] = __builtin_fmaf(__builtin_fmaf(as_f32(c[t & (CB - 1)][i]), d8, nm), sa[i >> 2][i & 3], acc[jj][i]);
          if (t == TB - 1) {   // the last token block of group 0 is done: group 1's row scales
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) sa[qd] = *(const v4f*)(sap + 32 + 8 * qd);
          }
#ifndef SK_DBG_NO_MINTERM
          if (gg == 0)   // min term Σ (-dmin·m)[row, 2p+h] · s8[token, 2p+h] on the matrix pipe
            acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(smq, as_f32((int)scc[jj][2]), acc[jj], 0, 0, 0);
#endif
#ifdef SK_DBG_NOP_T
          asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[jj]) :: "memory");
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SK_DBG_LATE_REFILL
#pragma unroll
        for (int jj = 0; jj < TB; ++jj)
#pragma unroll
          for (int gg = 0; gg < 2; ++gg)
            B[jj][gg] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane16 + 1024 * gg, soff_b + joff[jj], 0);
#endif
      };
      int p = pa;
      while (true) {
        pair_body(p, sc0, sc1);
        if (++p >= pb) break;
        pair_body(p, sc1, sc0);
        if (++p >= pb) break;
      }
    }

#ifdef SK_DBG_DUMP   // pre-reduction accumulators of workgroups < 256 into the partial-sum area (scripts/det_sk.py)
    if (wg < 256) {
#pragma unroll
      for (int jj = 0; jj < TB; ++jj)
#pragma unroll
        for (int i = 0; i < 16; ++i) partials[((int64_t)(wg * 4 + ks) * (TB * 16) + jj * 16 + i) * 64 + lane] = acc[jj][i];
    }
#endif
    // ---- K-slice reduction: every wave parks its tile, wave ks sums registers [ks NQ, (ks+1) NQ) ----
    SK_STAMP(3 + 5 * seg_no);
#ifdef SK_NOP_BEFORE_RED
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
    __syncthreads();   // all rings dead
    float* red = (float*)lds;   // [wave][jj * 16 + i][lane]
#pragma unroll
    for (int jj = 0; jj < TB; ++jj)
#pragma unroll
      for (int i = 0; i < 16; ++i) red[((ks * TB + jj) * 16 + i) * 64 + lane] = acc[jj][i];
    __syncthreads();
    float v[NQ];
#pragma unroll
    for (int x = 0; x < NQ; ++x) {
      const int e = ks * NQ + x;
      v[x] = (red[e * 64 + lane] + red[(TB * 16 + e) * 64 + lane]) + (red[(2 * TB * 16 + e) * 64 + lane] + red[(3 * TB * 16 + e) * 64 + lane]);
    }
    SK_STAMP(4 + 5 * seg_no);
    // partial-sum slot of workgroup g: [wave][x / 4][lane][4] fp32; written and read with write-through / L2-level
    // (sc1) accesses only, so no cache write-back or invalidate is needed around the flag (MI355X_MICROARCH.md,
    // "Valid forms": all stores sc1 and drained, flag after the workgroup barrier, all loads sc1)
    auto slot_off = [&](int g, int x4) { return (uint32_t)((g * sk::SLOT_FLOATS + ((ks * (NQ / 4) + x4) * 64 + lane) * 4) * 4); };
    constexpr int SC1 = 16;   // aux cache-policy bit of the raw buffer builtins: sc1
    if (p1 < P) {
      // ---- head of a unit: publish ----
#pragma unroll
      for (int x4 = 0; x4 < NQ / 4; ++x4) {
        const v4f val = {v[4 * x4], v[4 * x4 + 1], v[4 * x4 + 2], v[4 * x4 + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, val), rp, slot_off(wg, x4), 0, SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      bool poisoned = false;
      if (p0 > 0) {
        // ---- tail of a unit whose head(s) other workgroups computed: the workgroups from the one that holds the
        //      unit's first pair up to wg - 1 (each published before doing anything else) ----
        int fw = (int)(((int64_t)(ubase >> 2) * G) / total4);
        while (fw > 0 && bound(fw) > ubase) --fw;
        while (bound(fw + 1) <= ubase) ++fw;
        for (int wp = fw; wp < wg; ++wp) {
          if (bound(wp) == bound(wp + 1)) continue;   // empty range: published nothing
          if (lane == 0) {
            int spins = 0;
            while (__hip_atomic_load(flags + wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1 << 24)) break;   // never hang the GPU on a lost producer: poison the unit instead
            }
            if (spins > (1 << 24)) poisoned = true;
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int x4 = 0; x4 < NQ / 4; ++x4) {
            const v4f val = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rp, slot_off(wp, x4), 0, SC1));
            v[4 * x4] += val[0]; v[4 * x4 + 1] += val[1]; v[4 * x4 + 2] += val[2]; v[4 * x4 + 3] += val[3];
          }
        }
        poisoned = __builtin_amdgcn_readfirstlane((int)poisoned) != 0;
        __syncthreads();   // every wave has read the slots
        if (tid == 0)
          for (int wp = fw; wp < wg; ++wp) __hip_atomic_store(flags + wp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (poisoned) {
#pragma unroll
        for (int x = 0; x < NQ; ++x) v[x] = __builtin_nanf("");
      }
      // ---- write back: register e = jj 16 + i <-> token t0 + 32 jj + r, row n0 + 8 (i >> 2) + 4 h + (i & 3) ----
      const bool vec_ok = DT != GGQ_F32 && (ldy & 3) == 0 && ((uintptr_t)y & 7) == 0 && n0 + 32 <= n_rows;
#pragma unroll
      for (int x4 = 0; x4 < NQ; x4 += 4) {
        const int e = ks * NQ + x4, jj = e >> 4, qd = (e & 15) >> 2;
        const int t = t0 + 32 * jj + r, row = n0 + 8 * qd + 4 * h;
        if (t < batch) {
          if (vec_ok) {
            uint16_t hv[4];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              if (DT == GGQ_F16) hv[e4] = __builtin_bit_cast(uint16_t, (_Float16)v[x4 + e4]);
              else hv[e4] = Elem<GGQ_BF16>::cvt(v[x4 + e4]);
            }
            uint2 pk;
            pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
            pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
            *(uint2*)((uint16_t*)y + (int64_t)t * ldy + row) = pk;
          } else {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
              if (row + e4 < n_rows) Elem<DT>::st(y, (int64_t)t * ldy + row + e4, v[x4 + e4]);
          }
        }
      }
    }
    __syncthreads();   // the reduction buffer aliases the rings of the next segment
    SK_STAMP(5 + 5 * seg_no);
    ++seg_no;
  }
}

template <int T, int DT, int TB>
static int launch_sk_tb(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy,
                        hipStream_t s) {
  using L = SkLaunch<T, TB>;
  const int64_t n_tok_tiles = (batch + 32 * TB - 1) / (32 * TB);
  const int64_t n_units = ((n + 31) / 32) * n_tok_tiles;
  const int64_t total4 = n_units * (k / 256);
  if (total4 * 4 >= 0x7fffffffLL || n_units > 0x7fffffffLL / 64) return GGQ_ERR_SHAPE;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess) return GGQ_ERR_LAUNCH;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return GGQ_ERR_LAUNCH;
  int64_t G = (int64_t)cus * L::WPC;
  if (G > sk::MAX_WG) G = sk::MAX_WG;
  if (total4 < G) G = total4;
  G = (G + 7) / 8 * 8;
  auto kern = mmq_sk_kernel<T, DT, TB>;
  if (L::LDS > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS) != hipSuccess)
    return GGQ_ERR_LAUNCH;
  uint8_t* base = (uint8_t*)const_cast<void*>(q8);
  int* flags = (int*)(base + sk::tiles_bytes(batch, k));
  float* partials = (float*)(base + sk::tiles_bytes(batch, k) + sk::FLAG_BYTES);
  (void)hipGetLastError();   // a stale error of an earlier, unrelated call is not this launch's
  const int64_t tile_bytes = ((batch + 31) / 32) * (sk::padded_k(k) / 64) * sk::TILE_BYTES;   // 32-bit buffer offsets
  if (tile_bytes >= 0x7fffffffLL || (int64_t)32 * ggq_row_bytes(T, k) >= 0x7fffffffLL) return GGQ_ERR_SHAPE;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(256), L::LDS, s, (const uint8_t*)w, (const uint8_t*)q8, y, (int)k,
                     (int)n, (int)batch, ldy, (int)n_tok_tiles, (int)n_units, (int)G, (uint32_t)tile_bytes, partials,
                     flags);
  GGQ_HIP_CHECK_LAUNCH();
  return GGQ_OK;
}

template <int T, int DT>
static int launch_sk_dt(const void* w, const void* q8, void* y, int64_t batch, int64_t k, int64_t n, int64_t ldy,
                        hipStream_t s) {
  int tb = batch <= 32 ? 1 : 2;
#ifdef GGQ_TUNING
  if (const char* e = getenv("GGQ_SK_TB")) tb = atoi(e);
#endif
  if (tb == 1) return launch_sk_tb<T, DT, 1>(w, q8, y, batch, k, n, ldy, s);
  if (tb == 2) return launch_sk_tb<T, DT, 2>(w, q8, y, batch, k, n, ldy, s);
  return launch_sk_tb<T, DT, 4>(w, q8, y, batch, k, n, ldy, s);
}

template <int T>
static int launch_sk_t(const void* w, const void* q8, void* y, int dt, int64_t batch, int64_t k, int64_t n,
                       int64_t ldy, hipStream_t s) {
  switch (dt) {
    case GGQ_F32: return launch_sk_dt<T, GGQ_F32>(w, q8, y, batch, k, n, ldy, s);
    case GGQ_F16: return launch_sk_dt<T, GGQ_F16>(w, q8, y, batch, k, n, ldy, s);
    case GGQ_BF16: return launch_sk_dt<T, GGQ_BF16>(w, q8, y, batch, k, n, ldy, s);
    default: return GGQ_ERR_DTYPE;
  }
}

// the stream-K matmul on a LAYOUT 3 scratch (arguments already validated by ggq_mul_mat_q_pretiled)
int mmq_sk_launch(const void* w, const void* q8, void* y, int type, int dt, int64_t batch, int64_t k, int64_t n,
                  int64_t ldy, hipStream_t s) {
  switch (type) {
    case GGQ_TYPE_Q4_K: return launch_sk_t<GGQ_TYPE_Q4_K>(w, q8, y, dt, batch, k, n, ldy, s);
    case GGQ_TYPE_Q5_K: return launch_sk_t<GGQ_TYPE_Q5_K>(w, q8, y, dt, batch, k, n, ldy, s);
    default: return GGQ_ERR_TYPE;
  }
}

}  // namespace ggq
