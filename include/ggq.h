/*
 * ggq.h — C ABI of the MI355X-native ggml block-quant hot path
 * (dequantize + Q8_1 activation quantise + MMVQ GEMV + MMQ GEMM).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.
 * Every entry point names the reference interface it replaces
 * (paths relative to the reference repo, HK/ = hf-kernels/ggml-kernels/).
 *
 * Conventions
 *   - `type` is the ggml type id (HK/ggml/ggml-common.h:1128-1161):
 *     Q4_0=2 Q4_1=3 Q5_0=6 Q5_1=7 Q8_0=8 Q2_K=10 Q3_K=11 Q4_K=12 Q5_K=13 Q6_K=14,
 *     and for dequantise + MMVQ only (as in the reference, whose ggml_mul_mat_a8 has no IQ case,
 *     HK/ggml/mmq.cu:222-251) the nine IQ formats of its dispatch (HK/ggml/ggml_kernel.cu:145-189):
 *     IQ2_XXS=16 IQ2_XS=17 IQ3_XXS=18 IQ1_S=19 IQ4_NL=20 IQ3_S=21 IQ2_S=22 IQ4_XS=23 IQ1_M=29.
 *   - W is the raw GGUF tensor payload: `n_rows` rows, each `k/qk` blocks,
 *     row-major, device memory for the ggq_* (GPU) calls.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     All GPU calls are asynchronous on that stream; no host sync, no allocation.
 *   - Return value: GGQ_OK (0) or a negative GGQ_ERR_* code. Never exits,
 *     never throws across the ABI.
 */
#ifndef GGQ_H
#define GGQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGQ_ABI_VERSION 10   /* 10: ggq_mmq_x64_tile_tokens; 9: ggq_mmq_x64_unit_rows, ggq_mul_mat_vec_q_gather, ggq_mul_mat_q_gather on the streamed route; 8: ggq_*_x64 (64 x 64 wave tiles, batches from 33 tokens), GGQ_MMQ_ROUTE_X64; 2: IQ4_NL / IQ4_XS for dequantise + MMVQ, ggq_mmq_type_supported; 3: ggq_peer_*; 4: ggq_*_t16, ggq_mmq_route; 5: ggq_peer_scatter / _wait; 6: ggq_mul_mat_q_gather; 7: ggq_mmq_stream_unit_tokens */

/* ggml type ids (HK/ggml/ggml-common.h:1128-1161) */
enum ggq_type {
  GGQ_TYPE_Q4_0 = 2,
  GGQ_TYPE_Q4_1 = 3,
  GGQ_TYPE_Q5_0 = 6,
  GGQ_TYPE_Q5_1 = 7,
  GGQ_TYPE_Q8_0 = 8,
  GGQ_TYPE_Q8_1 = 9, /* activations only */
  GGQ_TYPE_Q2_K = 10,
  GGQ_TYPE_Q3_K = 11,
  GGQ_TYPE_Q4_K = 12,
  GGQ_TYPE_Q5_K = 13,
  GGQ_TYPE_Q6_K = 14,
  /* the IQ formats: dequantise + MMVQ only, as in the reference (HK/ggml/ggml_kernel.cu:145-189) */
  GGQ_TYPE_IQ2_XXS = 16,
  GGQ_TYPE_IQ2_XS = 17,
  GGQ_TYPE_IQ3_XXS = 18,
  GGQ_TYPE_IQ1_S = 19,
  GGQ_TYPE_IQ4_NL = 20,
  GGQ_TYPE_IQ3_S = 21,
  GGQ_TYPE_IQ2_S = 22,
  GGQ_TYPE_IQ4_XS = 23,
  GGQ_TYPE_IQ1_M = 29
};

/* activation / output element types (HK/ggml/dispatch_utils.h:14-20) */
enum ggq_dtype { GGQ_F32 = 0, GGQ_F16 = 1, GGQ_BF16 = 2 };

enum ggq_status {
  GGQ_OK = 0,
  GGQ_ERR_TYPE = -1,   /* unsupported ggml type id            */
  GGQ_ERR_SHAPE = -2,  /* k not a multiple of the block size… */
  GGQ_ERR_DTYPE = -3,  /* unsupported activation dtype        */
  GGQ_ERR_ARG = -4,    /* null pointer / negative size        */
  GGQ_ERR_LAUNCH = -5, /* HIP runtime reported an error       */
  GGQ_ERR_ALIGN = -6   /* pointer alignment below the contract */
};

/* ------------------------------------------------------------------ */
/* Format traits (host side, no GPU needed)                            */
/* ------------------------------------------------------------------ */

int ggq_abi_version(void);
const char* ggq_strerror(int status);

/* replaces ggml_get_block_size (HK/ggml/mmq.cu:57-81): elements per block, 0 if unsupported */
int ggq_block_elems(int type);
/* bytes per block (HK/ggml/ggml-common.h:17-108 struct sizes), 0 if unsupported */
int ggq_block_bytes(int type);
/* bytes of one weight row of k elements; <0 on error */
int64_t ggq_row_bytes(int type, int64_t k);
/* 1 when dequantize / MMVQ kernels exist for `type` */
int ggq_type_supported(int type);
/* 1 when the MMQ GEMM (ggq_mul_mat_q*, ggq_quantize_q8_1_mmq / _tiled) exists for `type`: the ten formats of the
 * reference's ggml_mul_mat_a8 switch (HK/ggml/mmq.cu:222-251) */
int ggq_mmq_type_supported(int type);
/* mmq_need_sum (HK/ggml/mmq.cu:84-106): 1 if the MMQ scratch stores half2(d,sum), 0 if float d */
int ggq_mmq_need_sum(int type);

/* K padding rules of the reference's scratch buffers.
 *   MMVQ: roundup(k,512)          (HK/ggml/ggml_kernel.cu:84)
 *   MMQ : k - k%512 + 512         (HK/ggml/mmq.cu:190-191) */
int64_t ggq_mmvq_padded_k(int64_t k);
int64_t ggq_mmq_padded_k(int64_t k);
/* scratch bytes = batch * padded/32 * 36 (HK/ggml/ggml_kernel.cu:90, mmq.cu:208); the MMQ
 * figure rounds batch up to a multiple of 32 (whole token tiles of the fragment-major layout) */
size_t ggq_mmvq_scratch_bytes(int64_t k);
size_t ggq_mmq_scratch_bytes(int64_t batch, int64_t k);

/* ------------------------------------------------------------------ */
/* GPU hot path (gfx950)                                               */
/* ------------------------------------------------------------------ */

/* replaces ggml_dequantize (HK/ggml/ggml_kernel.cu:68-78) + ggml_get_to_fp16_cuda
 * (HK/ggml/dequantize.cuh:525-568). w: m*n/qk blocks; out: m*n fp16. */
int ggq_dequantize_f16(const void* w, void* out_f16, int type, int64_t m, int64_t n,
                       void* stream);

/* replaces quantize_row_q8_1_cuda (HK/ggml/ggml_kernel.cu:13-66).
 * x: [batch,k] of x_dtype; q: batch * padded/32 block_q8_1 {half d; half sum; int8 qs[32]},
 * padded = ggq_mmvq_padded_k(k), padding quantised from zeros. */
int ggq_quantize_q8_1(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                      void* stream);

/* replaces quantize_mmq_q8_1_cuda (HK/ggml/mmq.cu:109-177).
 * q: block_q8_1_mmq {half2 ds[4]; int8 qs[128]} (HK/ggml/mmq.cuh:176-181),
 * block index = (k/128)*batch + token; ds slot holds half2(d,sum) when
 * ggq_mmq_need_sum(type) else float d; padded = ggq_mmq_padded_k(k). */
int ggq_quantize_q8_1_mmq(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                          int type, void* stream);

/* replaces ggml_mul_mat_vec_a8 (HK/ggml/ggml_kernel.cu:80-193) = quantize_q8_1 ∘ mul_mat_vec_q
 * (HK/ggml/mmvq.cuh:2-128). x: [1,k] dtype; y: [1,n_rows] dtype;
 * scratch: >= ggq_mmvq_scratch_bytes(k) device bytes, 16-byte aligned. */
int ggq_mul_mat_vec_q(const void* w, const void* x, void* y, int type, int dtype,
                      int64_t k, int64_t n_rows, void* scratch, void* stream);

/* replaces ggml_mul_mat_a8 (HK/ggml/mmq.cu:180-255) = quantize_mmq_q8_1 ∘ mul_mat_q
 * (HK/ggml/mmq.cuh:1917-2031). x: [batch,k]; y: [batch,n_rows] row-major (ldy = n_rows);
 * scratch: >= ggq_mmq_scratch_bytes(batch,k) device bytes, 16-byte aligned.
 * Reproducibility: the integer contraction is exact; the fp32 accumulation ORDER is fixed per (type, batch, k, n_rows) —
 * repeated launches, row permutations and token permutations of one call are bit-identical — but the kernel and its K-slice
 * count are chosen from the whole shape (ggq_mmq_route), so a row of a row-sharded matrix may differ from the same row of the
 * unsharded one in the last fp32 rounding (within the 1e-3 canon, never beyond it). */
int ggq_mul_mat_q(const void* w, const void* x, void* y, int type, int dtype,
                  int64_t batch, int64_t k, int64_t n_rows, void* scratch, void* stream);

/* Same as ggq_mul_mat_q but writes y with a row pitch of ldy elements at column
 * offset 0 — used by the row-sharded multi-GPU path to write a rank's [batch, n_rows]
 * slab straight into its slot of the gathered [batch, ldy] output. */
int ggq_mul_mat_q_ld(const void* w, const void* x, void* y, int type, int dtype,
                     int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                     void* scratch, void* stream);

/* mul_mat_q alone on an already-quantised scratch (layout of ggq_quantize_q8_1_mmq).
 * Lets a caller quantise X once and reuse it for several weight matrices. */
int ggq_mul_mat_q_prequant(const void* w, const void* q, void* y, int type, int dtype,
                           int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                           void* stream);

/* Fragment-major variant of the MMQ activation scratch (same 144 bytes per 128 elements and token,
 * same values as ggq_quantize_q8_1_mmq, regrouped per (k/128, token/32) into 4608-byte tiles
 * { int8 qs[4 groups][2 K-halves][32 tokens][16]; ds[2 group pairs][32 tokens][2] } so that one MFMA
 * operand fragment is 1 KB contiguous).  This is what ggq_mul_mat_q uses internally for the batches ggq_mmq_route() reports as
 * GGQ_MMQ_ROUTE_STREAM (the large ones), for every (type, k) ggq_mmq_tiled_supported() reports: all ten formats, k a whole
 * number of blocks, rows of at most 32 MiB; exported so a caller can quantise once for several weight
 * matrices (the reference quantises per call, HK/ggml/mmq.cu:208-230).
 * q: >= ggq_mmq_scratch_bytes(batch,k) bytes, 16-byte aligned.  w: 2-byte aligned as everywhere
 * (GGQ_ERR_ALIGN otherwise). */
int ggq_mmq_tiled_supported(int type, int64_t k);
int ggq_quantize_q8_1_tiled(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                            int type, void* stream);
int ggq_mul_mat_q_pretiled(const void* w, const void* q, void* y, int type, int dtype,
                           int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                           void* stream);

/* Fused epilogues (no reference counterpart: ggml_mul_mat_a8 has none; SURVEY §8f rank 3, the caller-side layer).
 * Applied to the fp32 accumulator in the streamed kernel's write-back, before the one rounding to `dtype`:
 *   GGQ_EPI_BIAS      y[t, r] = acc + bias[r]                 aux = bias: n_rows elements of `dtype`
 *   GGQ_EPI_SILU_MUL  y[t, r] = silu(gate[t, r]) * acc        aux = gate: [batch, ldy] of `dtype`, y's layout
 * (the FFN's silu(x W_gate^T) * (x W_up^T): run the gate matmul, then the up matmul with this epilogue).
 * ggq_mul_mat_q_epi = ggq_quantize_q8_1_tiled + ggq_mul_mat_q_pretiled_epi (always the streamed kernel). */
enum ggq_epilogue { GGQ_EPI_NONE = 0, GGQ_EPI_BIAS = 1, GGQ_EPI_SILU_MUL = 2 };
int ggq_mul_mat_q_pretiled_epi(const void* w, const void* q, void* y, int type, int dtype,
                               int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                               int epilogue, const void* aux, void* stream);
int ggq_mul_mat_q_epi(const void* w, const void* x, void* y, int type, int dtype,
                      int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                      int epilogue, const void* aux, void* scratch, void* stream);

/* 16-token-tile path for the HBM-bound batches (what ggq_mul_mat_q runs between the GEMV-like batches and the 32-token
 * MFMA tile for the formats ggq_mmq_t16_type_supported() reports; same role as mul_mat_q's small mmq_x instances,
 * HK/ggml/kernel_instances/mmq_kernel.cuh:21-32 + mmq.cuh:1917-1986).  The scratch holds the same Q8_1 values as
 * ggq_quantize_q8_1_mmq, regrouped per (k/256, token/16) into 4608-byte tiles { int8 frag[4][4 K-chunks][16 tokens][16];
 * half2(d, sum) | float d  ds[2 halves][4 token quads][4 groups][4 tokens] } in the element order the weight format's
 * MFMA operand needs (one operand fragment = 1 KB contiguous).  q: >= ggq_mmq_scratch_bytes(batch,k) bytes, 16-byte
 * aligned.  ggq_mul_mat_q_t16 takes the epilogue arguments of ggq_mul_mat_q_pretiled_epi.
 * Formats: Q4_K Q5_K (batch <= 32) and Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 Q6_K Q3_K (batch <= 16; Q6_K / Q3_K tensors of 1 KB .. 4 GiB); k a multiple of 256
 * (ggq_mmq_t16_supported; GGQ_ERR_SHAPE otherwise).  Batches of at most 8 use the 8-token form of the tile (2304 bytes). */
int ggq_mmq_t16_type_supported(int type);
int ggq_mmq_t16_supported(int type, int64_t k, int64_t batch);
int ggq_quantize_q8_1_t16(const void* x, int x_dtype, void* q, int64_t batch, int64_t k,
                          int type, void* stream);
int ggq_mul_mat_q_t16(const void* w, const void* q, void* y, int type, int dtype,
                      int64_t batch, int64_t k, int64_t n_rows, int64_t ldy,
                      int epilogue, const void* aux, void* stream);

/* 64-row x 64-token wave tiles for the batches from 33 tokens up — and, as 32-row x 32-token one-tile units, for 17 - 32 — (round 4; the role of mul_mat_q's mmq_x = 64 ... 128 instances,
 * HK/ggml/kernel_instances/mmq_kernel.cuh:21-32 + mmq.cuh:1917-1986, at the reference benchmark's own batch sizes,
 * benchmarks/benchmark_mmq.py:152).  K loop in hand-scheduled gfx950 assembly (scripts/gen_mmq_x64.py).  The scratch ("x64 layout",
 * ggq_quantize_q8_1_x64) holds the same Q8_1 values as ggq_quantize_q8_1_mmq, regrouped per (k/256, token/32) into 10240-byte records
 * { int8 frag[8 groups][2 K-halves][32 tokens][16];  float d8[8 groups][32 tokens] (the fp16-rounded d for the need_sum formats), tokens
 * in accumulator-register order;  fp16 s8 operand of the min-term MFMA [2][32 tokens][8] }, token tiles padded to a multiple of 64
 * tokens.  q: >= ggq_mmq_scratch_bytes(batch, k) bytes, 16-byte aligned.  Formats: ggq_mmq_x64_type_supported(); k a multiple of 256,
 * scratch below 4 GiB (ggq_mmq_x64_supported; GGQ_ERR_SHAPE otherwise).  Epilogue arguments as ggq_mul_mat_q_pretiled_epi. */
int ggq_mmq_x64_type_supported(int type);
int ggq_mmq_x64_supported(int type, int64_t k, int64_t batch);
/* K-slices (= waves) per 64 x 64 unit the x64 kernel uses for a shape: 8 while there is at most one unit per CU, 1 from 2048 units (a full
 * round of one-wave workgroups: the large batches; 64-row units only), else 4 (32-row units: 8 / 4 by the same rule on their count; 96-row
 * units: always 4).  Host-only. */
int ggq_mmq_x64_k_slices(int64_t batch, int64_t k, int64_t n_rows);
/* Weight rows per unit: 32 below 160 units of 64 rows (Q5_K: always), else 64, or 96 where that makes the launch one even round of at most 256
 * workgroups.  A row's bits do not depend on it (at equal K-slice count).  Host-only. */
int ggq_mmq_x64_unit_rows(int type, int64_t batch, int64_t k, int64_t n_rows);
/* Tokens per wave tile: 32 (one MFMA tile per wave and group, 32-row units) up to 32 tokens and, at 33 - 64 tokens, while the launch then
 * has at most 256 units; else 64.  Host-only. */
int ggq_mmq_x64_tile_tokens(int type, int64_t batch, int64_t k, int64_t n_rows);
int ggq_quantize_q8_1_x64(const void* x, int x_dtype, void* q, int64_t batch, int64_t k, int type, void* stream);
int ggq_mul_mat_q_x64(const void* w, const void* q, void* y, int type, int dtype, int64_t batch, int64_t k, int64_t n_rows,
                      int64_t ldy, int epilogue, const void* aux, void* stream);

/* Which kernel ggq_mul_mat_q / _ld / _epi run for a (type, batch, k, n_rows): the role of the reference's tile
 * heuristic (mul_mat_q_case + get_mmq_x_max_host, HK/ggml/kernel_instances/mmq_kernel.cuh:21-32, mmq.cuh:155-164).
 * Host-only, no GPU needed.  DOT4 = batch <= 8 GEMV-like kernel, LDS_TILE = barrier-coupled kernel on the reference
 * layout, STREAM = 32 / 64-token MFMA units on the fragment-major scratch, T16 = 16-token tiles (ggq_mul_mat_q_t16). */
enum ggq_mmq_route_id { GGQ_MMQ_ROUTE_NONE = 0, GGQ_MMQ_ROUTE_DOT4 = 1, GGQ_MMQ_ROUTE_LDS_TILE = 2, GGQ_MMQ_ROUTE_STREAM = 3,
                        GGQ_MMQ_ROUTE_T16 = 4, GGQ_MMQ_ROUTE_X64 = 5 };
int ggq_mmq_route(int type, int64_t batch, int64_t k, int64_t n_rows);
/* Tokens per workgroup unit (32 or 64) the STREAM route uses for a (type, batch, n_rows): the other half of the tile heuristic's role
 * (mmq_x of mul_mat_q_case, HK/ggml/kernel_instances/mmq_kernel.cuh:21-32).  Host-only. */
int ggq_mmq_stream_unit_tokens(int type, int64_t batch, int64_t n_rows);

/* mul_mat_vec_q alone on an already-quantised scratch (layout of ggq_quantize_q8_1). */
int ggq_mul_mat_vec_q_prequant(const void* w, const void* q, void* y, int type, int dtype,
                               int64_t k, int64_t n_rows, void* stream);

/* ------------------------------------------------------------------ */
/* Host CPU twin (product CPU op, libggq_cpu)                          */
/* ------------------------------------------------------------------ */

/* replaces the ggml-cpu op: ggml_dequantize (ggml-cpu/custom_ops.cpp:11-36) +
 * dequantize_row_q{4_0,4_1,5_0,5_1,8_0} (ggml-cpu/ggml-quants.hpp:4-112).
 * fp32 output, host memory. nthreads <= 1 reproduces the reference's single-thread loop. */
int ggq_cpu_dequantize_f32(const void* w, float* out, int type, int64_t m, int64_t n,
                           int nthreads);
/* the same with the vector path selectable: simd = 0 forces the scalar loops (the reference's form), 2 at most AVX2,
 * any other value the widest unit the host has (AVX-512 F+BW+VL: sixteen elements per instruction; AVX2: eight) —
 * integer subtract, exact int -> float conversion, one multiply, one add per element in every path: bit-identical results.
 * ggq_cpu_simd_name(): "avx512", "avx2" or "scalar". */
int ggq_cpu_dequantize_f32_ex(const void* w, float* out, int type, int64_t m, int64_t n,
                              int nthreads, int simd);
const char* ggq_cpu_simd_name(void);

/* Host twin of the quantised GEMM for Q4_K, Q4_0 and Q8_0 (the reference has no CPU matmul: its ggml-cpu op only
 * dequantises, ggml-cpu/custom_ops.cpp:16-34; this is the host baseline of the headline metric).  Same arithmetic as the GPU
 * path's canon: ggq_cpu_quantize_q8_1_mmq = quantize_mmq_q8_1 (HK/ggml/mmq.cu:109-154; q: block_q8_1_mmq layout,
 * batch * padded/32 * 36 bytes), ggq_cpu_mul_mat_q = the float sequences of the tensor-core bodies (HK/ggml/mmq.cuh:330-394,
 * 913-974, 1274-1363) on exact integer dots; y: fp32 [batch, n_rows].  simd: 0 scalar, 2 at most AVX2, otherwise the widest
 * of AVX-512 VNNI / AVX2 the host has — all paths bit-identical.  ggq_cpu_mmq_simd_name(): "avx512-vnni", "avx2", "scalar". */
int ggq_cpu_quantize_q8_1_mmq(const float* x, void* q, int64_t batch, int64_t k, int type);
int ggq_cpu_mul_mat_q(const void* w, const void* q, float* y, int type, int64_t batch, int64_t k, int64_t n_rows,
                      int nthreads, int simd);
const char* ggq_cpu_mmq_simd_name(void);

/* ---- peer-mapped output slabs (multi-GPU, one process per GPU; no reference counterpart: the reference has no
 * multi-device code).  A rank exports the gather buffer it allocated as 64 opaque bytes + the byte offset of
 * the pointer inside its allocation; the other ranks import it and may then pass the mapped pointer as the `y` / `dst` of
 * any entry point above: the matmul's slab is written straight into the peer's buffer over xGMI (device-to-device stores),
 * no collective and no staging copy.  ggq_peer_write_2d copies a [rows x row_bytes] slab with independent pitches on a
 * stream.  Ordering across processes is the caller's (an event / barrier after the stream has drained).
 * Needs dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0). */
int ggq_peer_export(const void* dev_ptr, void* handle_out_64_bytes, int64_t* offset_out);
int ggq_peer_import(const void* handle_64_bytes, int64_t offset, void** dev_ptr_out);
int ggq_peer_close(void* dev_ptr, int64_t offset);
int ggq_peer_write_2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t row_bytes,
                      int64_t rows, void* stream);
/* Device-side hand-off of the gather (ABI 5; no host barrier, no stream drain).  NOT replayable from a captured HIP graph:
 * `generation`, the buffer parity and the dst / flag pointers are host-computed arguments, so a replayed ggq_peer_wait would
 * find its generation already published and return at once (stale slabs, no error).  Enqueue the pair eagerly per gather.
 * The 2-second give-up of ggq_peer_wait only sets *status: poll it (PeerSlabGather.status(), checked by close()) before
 * trusting results produced behind a wait that may have timed out.
 * ggq_peer_scatter: one kernel stores the rank's [rows x row_bytes] slab into `dsts[0..n_dst)` (slot `rank` of each peer's
 *   buffer, mapped with ggq_peer_import; pitches in bytes, 16-byte multiples) and, once every workgroup has drained and
 *   released its stores at system scope, writes `generation` into `flags[d]` (a 4-byte word in peer d's memory, one per
 *   source rank).  `arrivals` is a zero-initialised 4-byte device word of the caller's own memory (workgroup count-in; the
 *   kernel leaves it zero).  `dsts` / `flags` are HOST arrays of device pointers, at most 8 peers.
 * ggq_peer_wait: one tiny kernel on `stream` that returns once the rank's own `flags[0..n_src)` all hold `generation`
 *   (signed wrap-around compare), then acquires at system scope: work enqueued behind it sees every peer's slab.  A peer
 *   that has not arrived after about two seconds sets *status (a 4-byte device word) to 1 instead of hanging the device.
 * Generations increase by one per gather; a buffer must not be rewritten while a peer may still read the previous
 * contents (ggq.dist.PeerSlabGather alternates two buffers). */
int ggq_peer_scatter(const void* src, int64_t src_pitch, void* const* dsts, void* const* flags, int n_dst,
                     int64_t dst_pitch, int64_t row_bytes, int64_t rows, uint32_t generation, void* arrivals,
                     void* stream);
int ggq_peer_wait(const void* flags, int n_src, uint32_t generation, void* status, void* stream);

/* The GEMM with a multi-destination write-back: ggq_mul_mat_q_ld + ggq_peer_scatter in ONE kernel.  Y (row pitch ldy elements) is
 * stored into dsts[0 .. n_dst) — dsts[0] the caller's own slot, the others the same slot of the peers' gather buffers — and once
 * every workgroup has released its stores at system scope `generation` is written into flags[0 .. n_flag) (ggq_peer_wait on the
 * consumer side, as after ggq_peer_scatter; `arrivals` as there).  dsts / flags: HOST arrays of device pointers, at most 8 each.
 * For the (type, batch, shape) ggq_mmq_route() sends to the 16-token-tile kernel (GGQ_MMQ_ROUTE_T16: the last workgroup publishes) or
 * to the streamed kernel (GGQ_MMQ_ROUTE_STREAM: every wave that stores arrives, the last arrival publishes) or to the 64 x 64 wave-tile
 * kernel (GGQ_MMQ_ROUTE_X64: a workgroup arrives once its stores have drained); GGQ_ERR_SHAPE
 * otherwise — the caller then falls back to ggq_mul_mat_q_ld into its own slot + ggq_peer_scatter.
 * ggq_mul_mat_vec_q_gather: the same for one token — the fused GEMV (ggq_mul_mat_vec_q) storing its n_rows outputs into
 * dsts[0 .. n_dst) and publishing from its last wave; every format of ggq_mul_mat_vec_q. */
int ggq_mul_mat_q_gather(const void* w, const void* x, void* const* dsts, int n_dst, void* const* flags, int n_flag,
                         uint32_t generation, void* arrivals, int type, int dtype, int64_t batch, int64_t k,
                         int64_t n_rows, int64_t ldy, void* scratch, void* stream);
int ggq_mul_mat_vec_q_gather(const void* w, const void* x, void* const* dsts, int n_dst, void* const* flags, int n_flag,
                             uint32_t generation, void* arrivals, int type, int dtype, int64_t k, int64_t n_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GGQ_H */
