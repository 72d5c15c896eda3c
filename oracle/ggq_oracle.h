/*
 * ggq_oracle.h — CPU restatement of the reference's algorithm for the
 * ggml block-quant hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product path (ggml-libtorch_amd/) never links,
 * imports or executes anything under oracle/.
 *
 * Pinning status:
 *   - dequantize_row_{q4_0,q4_1,q5_0,q5_1,q8_0} -> fp32: PINNED bit-exact against
 *     the reference's own compiled ggml-cpu op (oracle/_ref, tests/golden/*.npz).
 *   - fp16 (GPU-semantics) dequantize, K-quant dequantize, quantize_q8_1,
 *     mul_mat_vec_q, mul_mat_q: PARITY UNPINNED — the reference's GPU path cannot
 *     run here (CUDA) and ships no golden vectors; these restatements are
 *     cross-checked by an independent numpy derivation (oracle/ggq_numpy.py).
 */
#ifndef GGQ_ORACLE_H
#define GGQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* fp16 <-> fp32 (IEEE binary16, round-to-nearest-even) */
uint16_t oracle_f32_to_f16(float f);
float oracle_f16_to_f32(uint16_t h);
uint16_t oracle_f32_to_bf16(float f);
float oracle_bf16_to_f32(uint16_t h);

int oracle_block_elems(int type);
int oracle_block_bytes(int type);

/* ggml-cpu semantics (fp32 out): ggml-cpu/ggml-quants.hpp:4-112. Legacy formats only.
 * returns 0, or -1 for a type the reference's CPU op does not handle. */
int oracle_dequantize_row_f32(int type, const void* w, float* y, int64_t k);

/* GPU semantics (fp16 out, every __h* intrinsic = one IEEE fp16 op):
 * HK/ggml/dequantize.cuh:3-254, and :399-433 for IQ4_NL (20) / IQ4_XS (23).  y holds raw fp16 bits. */
int oracle_dequantize_row_f16(int type, const void* w, uint16_t* y, int64_t k);

/* Exact-arithmetic (double) dequantisation, one value per element, no intermediate
 * rounding: the mathematical definition in SURVEY §2.2 — used for the integer-unpack
 * range checks and as an independent sanity reference. */
int oracle_dequantize_row_f64(int type, const void* w, double* y, int64_t k);

/* quantize_q8_1 (HK/ggml/ggml_kernel.cu:13-66): x fp32 [batch,k] -> block_q8_1[batch][padded/32],
 * padded = roundup(k,512). */
void oracle_quantize_q8_1(const float* x, void* q, int64_t batch, int64_t k);

/* quantize_mmq_q8_1 (HK/ggml/mmq.cu:109-177): block_q8_1_mmq, index (k/128)*batch + token,
 * padded = k - k%512 + 512. need_sum selects half2(d,sum) vs float d. */
void oracle_quantize_q8_1_mmq(const float* x, void* q, int64_t batch, int64_t k, int need_sum);

/* mul_mat_vec_q (HK/ggml/mmvq.cuh:2-38) with the per-format vec_dot_*_q8_1
 * (HK/ggml/vecdotq.cuh:43-605; :842-888 for IQ4_NL / IQ4_XS).  q8: block_q8_1 array of one row (batch 1).
 * y[n_rows] fp32 (before the cast to the output dtype); yabs[n_rows] = sum of |lane terms|
 * (tolerance scale for fp-accumulate comparisons; may be NULL). */
int oracle_mul_mat_vec_q(int type, const void* w, const void* q8, float* y, float* yabs,
                         int64_t k, int64_t n_rows);

/* mul_mat_q (HK/ggml/mmq.cuh:1917-1986) with the tensor-core vec_dot_*_q8_1_mma bodies
 * (dp4a bodies for Q2_K/Q3_K, which have no mma variant: mmq.cuh:1862-1876).
 * q8: block_q8_1_mmq scratch from oracle_quantize_q8_1_mmq.  y[batch][n_rows] fp32. */
int oracle_mul_mat_q(int type, const void* w, const void* q8, float* y, float* yabs,
                     int64_t batch, int64_t k, int64_t n_rows);

#ifdef __cplusplus
}
#endif
#endif
