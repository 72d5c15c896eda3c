// traits.cpp — format traits / error strings of the C ABI (include/ggq.h).
// Host-only; compiled into both libggq_hip.so and libggq_cpu.so.
#include "../../../include/ggq.h"

extern "C" int ggq_abi_version(void) { return GGQ_ABI_VERSION; }

extern "C" const char* ggq_strerror(int status) {
  switch (status) {
    case GGQ_OK: return "ok";
    case GGQ_ERR_TYPE: return "unsupported ggml quantisation type";
    case GGQ_ERR_SHAPE: return "shape is not compatible with the block format";
    case GGQ_ERR_DTYPE: return "unsupported activation dtype (float32, float16, bfloat16 only)";
    case GGQ_ERR_ARG: return "invalid argument (null pointer or negative size)";
    case GGQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case GGQ_ERR_ALIGN: return "pointer alignment below the ABI contract";
    default: return "unknown ggq status";
  }
}

// ggml_get_block_size, HK/ggml/mmq.cu:57-81
extern "C" int ggq_block_elems(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_0: case GGQ_TYPE_Q5_1:
    case GGQ_TYPE_Q8_0: case GGQ_TYPE_Q8_1: case GGQ_TYPE_IQ4_NL:
      return 32;
    case GGQ_TYPE_Q2_K: case GGQ_TYPE_Q3_K: case GGQ_TYPE_Q4_K: case GGQ_TYPE_Q5_K:
    case GGQ_TYPE_Q6_K: case GGQ_TYPE_IQ4_XS:
      return 256;
    default: return 0;
  }
}

extern "C" int ggq_block_bytes(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: return 18;
    case GGQ_TYPE_Q4_1: return 20;
    case GGQ_TYPE_Q5_0: return 22;
    case GGQ_TYPE_Q5_1: return 24;
    case GGQ_TYPE_Q8_0: return 34;
    case GGQ_TYPE_Q8_1: return 36;
    case GGQ_TYPE_Q2_K: return 84;
    case GGQ_TYPE_Q3_K: return 110;
    case GGQ_TYPE_Q4_K: return 144;
    case GGQ_TYPE_Q5_K: return 176;
    case GGQ_TYPE_Q6_K: return 210;
    case GGQ_TYPE_IQ4_NL: return 18;    // block_iq4_nl, HK/ggml/ggml-common.h:176-182
    case GGQ_TYPE_IQ4_XS: return 136;   // block_iq4_xs, HK/ggml/ggml-common.h:184-191
    default: return 0;
  }
}

extern "C" int64_t ggq_row_bytes(int type, int64_t k) {
  const int qk = ggq_block_elems(type);
  if (qk == 0) return GGQ_ERR_TYPE;
  if (k < 0 || k % qk) return GGQ_ERR_SHAPE;
  return k / qk * ggq_block_bytes(type);
}

extern "C" int ggq_type_supported(int type) {
  return type != GGQ_TYPE_Q8_1 && ggq_block_elems(type) != 0;
}

// the ten cases of ggml_mul_mat_a8's switch (HK/ggml/mmq.cu:222-251): no IQ format
extern "C" int ggq_mmq_type_supported(int type) {
  return ggq_type_supported(type) && type != GGQ_TYPE_IQ4_NL && type != GGQ_TYPE_IQ4_XS;
}

// mmq_need_sum, HK/ggml/mmq.cu:84-106
extern "C" int ggq_mmq_need_sum(int type) {
  switch (type) {
    case GGQ_TYPE_Q4_0: case GGQ_TYPE_Q4_1: case GGQ_TYPE_Q5_1: case GGQ_TYPE_Q4_K:
    case GGQ_TYPE_Q5_K:
      return 1;
    default: return 0;
  }
}

extern "C" int64_t ggq_mmvq_padded_k(int64_t k) { return (k + 511) / 512 * 512; }   // ggml_kernel.cu:84
extern "C" int64_t ggq_mmq_padded_k(int64_t k) { return k - k % 512 + 512; }         // mmq.cu:190-191

extern "C" size_t ggq_mmvq_scratch_bytes(int64_t k) {
  return (size_t)(ggq_mmvq_padded_k(k) / 32 * 36);
}
extern "C" size_t ggq_mmq_scratch_bytes(int64_t batch, int64_t k) {
  // whole 32-token tiles: the fragment-major layout (ggq_quantize_q8_1_tiled) addresses tiles
  return (size_t)((batch + 31) / 32 * 32) * (size_t)(ggq_mmq_padded_k(k) / 32 * 36);
}
