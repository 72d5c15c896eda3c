// binding.cpp — the drop-in operator boundary: torch.ops.<ext>.{ggml_dequantize,
// ggml_mul_mat_vec_a8, ggml_mul_mat_a8} with the reference's exact schemas
// (HK/torch-ext/torch_binding.cpp:6-21, prototypes HK/torch-ext/torch_binding.h:6-15),
// implemented on PyTorch-ROCm's CUDA dispatch key over the C ABI of libggq_hip
// (include/ggq.h).  This file owns allocation, device guard, stream lookup and the
// error -> TORCH_CHECK translation; it contains no kernels.
#include <Python.h>
#include <torch/all.h>
#include <torch/library.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "../../../include/ggq.h"

#ifndef TORCH_EXTENSION_NAME
#define TORCH_EXTENSION_NAME _ggml
#endif

namespace {

int ggq_dtype_of(const torch::Tensor& X, const char* op) {
  switch (X.scalar_type()) {
    case at::ScalarType::Float: return GGQ_F32;
    case at::ScalarType::Half: return GGQ_F16;
    case at::ScalarType::BFloat16: return GGQ_BF16;
    default:
      TORCH_CHECK(false, op, ": X must be float32, float16 or bfloat16 (HK/ggml/dispatch_utils.h:14-20), got ",
                  X.scalar_type());
  }
  return -1;
}

void check_weight(const torch::Tensor& W, int64_t type, int64_t rows, int64_t cols, const char* op) {
  TORCH_CHECK(W.is_cuda(), op, ": W must live on the GPU");
  TORCH_CHECK(W.is_contiguous(), op, ": W must be contiguous");
  TORCH_CHECK(ggq_type_supported((int)type), op, ": unsupported ggml quantisation type ", type);
  const int64_t rb = ggq_row_bytes((int)type, cols);
  TORCH_CHECK(rb >= 0, op, ": K = ", cols, " is not a multiple of the block size ", ggq_block_elems((int)type));
  TORCH_CHECK((int64_t)W.nbytes() == rows * rb, op, ": W holds ", W.nbytes(), " bytes, expected ", rows * rb,
              " (", rows, " rows x ", rb, " bytes) for type ", type);
}

hipStream_t current_stream(const torch::Tensor& t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

}  // namespace

// HK/ggml/ggml_kernel.cu:68-78 — fp16 [m,n] whatever the caller's dtype
torch::Tensor ggml_dequantize(torch::Tensor W, int64_t type, int64_t m, int64_t n) {
  TORCH_CHECK(m >= 0 && n >= 0, "ggml_dequantize: negative shape");
  TORCH_CHECK(ggq_type_supported((int)type), "ggml_dequantize: unsupported ggml quantisation type ", type);
  TORCH_CHECK((m * n) % ggq_block_elems((int)type) == 0, "ggml_dequantize: m*n = ", m * n,
              " is not a multiple of the block size ", ggq_block_elems((int)type));
  check_weight(W, type, 1, m * n, "ggml_dequantize");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(W));
  auto options = torch::TensorOptions().dtype(torch::kFloat16).device(W.device());
  at::Tensor DW = torch::empty({m, n}, options);
  const int rc = ggq_dequantize_f16(W.data_ptr(), DW.data_ptr(), (int)type, m, n, current_stream(W));
  TORCH_CHECK(rc == GGQ_OK, "ggml_dequantize: ", ggq_strerror(rc));
  return DW;
}

// HK/ggml/ggml_kernel.cu:80-193 — X [1, K], Y [1, row] in X's dtype
torch::Tensor ggml_mul_mat_vec_a8(torch::Tensor W, torch::Tensor X, int64_t type, int64_t row) {
  TORCH_CHECK(X.dim() == 2 && X.size(0) == 1, "ggml_mul_mat_vec_a8: X must have shape [1, hidden_size]");
  TORCH_CHECK(X.is_cuda() && X.device() == W.device(), "ggml_mul_mat_vec_a8: X and W must be on the same GPU");
  const int dt = ggq_dtype_of(X, "ggml_mul_mat_vec_a8");
  const int64_t col = X.size(1);
  TORCH_CHECK(row >= 0, "ggml_mul_mat_vec_a8: negative row count");
  check_weight(W, type, row, col, "ggml_mul_mat_vec_a8");
  X = X.contiguous();
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(X));
  at::Tensor Y = torch::empty({1, row}, torch::TensorOptions().dtype(X.dtype()).device(W.device()));
  const int64_t padded = ggq_mmvq_padded_k(col);
  at::Tensor quant_X = torch::empty({1, padded / 32 * 9}, torch::TensorOptions().dtype(torch::kInt32).device(W.device()));
  const int rc = ggq_mul_mat_vec_q(W.data_ptr(), X.data_ptr(), Y.data_ptr(), (int)type, dt, col, row,
                                   quant_X.data_ptr(), current_stream(X));
  TORCH_CHECK(rc == GGQ_OK, "ggml_mul_mat_vec_a8: ", ggq_strerror(rc));
  return Y;
}

// HK/ggml/mmq.cu:180-255 — X [tokens, K] or [batch, tokens, K]
torch::Tensor ggml_mul_mat_a8(torch::Tensor W, torch::Tensor X, int64_t type, int64_t row) {
  const int64_t x_ndim = X.dim();
  TORCH_CHECK(x_ndim == 2 || x_ndim == 3,
              "X must have shape [num_tokens, hidden_size] or [batch_size, num_tokens, hidden_size]");
  TORCH_CHECK(X.is_cuda() && X.device() == W.device(), "ggml_mul_mat_a8: X and W must be on the same GPU");
  const int dt = ggq_dtype_of(X, "ggml_mul_mat_a8");
  const int64_t col = X.size(x_ndim - 1);
  TORCH_CHECK(row >= 0, "ggml_mul_mat_a8: negative row count");
  TORCH_CHECK(ggq_mmq_type_supported((int)type), "ggml_mul_mat_a8: no quantised GEMM for ggml type ", type,
              " (the reference's switch has the same ten cases, HK/ggml/mmq.cu:222-251)");
  check_weight(W, type, row, col, "ggml_mul_mat_a8");
  X = X.contiguous();
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(X));
  auto options = torch::TensorOptions().dtype(X.dtype()).device(W.device());
  at::Tensor Y;
  int64_t batch;
  if (x_ndim == 2) {
    batch = X.size(0);
    Y = torch::empty({batch, row}, options);
  } else {
    batch = X.size(0) * X.size(1);
    Y = torch::empty({X.size(0), X.size(1), row}, options);
  }
  if (batch == 0 || row == 0) return Y;
  // reference: {batch, padded/32*9} ints (mmq.cu:208); here the same bytes per token with the batch rounded up to whole
  // 32-token tiles (the fragment-major layouts address tiles): ggq_mmq_scratch_bytes
  const int64_t scratch_ints = (int64_t)((ggq_mmq_scratch_bytes(batch, col) + 3) / 4);
  at::Tensor quant_X = torch::empty({scratch_ints}, torch::TensorOptions().dtype(torch::kInt32).device(W.device()));
  const int rc = ggq_mul_mat_q(W.data_ptr(), X.data_ptr(), Y.data_ptr(), (int)type, dt, batch, col, row,
                               quant_X.data_ptr(), current_stream(X));
  TORCH_CHECK(rc == GGQ_OK, "ggml_mul_mat_a8: ", ggq_strerror(rc));
  return Y;
}

// TORCH_LIBRARY does not macro-expand its name argument; go through one level of indirection.
#define GGQ_TORCH_LIBRARY(NAME, MODULE) TORCH_LIBRARY(NAME, MODULE)
GGQ_TORCH_LIBRARY(TORCH_EXTENSION_NAME, ops) {
  // Dequantization for GGML.
  ops.def("ggml_dequantize(Tensor W, int type, SymInt m, SymInt n) -> Tensor");
  ops.impl("ggml_dequantize", torch::kCUDA, &ggml_dequantize);
  // mmvq kernel for GGML.
  ops.def("ggml_mul_mat_vec_a8(Tensor W, Tensor X, int type, SymInt row) -> Tensor");
  ops.impl("ggml_mul_mat_vec_a8", torch::kCUDA, &ggml_mul_mat_vec_a8);
  // mmq kernel for GGML.
  ops.def("ggml_mul_mat_a8(Tensor W, Tensor X, int type, SymInt row) -> Tensor");
  ops.impl("ggml_mul_mat_a8", torch::kCUDA, &ggml_mul_mat_a8);
}

// Lets `import _ggml` load the shared object (the reference's REGISTER_EXTENSION,
// HK/torch-ext/registration.h:25-30).
#define GGQ_CONCAT_(A, B) A##B
#define GGQ_CONCAT(A, B) GGQ_CONCAT_(A, B)
#define GGQ_STR_(A) #A
#define GGQ_STR(A) GGQ_STR_(A)
PyMODINIT_FUNC GGQ_CONCAT(PyInit_, TORCH_EXTENSION_NAME)() {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, GGQ_STR(TORCH_EXTENSION_NAME), nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
