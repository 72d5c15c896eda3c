// custom_ops.cpp — Python module `custom_ops` with the reference's CPU op surface:
//   ggml_dequantize(W: Tensor, type: int, m: int, n: int) -> Tensor  (fp32, CPU)
// mirrors ggml-cpu/custom_ops.cpp:11-40 of the reference (pybind11 function, same name and
// argument meaning) over the C ABI ggq_cpu_dequantize_f32.  Differences: validates its
// input and raises on an unsupported type instead of returning uninitialised memory.
#include <torch/extension.h>

#include "../../../include/ggq.h"

// the reference loop is single-threaded (ggml-cpu/ggml-quants.hpp); so is the op (no environment is read: a caller that
// wants threads calls ggq_cpu_dequantize_f32_ex through the C ABI)
static int default_threads() { return 1; }

static torch::Tensor ggml_dequantize(const torch::Tensor W, int type, int64_t m, int64_t n) {
  TORCH_CHECK(W.device().is_cpu(), "custom_ops.ggml_dequantize: W must be a CPU tensor");
  TORCH_CHECK(W.is_contiguous(), "custom_ops.ggml_dequantize: W must be contiguous");
  TORCH_CHECK(m >= 0 && n >= 0, "custom_ops.ggml_dequantize: negative shape");
  const int64_t rb = ggq_row_bytes(type, m * n);
  TORCH_CHECK(rb >= 0, "custom_ops.ggml_dequantize: ", ggq_strerror((int)rb), " (type ", type, ", m*n ", m * n, ")");
  TORCH_CHECK((int64_t)W.nbytes() == rb, "custom_ops.ggml_dequantize: W holds ", W.nbytes(),
              " bytes, expected ", rb, " for type ", type, " and shape [", m, ", ", n, "]");
  torch::Tensor out = torch::empty({m, n}, torch::TensorOptions().dtype(torch::kFloat32));
  const int rc = ggq_cpu_dequantize_f32(W.data_ptr(), out.data_ptr<float>(), type, m, n, default_threads());
  TORCH_CHECK(rc == GGQ_OK, "custom_ops.ggml_dequantize: ", ggq_strerror(rc), " (type ", type, ")");
  return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
  mod.def("ggml_dequantize", &ggml_dequantize, "dequantize GGML tensor");
}
