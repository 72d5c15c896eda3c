"""ggml — MI355X-native drop-in for the reference's ``ggml`` kernel package.

The public surface is the reference's (HK/torch-ext/ggml/__init__.py:15-44 of Isotr0py/ggml-libtorch): three
functions with the same names, positional arguments and the batch-1 assertion of the GEMV wrapper.  Each one
forwards to the operator of the same name that ``_ggml*.so`` registers under ``torch.ops._ggml`` (hand-written
gfx950 HIP kernels behind the C ABI of include/ggq.h); there is no Python or CPU fallback.
"""
import torch

from ._ops import ops

__all__ = ["ggml_dequantize", "ggml_mul_mat_vec_a8", "ggml_mul_mat_a8", "ops"]


def _type_id(quant_type) -> int:
    # callers pass gguf.GGMLQuantizationType members (IntEnum) as well as plain ints
    return int(quant_type)


def ggml_dequantize(W: torch.Tensor, quant_type: int, m: int, n: int) -> torch.Tensor:
    """Block-quantised bytes ``W`` (uint8, ``m`` rows of ``n`` elements) -> fp16 tensor ``[m, n]`` on W's device."""
    return ops.ggml_dequantize(W, _type_id(quant_type), m, n)


def ggml_mul_mat_vec_a8(W: torch.Tensor, X: torch.Tensor, quant_type: int, row: int) -> torch.Tensor:
    """``X [1, K] · Wᵀ`` with Q8_1-quantised activations (quantised GEMV); ``row`` = rows of W.  Batch must be 1."""
    if X.size(0) != 1:
        raise AssertionError("Batch size must be 1 for MMVQ kernel")
    return ops.ggml_mul_mat_vec_a8(W, X, _type_id(quant_type), row)


def ggml_mul_mat_a8(W: torch.Tensor, X: torch.Tensor, quant_type: int, row: int) -> torch.Tensor:
    """``X [batch, K]`` (or ``[b, t, K]``) ``· Wᵀ`` with Q8_1-quantised activations (int8-MFMA GEMM), any batch."""
    return ops.ggml_mul_mat_a8(W, X, _type_id(quant_type), row)
