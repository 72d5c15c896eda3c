"""ggml — MI355X-native drop-in for the reference's ``ggml`` kernel package.

Same three wrappers, names, argument meaning and assertions as
HK/torch-ext/ggml/__init__.py:15-44 of Isotr0py/ggml-libtorch, backed by
hand-written gfx950 HIP kernels registered as ``torch.ops._ggml.*``.
"""
import torch

from ._ops import ops


def ggml_dequantize(
    W: torch.Tensor,
    quant_type: int,
    m: int,
    n: int,
) -> torch.Tensor:
    """Dequantize the GGML tensor (fp16 [m, n] on the GPU)."""
    return ops.ggml_dequantize(W, int(quant_type), m, n)


def ggml_mul_mat_vec_a8(
    W: torch.Tensor,
    X: torch.Tensor,
    quant_type: int,
    row: int,
) -> torch.Tensor:
    """Mulmat with MMVQ kernel, require batch_size==1."""
    batch = X.size(0)
    assert batch == 1, "Batch size must be 1 for MMVQ kernel"
    return ops.ggml_mul_mat_vec_a8(W, X, int(quant_type), row)


def ggml_mul_mat_a8(
    W: torch.Tensor,
    X: torch.Tensor,
    quant_type: int,
    row: int,
) -> torch.Tensor:
    """Mulmat through MMQ kernel for arbitrary batch size."""
    return ops.ggml_mul_mat_a8(W, X, int(quant_type), row)


__all__ = ["ggml_dequantize", "ggml_mul_mat_vec_a8", "ggml_mul_mat_a8", "ops"]
