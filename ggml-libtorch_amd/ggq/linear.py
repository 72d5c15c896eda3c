"""Caller-side helpers on top of the C ABI: quantise the activations once, multiply by several weight matrices.

The reference quantises X inside every ggml_mul_mat_a8 call (HK/ggml/mmq.cu:208-230), so an FFN block that
multiplies the same X by gate and up projections quantises it twice.  `QuantizedActivations` keeps the
fragment-major Q8_1 scratch of one X and runs the streamed MMQ kernel against any number of weight matrices of
the same need_sum class.  Results are bit-identical to `ggml.ggml_mul_mat_a8` for the same (W, X) wherever the
op itself takes the streamed kernel (batch >= 5; >= 33 for Q6_K, >= 65 for Q8_0: `ggq_mul_mat_q_ld` in mmq.hip) —
there it is the same two kernels with the first one hoisted; at smaller batches the op runs its dot4 / LDS-tile
kernels, whose fp32 summation order differs (same 1e-3 contract).  (caller-side layer.)
"""
import ctypes

import torch

from . import lib as ggqlib
from .formats import NEED_SUM


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


class QuantizedActivations:
    """Q8_1 (fragment-major) form of X [batch, K] for the formats whose scratch stores half2(d, sum)
    (`need_sum=True`: Q4_0, Q4_1, Q5_1, Q4_K, Q5_K) or float d (`need_sum=False`: the others)."""

    def __init__(self, x: torch.Tensor, need_sum: bool):
        if not x.is_cuda or x.dim() != 2 or not x.is_contiguous():
            raise ValueError("x must be a contiguous [batch, K] tensor on the GPU")
        self.L = ggqlib.hip()
        self.x_dtype = x.dtype
        self.batch, self.k = x.shape
        self.need_sum = bool(need_sum)
        self.scratch = torch.empty(max(16, int(self.L.ggq_mmq_scratch_bytes(self.batch, self.k))), dtype=torch.uint8,
                                   device=x.device)
        self.device = x.device
        rep = 12 if need_sum else 8   # any format of the class selects the layout of the ds words (Q4_K / Q8_0)
        with torch.cuda.device(x.device):   # the C ABI launches on the current device (binding.cpp holds a device guard)
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            ggqlib.check(self.L.ggq_quantize_q8_1_tiled(_vp(x), ggqlib.dtype_code(x.dtype), _vp(self.scratch), self.batch,
                                                        self.k, rep, stream), "ggq_quantize_q8_1_tiled")

    def matmul(self, w: torch.Tensor, quant_type: int, rows: int, out: torch.Tensor = None) -> torch.Tensor:
        """Y [batch, rows] = X · W^T for one block-quantised weight matrix (raw GGUF bytes, uint8 on the GPU)."""
        t = int(quant_type)
        if (t in {int(q) for q in NEED_SUM}) != self.need_sum:
            raise ValueError("this weight format needs the other Q8_1 scratch flavour (need_sum mismatch)")
        if not self.L.ggq_mmq_tiled_supported(t, self.k):
            raise ValueError(f"type {t} with K={self.k} is not handled by the streamed kernel")
        if w.device != self.device or (out is not None and out.device != self.device):
            raise ValueError(f"weights / output must live on the activations' device {self.device}")
        y = out if out is not None else torch.empty((self.batch, rows), dtype=self.x_dtype, device=self.device)
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            ggqlib.check(self.L.ggq_mul_mat_q_pretiled(_vp(w), _vp(self.scratch), _vp(y), t, ggqlib.dtype_code(self.x_dtype),
                                                       self.batch, self.k, rows, y.stride(0), stream), "ggq_mul_mat_q_pretiled")
        return y


def gate_up(x: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor, quant_type: int, rows: int):
    """(x·W_gate^T, x·W_up^T) with one activation quantisation — the FFN shape of the benchmark (K=4096, N=11008)."""
    qa = QuantizedActivations(x, int(quant_type) in {int(q) for q in NEED_SUM})
    return qa.matmul(w_gate, quant_type, rows), qa.matmul(w_up, quant_type, rows)
