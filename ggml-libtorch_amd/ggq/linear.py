"""Caller-side helpers on top of the C ABI: quantise the activations once, multiply by several weight matrices.

The reference quantises X inside every ggml_mul_mat_a8 call (HK/ggml/mmq.cu:208-230), so an FFN block that
multiplies the same X by gate and up projections quantises it twice.  `QuantizedActivations` keeps the
fragment-major Q8_1 scratch of one X and runs the streamed MMQ kernel against any number of weight matrices of
the same need_sum class.  Results are bit-identical to `ggml.ggml_mul_mat_a8` for the same (W, X) wherever the
op itself takes the streamed kernel (batch >= 5; >= 9 for Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q4_K, >= 33 for Q6_K, >= 65 for Q8_0:
`ggq_mul_mat_q_ld` in mmq.hip) —
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


EPI_NONE, EPI_BIAS, EPI_SILU_MUL = 0, 1, 2   # enum ggq_epilogue (include/ggq.h)


def _matmul_epi(qa: "QuantizedActivations", w, quant_type, rows, epi, aux, out=None):
    """qa.matmul with a fused epilogue of the streamed kernel's write-back (ggq_mul_mat_q_pretiled_epi)."""
    t = int(quant_type)
    if (t in {int(q) for q in NEED_SUM}) != qa.need_sum:
        raise ValueError("this weight format needs the other Q8_1 scratch flavour (need_sum mismatch)")
    if w.device != qa.device or aux.device != qa.device or aux.dtype != qa.x_dtype or not aux.is_contiguous():
        raise ValueError("weights and the epilogue operand must be contiguous, of X's dtype, on X's device")
    y = out if out is not None else torch.empty((qa.batch, rows), dtype=qa.x_dtype, device=qa.device)
    if out is not None and out.data_ptr() == aux.data_ptr():
        raise ValueError("the epilogue operand must not alias the output (the kernel declares both __restrict__)")
    if epi == EPI_BIAS and aux.numel() != rows:
        raise ValueError("bias must have one element per output row")
    if epi == EPI_SILU_MUL and (aux.shape != y.shape or aux.stride(0) != y.stride(0)):
        raise ValueError("the gate operand must have the output's shape and row pitch")
    with torch.cuda.device(qa.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(qa.device).cuda_stream)
        ggqlib.check(qa.L.ggq_mul_mat_q_pretiled_epi(_vp(w), _vp(qa.scratch), _vp(y), t, ggqlib.dtype_code(qa.x_dtype),
                                                     qa.batch, qa.k, rows, y.stride(0), epi, _vp(aux), stream),
                     "ggq_mul_mat_q_pretiled_epi")
    return y


class QuantLinear(torch.nn.Module):
    """y = x · W^T (+ bias) with W kept in its ggml block-quantised form (raw GGUF tensor bytes).

    The caller-side layer SURVEY §8f ranks third (the reference stops at the three ops; its callers,
    benchmarks/benchmark_mmq.py:80-87, call them bare).  One token goes through the GEMV op, more through the
    quantised GEMM; a bias is added inside the GEMM kernel's write-back (GGQ_EPI_BIAS: one rounding of acc + bias),
    not by a second elementwise kernel."""

    def __init__(self, weight: torch.Tensor, quant_type: int, in_features: int, out_features: int, bias: torch.Tensor = None):
        super().__init__()
        from .formats import row_bytes
        if weight.dtype != torch.uint8 or weight.numel() != out_features * row_bytes(quant_type, in_features):
            raise ValueError("weight must be the uint8 GGUF payload of an [out_features, in_features] tensor")
        self.quant_type, self.in_features, self.out_features = int(quant_type), in_features, out_features
        self.register_buffer("weight", weight.reshape(out_features, -1).contiguous())
        self.register_buffer("bias", None if bias is None else bias.contiguous())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        import ggml
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.in_features).contiguous()
        L = ggqlib.hip()
        mmq = bool(L.ggq_mmq_tiled_supported(self.quant_type, self.in_features))
        if x2.shape[0] == 1 or not mmq:   # GEMV op (also the only matmul of the IQ formats: one token at a time)
            rows = [ggml.ggml_mul_mat_vec_a8(self.weight, x2[i:i + 1], self.quant_type, self.out_features) for i in range(x2.shape[0])]
            y = torch.cat(rows, dim=0) if rows else x2.new_empty((0, self.out_features))
            if self.bias is not None:
                y = y + self.bias.to(y.dtype)
        elif self.bias is None:
            y = ggml.ggml_mul_mat_a8(self.weight, x2, self.quant_type, self.out_features)
        else:
            qa = QuantizedActivations(x2, self.quant_type in {int(q) for q in NEED_SUM})
            y = _matmul_epi(qa, self.weight, self.quant_type, self.out_features, EPI_BIAS, self.bias.to(x2.dtype))
        return y.reshape(*lead, self.out_features)


class QuantGatedFFN(torch.nn.Module):
    """down( silu(x W_gate^T) * (x W_up^T) ) — the Llama FFN the benchmark shapes come from (K = 4096, N = 11008).

    X is quantised once for gate and up (the reference would quantise it in each of the two ggml_mul_mat_a8 calls),
    and silu(gate) * up is the epilogue of the up matmul (GGQ_EPI_SILU_MUL): no elementwise kernel, no fp16
    rounding of `up` before the product.  Two launches + one quantisation instead of two ops + two elementwise kernels."""

    def __init__(self, w_gate, w_up, w_down, quant_type, hidden: int, intermediate: int, down_quant_type=None):
        super().__init__()
        self.quant_type, self.hidden, self.intermediate = int(quant_type), hidden, intermediate
        self.register_buffer("w_gate", w_gate.reshape(intermediate, -1).contiguous())
        self.register_buffer("w_up", w_up.reshape(intermediate, -1).contiguous())
        self.down = QuantLinear(w_down, self.quant_type if down_quant_type is None else down_quant_type, intermediate, hidden)

    def gate_up(self, x2: torch.Tensor) -> torch.Tensor:
        qa = QuantizedActivations(x2, self.quant_type in {int(q) for q in NEED_SUM})
        gate = qa.matmul(self.w_gate, self.quant_type, self.intermediate)
        return _matmul_epi(qa, self.w_up, self.quant_type, self.intermediate, EPI_SILU_MUL, gate)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.hidden).contiguous()
        return self.down(self.gate_up(x2)).reshape(*lead, self.hidden)
