"""CPU suite: the `custom_ops` module (reference CPU op surface, ggml-cpu/custom_ops.cpp) —
reads like tests/test_dequantize.py of the reference with the GGUF samples replaced by
synthetic blocks and gguf.dequantize by the golden vectors / oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

from ggq import synth, lib as ggqlib
from ggq.formats import GGMLType
from oracle import ggq_numpy as N

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QUANT_TYPES = [GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0]
HIDDEN_SIZES = [256, 1024]


@pytest.fixture(scope="module")
def ops():
    import custom_ops
    return custom_ops


@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("quant_type", QUANT_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_dequantize(ops, hidden_size, quant_type):
    for rows in (hidden_size, 96):
        data = synth.random_weight(quant_type, rows, hidden_size, seed=rows)
        ref_output = torch.tensor(N.gguf_dequantize(data, quant_type).reshape(rows, hidden_size))
        output = ops.ggml_dequantize(torch.tensor(data, device="cpu"), quant_type, rows, hidden_size)
        assert output.dtype == torch.float32 and output.shape == (rows, hidden_size)
        torch.testing.assert_close(output, ref_output, atol=1e-2, rtol=4e-2)


@pytest.mark.parametrize("quant_type", QUANT_TYPES, ids=lambda t: t.name)
def test_bit_exact_vs_reference_golden(ops, quant_type):
    g = np.load(os.path.join(GOLD, "reference_cpu_dequant.npz"))
    blocks, bits = g[f"{quant_type.name}_blocks"], g[f"{quant_type.name}_f32_bits"]
    out = ops.ggml_dequantize(torch.from_numpy(blocks.reshape(1, -1).copy()), int(quant_type), 1, blocks.shape[0] * 32)
    assert np.array_equal(out.numpy().reshape(-1).view(np.uint32), bits)


@pytest.mark.parametrize("quant_type", QUANT_TYPES, ids=lambda t: t.name)
def test_threaded_equals_single_thread(quant_type):
    L = ggqlib.cpu()
    w = synth.random_weight(quant_type, 64, 1024, seed=1)
    a = np.empty((64, 1024), np.float32)
    b = np.empty_like(a)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    assert L.ggq_cpu_dequantize_f32(p(w), p(a), int(quant_type), 64, 1024, 1) == 0
    assert L.ggq_cpu_dequantize_f32(p(w), p(b), int(quant_type), 64, 1024, 4) == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("quant_type", QUANT_TYPES, ids=lambda t: t.name)
def test_vector_path_equals_scalar_path_bit_for_bit(quant_type):
    """ggq_cpu_dequantize_f32_ex(simd=1 / 2) (AVX-512 / AVX2 where the host has them) vs simd=0 (the reference's scalar loops) on random
    and edge blocks, incl. NaN / inf / subnormal scales: identical bit patterns, so the golden test above pins both."""
    L = ggqlib.cpu()
    blocks = np.concatenate([synth.random_blocks(quant_type, 1500, seed=2), synth.edge_blocks(quant_type)])
    nb = blocks.shape[0]
    a = np.empty(nb * 32, np.float32)
    b = np.empty_like(a)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    assert L.ggq_cpu_dequantize_f32_ex(p(blocks), p(a), int(quant_type), 1, nb * 32, 1, 0) == 0
    assert L.ggq_cpu_dequantize_f32_ex(p(blocks), p(b), int(quant_type), 1, nb * 32, 3, 1) == 0     # widest unit (AVX-512 / AVX2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    b[:] = 0
    assert L.ggq_cpu_dequantize_f32_ex(p(blocks), p(b), int(quant_type), 1, nb * 32, 1, 2) == 0     # at most AVX2
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert L.ggq_cpu_simd_name() in (b"avx512", b"avx2", b"scalar")


def test_errors(ops):
    w = torch.from_numpy(synth.random_weight(GGMLType.Q4_0, 4, 64, seed=0))
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w, int(GGMLType.Q4_K), 4, 64)      # the reference returns garbage here
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w, int(GGMLType.Q4_0), 4, 128)     # byte count mismatch
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w.T, int(GGMLType.Q4_0), 4, 64)    # non-contiguous
    assert ops.ggml_dequantize(w[:0], int(GGMLType.Q4_0), 0, 64).shape == (0, 64)


# ---- host twin of the quantised GEMM (ggq_cpu_quantize_q8_1_mmq + ggq_cpu_mul_mat_q): bench.py's threaded CPU column ----
MMQ_CPU_TYPES = [GGMLType.Q4_K, GGMLType.Q4_0, GGMLType.Q8_0]


def _cpu_mmq(w, x, t, n_rows, threads, simd):
    L = ggqlib.cpu()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    batch, k = x.shape
    q = np.zeros(((k - k % 512 + 512) // 128) * batch * 144, np.uint8)
    assert L.ggq_cpu_quantize_q8_1_mmq(p(x), p(q), batch, k, int(t)) == 0
    y = np.full((batch, n_rows), np.nan, np.float32)
    assert L.ggq_cpu_mul_mat_q(p(w), p(q), p(y), int(t), batch, k, n_rows, threads, simd) == 0
    return q, y


@pytest.mark.parametrize("quant_type", MMQ_CPU_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("shape", [(1, 256, 3), (5, 512, 17), (8, 1280, 64), (33, 2048, 40)], ids=str)
def test_cpu_mmq_bit_exact_vs_oracle(quant_type, shape):
    """every SIMD path and the threaded split give the oracle's bits: integer dots are exact, the float sequence is the oracle's"""
    from oracle import oracle as O
    batch, k, n_rows = shape
    w = synth.random_weight(quant_type, n_rows, k, seed=k + batch)
    x = np.random.default_rng(batch).standard_normal((batch, k)).astype(np.float32)
    x[0, :32] = 0.0                                              # an all-zero group (d = 0)
    want_q = O.quantize_q8_1_mmq(x, quant_type)
    want_y, _ = O.mul_mat_q(w, x, quant_type, n_rows)
    for threads, simd in ((1, 0), (1, 1), (1, 2), (3, 1)):
        q, y = _cpu_mmq(w, x, quant_type, n_rows, threads, simd)
        assert np.array_equal(q, want_q), (threads, simd)
        assert np.array_equal(y.view(np.uint32), want_y.view(np.uint32)), (threads, simd)
    assert ggqlib.cpu().ggq_cpu_mmq_simd_name() in (b"avx512-vnni", b"avx2", b"scalar")


def test_cpu_mmq_extreme_bytes():
    """Q8_0 weights of -128 / 127 against activations of +-127: the unsigned x signed byte-dot must not saturate"""
    from oracle import oracle as O
    k, n_rows = 512, 4
    w = synth.random_weight(GGMLType.Q8_0, n_rows, k, seed=0).reshape(n_rows, k // 32, 34).copy()
    w[0, :, 2:] = 0x80
    w[1, :, 2:] = 0x7F
    w[2, :, 2::2] = 0x80
    w = w.reshape(n_rows, -1)
    x = np.ones((3, k), np.float32)
    x[1] = -1.0
    x[2, ::3] = -1.0
    want, _ = O.mul_mat_q(w, x, GGMLType.Q8_0, n_rows)
    for simd in (0, 1, 2):
        _, y = _cpu_mmq(w, x, GGMLType.Q8_0, n_rows, 1, simd)
        assert np.array_equal(y.view(np.uint32), want.view(np.uint32)), simd


def test_cpu_mmq_errors():
    L = ggqlib.cpu()
    assert L.ggq_cpu_mul_mat_q(None, None, None, int(GGMLType.Q6_K), 1, 256, 1, 1, 1) == -1    # only the three headline formats
    assert L.ggq_cpu_mul_mat_q(None, None, None, int(GGMLType.Q4_K), 1, 100, 1, 1, 1) == -2
    assert L.ggq_cpu_mul_mat_q(None, None, None, int(GGMLType.Q4_K), 1, 256, 1, 1, 1) == -4
    assert L.ggq_cpu_mul_mat_q(None, None, None, int(GGMLType.Q4_K), 0, 256, 1, 1, 1) == 0
    assert L.ggq_cpu_quantize_q8_1_mmq(None, None, 1, 256, 9) == -1
