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
