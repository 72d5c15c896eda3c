"""The reference's harness, as its own files spell it (missing 4): the import line
``from gguf import GGMLQuantizationType, GGUFReader, ReaderTensor, dequantize`` + ``import custom_ops as ops`` (tests/test_dequantize.py:6-8)
or ``import ggml as ops`` (hf-kernels/ggml-kernels/tests/kernels/test_cuda_kernels.py:3-5), a ``./samples``-style directory of
``Quant_{TYPE}_{hidden}.gguf`` files, the same calls and the same tolerances.  gguf-py is not installed here: tests/compat/gguf stands in
for those four names (reader = the product's, ``dequantize`` = the numpy checker) and steps aside when the real package is importable.
The sample files are synthetic (no network): written by ggq.gguf_io.write_sample_file under the reference's naming convention."""
import importlib.util
import os
import sys
from pathlib import Path

import pytest
import torch

if importlib.util.find_spec("gguf") is None:
    sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat"))
from gguf import GGMLQuantizationType, GGUFReader, ReaderTensor, dequantize  # noqa: E402

HIDDEN_SIZES = [256, 1024]


@pytest.fixture(scope="module")
def samples(tmp_path_factory):
    from ggq import gguf_io
    d = tmp_path_factory.mktemp("samples")
    for t in GGMLQuantizationType:
        if t.name in ("Q8_1", "Q8_K"):   # activation formats: never a weight file
            continue
        for h in HIDDEN_SIZES:
            try:
                gguf_io.write_sample_file(d, t, h, seed=h + int(t), d_scale=2.0 ** -4 if int(t) >= 10 else 1.0, row_multiples=(1, 2))
            except (ValueError, KeyError):
                pass
    return d


def get_gguf_sample_tensors(sample_dir, hidden_size, quant_type):
    """tests/test_dequantize.py:15-22"""
    sample_file = Path(sample_dir) / f"Quant_{quant_type.name}_{hidden_size}.gguf"
    tensors = GGUFReader(sample_file).tensors
    assert tensors and all(isinstance(t, ReaderTensor) for t in tensors)
    return tensors


@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("quant_type", [GGMLQuantizationType.Q4_0, GGMLQuantizationType.Q5_0, GGMLQuantizationType.Q8_0,
                                        GGMLQuantizationType.Q4_1, GGMLQuantizationType.Q5_1], ids=lambda t: t.name)
@torch.inference_mode()
def test_dequantize_cpu_op(samples, hidden_size, quant_type):
    """tests/test_dequantize.py:58-75 (its QUANT_TYPES are Q4_0, Q5_0, Q8_0; the CPU op also has Q4_1 / Q5_1)"""
    import custom_ops as ops
    for tensor in get_gguf_sample_tensors(samples, hidden_size, quant_type):
        shape = list(map(int, tensor.name.split("_")[-1].split("x")))
        ref_output = torch.tensor(dequantize(tensor.data, quant_type)).to(torch.float)
        output = ops.ggml_dequantize(torch.tensor(tensor.data, device="cpu"), quant_type, *shape).to(torch.float)
        torch.testing.assert_close(output, ref_output, atol=1e-2, rtol=4e-2)


GPU_QUANT_TYPES = [t for t in GGMLQuantizationType if t.name not in ("Q8_1", "Q8_K")]
MMQ_TYPES = [t for t in GPU_QUANT_TYPES if not t.name.startswith("IQ")]


@pytest.mark.gpu
@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float32], ids=str)
@pytest.mark.parametrize("quant_type", GPU_QUANT_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_dequantize_and_mmvq(samples, hidden_size, dtype, quant_type):
    """test_cuda_kernels.py:42-80"""
    import ggml as ops
    torch.manual_seed(0)
    x = torch.rand((1, hidden_size), dtype=dtype, device="cuda")
    for tensor in get_gguf_sample_tensors(samples, hidden_size, quant_type):
        shape = list(map(int, tensor.name.split("_")[-1].split("x")))
        weight = torch.tensor(dequantize(tensor.data, quant_type), device="cuda").to(dtype)
        qweight = torch.tensor(tensor.data, device="cuda")
        output = ops.ggml_dequantize(qweight, quant_type, *shape).to(dtype)
        torch.testing.assert_close(output, weight, atol=1e-2, rtol=4e-2)
        output = ops.ggml_mul_mat_vec_a8(qweight, x, quant_type, qweight.shape[0]).to(dtype)
        torch.testing.assert_close(output, x @ weight.T, atol=1, rtol=1e-1)


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens", [7, 83, 128, 2048])
@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float32], ids=str)
@pytest.mark.parametrize("quant_type", MMQ_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_mmq(samples, num_tokens, hidden_size, dtype, quant_type):
    """test_cuda_kernels.py:83-129"""
    import ggml as ops
    torch.manual_seed(0)
    x = torch.rand((num_tokens, hidden_size), dtype=dtype, device="cuda")
    for tensor in get_gguf_sample_tensors(samples, hidden_size, quant_type):
        weight = torch.tensor(dequantize(tensor.data, quant_type), device="cuda").to(dtype)
        ref_output = x @ weight.T
        qweight = torch.tensor(tensor.data, device="cuda")
        output = ops.ggml_mul_mat_a8(qweight, x, quant_type, qweight.shape[0]).to(dtype)
        atols = {torch.half: 1, torch.bfloat16: 1.5, torch.float: 1.2}
        rtols = {torch.half: 1e-1, torch.bfloat16: 1e4, torch.float: 2e1}
        torch.testing.assert_close(output, ref_output, atol=atols[dtype], rtol=rtols[dtype])
