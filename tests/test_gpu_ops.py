"""The drop-in operator surface on the GPU: written to read like the reference's own
tests (HK/tests/kernels/test_cuda_kernels.py) — same calls, same tolerances — with the
GGUF sample files written locally (synthetic block-valid tensors, ggq.gguf_io) and read back
with our GGUF reader, and gguf.dequantize replaced by its numpy restatement (oracle/ggq_numpy.py)."""
import numpy as np
import pytest
import torch

from ggq import synth
from ggq.formats import GGMLType, WEIGHT_TYPES, IQ_TYPES
from oracle import ggq_numpy as N

pytestmark = pytest.mark.gpu

DTYPES = [torch.half, torch.bfloat16, torch.float32]
HIDDEN_SIZES = [256, 1024]
NUM_TOKENS = [7, 83, 128, 2048]
QUANT_TYPES = WEIGHT_TYPES  # the reference lists Q2_K..Q6_K, Q4_0, Q5_0, Q8_0; Q4_1/Q5_1 added
# dequantise / MMVQ also take the IQ formats built so far (reference: test_cuda_kernels.py:18-25 lists all nine)
VEC_QUANT_TYPES = WEIGHT_TYPES + IQ_TYPES


_SAMPLE_DIR = None


def get_gguf_sample_tensors(hidden_size, quant_type):
    """benchmarks/utils.py:25-31 / HK/tests/utils.py:25-31 with the HF-hub download replaced by a locally
    written `Quant_{TYPE}_{hidden}.gguf` (ggq.gguf_io.write_sample_file) and gguf.GGUFReader by ours.
    The reference's absolute tolerances (atol = 1) are tuned to its sample checkpoints, so the K-quant
    block scales are shrunk until |w| <~ 1 like a real tensor (the default recipe reaches |w| ~ 16)."""
    global _SAMPLE_DIR
    import os, tempfile
    from ggq import gguf_io
    if _SAMPLE_DIR is None:
        _SAMPLE_DIR = tempfile.mkdtemp(prefix="ggq_gguf_samples_")
    path = os.path.join(_SAMPLE_DIR, gguf_io.sample_filename(quant_type, hidden_size))
    if not os.path.exists(path):
        d_scale = 2.0 ** -4 if int(quant_type) >= 10 else 1.0
        gguf_io.write_sample_file(_SAMPLE_DIR, quant_type, hidden_size, seed=hidden_size, d_scale=d_scale,
                                  row_multiples=(1,))
    return gguf_io.GGUFReader(path).tensors


def sample_tensors(hidden_size, quant_type):
    """(rows, uint8 [rows, row_bytes]) per tensor of the sample file, plus one ragged 96-row tensor"""
    out = [(t.data.shape[0], np.array(t.data)) for t in get_gguf_sample_tensors(hidden_size, quant_type)]
    d_scale = 2.0 ** -4 if int(quant_type) >= 10 else 1.0
    out.append((96, synth.random_weight(quant_type, 96, hidden_size, seed=96, d_scale=d_scale)))
    return out


@pytest.fixture(scope="module")
def ops():
    import ggml
    return ggml


@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("quant_type", VEC_QUANT_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_dequantize(ops, hidden_size, dtype, quant_type):
    for rows, data in sample_tensors(hidden_size, quant_type):
        ref_output = torch.tensor(N.gguf_dequantize(data, quant_type).reshape(rows, hidden_size), device="cuda").to(dtype)
        output = ops.ggml_dequantize(torch.tensor(data, device="cuda"), quant_type, rows, hidden_size).to(dtype)
        torch.testing.assert_close(output, ref_output, atol=1e-2, rtol=4e-2)


@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("quant_type", VEC_QUANT_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_mmvq(ops, hidden_size, dtype, quant_type):
    torch.manual_seed(0)
    x = torch.rand((1, hidden_size), dtype=dtype, device="cuda")
    for rows, data in sample_tensors(hidden_size, quant_type):
        weight = torch.tensor(N.gguf_dequantize(data, quant_type).reshape(rows, hidden_size), device="cuda").to(dtype)
        ref_output = x @ weight.T
        qweight = torch.tensor(data, device="cuda")
        output = ops.ggml_mul_mat_vec_a8(qweight, x, quant_type, qweight.shape[0]).to(dtype)
        torch.testing.assert_close(output, ref_output, atol=1, rtol=1e-1)


@pytest.mark.parametrize("num_tokens", NUM_TOKENS)
@pytest.mark.parametrize("hidden_size", HIDDEN_SIZES)
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("quant_type", QUANT_TYPES, ids=lambda t: t.name)
@torch.inference_mode()
def test_mmq(ops, num_tokens, hidden_size, dtype, quant_type):
    torch.manual_seed(0)
    x = torch.rand((num_tokens, hidden_size), dtype=dtype, device="cuda")
    for rows, data in sample_tensors(hidden_size, quant_type):
        weight = torch.tensor(N.gguf_dequantize(data, quant_type).reshape(rows, hidden_size), device="cuda").to(dtype)
        ref_output = x @ weight.T
        qweight = torch.tensor(data, device="cuda")
        output = ops.ggml_mul_mat_a8(qweight, x, quant_type, qweight.shape[0]).to(dtype)
        atols = {torch.half: 1, torch.bfloat16: 1.5, torch.float: 1.2}
        rtols = {torch.half: 1e-1, torch.bfloat16: 1e4, torch.float: 2e1}
        torch.testing.assert_close(output, ref_output, atol=atols[dtype], rtol=rtols[dtype])


@pytest.mark.parametrize("batch_size", [2, 8])
@pytest.mark.parametrize("num_tokens", [7, 128])
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("quant_type", [GGMLType.Q4_K, GGMLType.Q6_K, GGMLType.Q4_0, GGMLType.Q8_0], ids=lambda t: t.name)
@torch.inference_mode()
def test_mmq_batching(ops, batch_size, num_tokens, dtype, quant_type):
    hidden_size = 256
    torch.manual_seed(0)
    x = torch.rand((batch_size, num_tokens, hidden_size), dtype=dtype, device="cuda")
    for rows, data in sample_tensors(hidden_size, quant_type):
        weight = torch.tensor(N.gguf_dequantize(data, quant_type).reshape(rows, hidden_size), device="cuda").to(dtype)
        ref_output = x @ weight.T
        qweight = torch.tensor(data, device="cuda")
        output = ops.ggml_mul_mat_a8(qweight, x, quant_type, qweight.shape[0]).to(dtype)
        assert output.shape == (batch_size, num_tokens, rows)
        atols = {torch.half: 1, torch.bfloat16: 2, torch.float: 1}
        rtols = {torch.half: 1e-1, torch.bfloat16: 1e-1, torch.float: 2e-1}
        torch.testing.assert_close(output, ref_output, atol=atols[dtype], rtol=rtols[dtype])
        # 3-D call == 2-D call on the flattened tokens, bit for bit
        flat = ops.ggml_mul_mat_a8(qweight, x.reshape(-1, hidden_size), quant_type, rows)
        assert torch.equal(flat.reshape(batch_size, num_tokens, rows), output)


def test_op_errors(ops):
    w = torch.zeros((4, 18 * 8), dtype=torch.uint8, device="cuda")
    x = torch.zeros((1, 256), dtype=torch.float16, device="cuda")
    with pytest.raises(AssertionError):
        ops.ggml_mul_mat_vec_a8(w, torch.zeros((2, 256), dtype=torch.float16, device="cuda"), 2, 4)
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w, 5, 4, 256)  # unsupported type id
    with pytest.raises(RuntimeError):
        ops.ggml_mul_mat_a8(w, x.to(torch.float64), 2, 4)
    with pytest.raises(RuntimeError):
        ops.ggml_mul_mat_a8(w, torch.zeros((256,), dtype=torch.float16, device="cuda"), 2, 4)  # 1-D X
    with pytest.raises(RuntimeError):
        ops.ggml_mul_mat_a8(w[:, :100].contiguous(), x, 2, 4)  # wrong byte count
    assert ops.ggml_mul_mat_a8(w, x[:0], 2, 4).shape == (0, 4)
    # IQ4_NL shares Q4_0's 18-byte block: dequantise / MMVQ accept id 20, the GEMM op refuses it (the reference's
    # ggml_mul_mat_a8 silently returns uninitialised memory for it, HK/ggml/mmq.cu:222-251)
    assert ops.ggml_dequantize(w, 20, 4, 256).shape == (4, 256)
    assert ops.ggml_mul_mat_vec_a8(w, x, 20, 4).shape == (1, 4)
    with pytest.raises(RuntimeError):
        ops.ggml_mul_mat_a8(w, torch.zeros((2, 256), dtype=torch.float16, device="cuda"), 20, 4)
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w, 24, 4, 256)  # id 24 (GGML_TYPE_I8) is no quantised weight format: refused, not dispatched to a null pointer
    with pytest.raises(RuntimeError):
        ops.ggml_dequantize(w, 21, 4, 256)  # IQ3_S is built, but 4 x 18 bytes are not 4 rows of 110-byte super-blocks: byte-count check


@pytest.mark.parametrize("quant_type", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q6_K, GGMLType.Q4_0], ids=lambda t: t.name)
@pytest.mark.parametrize("num_tokens", [7, 40, 128])
@torch.inference_mode()
def test_shared_activation_quantisation(ops, quant_type, num_tokens):
    """ggq.linear: quantise X once, multiply by gate and up — bit-identical to two ggml_mul_mat_a8 calls whenever
    that op takes the streamed kernel (same two kernels, the first hoisted), within the op's tolerance otherwise"""
    from ggq import linear
    hidden, rows = 1024, 352
    torch.manual_seed(1)
    x = torch.randn((num_tokens, hidden), dtype=torch.half, device="cuda")
    wg = torch.tensor(synth.random_weight(quant_type, rows, hidden, seed=1), device="cuda")
    wu = torch.tensor(synth.random_weight(quant_type, rows, hidden, seed=2), device="cuda")
    yg, yu = linear.gate_up(x, wg, wu, quant_type, rows)
    rg = ops.ggml_mul_mat_a8(wg, x, quant_type, rows)
    ru = ops.ggml_mul_mat_a8(wu, x, quant_type, rows)
    streamed_from = {GGMLType.Q8_0: 65, GGMLType.Q6_K: 33, GGMLType.Q4_0: 9, GGMLType.Q4_1: 9, GGMLType.Q5_0: 9, GGMLType.Q5_1: 9,
                     GGMLType.Q4_K: 9}.get(quant_type, 5)   # ggq_mul_mat_q_ld's routing (mmq.hip)
    if num_tokens >= streamed_from:
        assert torch.equal(yg, rg) and torch.equal(yu, ru)
    else:
        torch.testing.assert_close(yg, rg, atol=2e-2, rtol=2e-3)
        torch.testing.assert_close(yu, ru, atol=2e-2, rtol=2e-3)
    from ggq.formats import NEED_SUM
    mine = quant_type in NEED_SUM
    other = GGMLType.Q8_0 if mine else GGMLType.Q4_K   # a format of the other scratch flavour
    with pytest.raises(ValueError):
        linear.QuantizedActivations(x, mine).matmul(wg, other, rows)


@pytest.mark.parametrize("quant_type", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q6_K, GGMLType.Q5_1], ids=lambda t: t.name)
@pytest.mark.parametrize("num_tokens", [3, 12, 40, 128])
@torch.inference_mode()
def test_fused_epilogues(ops, quant_type, num_tokens):
    """GGQ_EPI_BIAS / GGQ_EPI_SILU_MUL in the streamed kernel's write-back (ggq.linear): with fp32 in/out the bias
    form is acc + bias bit for bit, the gated form silu(gate) * acc within fp32 rounding of the exponential;
    with fp16 both are ONE rounding of that fp32 value (checked against the fp32 run)."""
    from ggq import linear
    from ggq.formats import NEED_SUM
    hidden, rows = 512, 200      # ragged last 32-row tile
    w = torch.tensor(synth.random_weight(quant_type, rows, hidden, seed=7, d_scale=2.0 ** -4 if int(quant_type) >= 10 else 1.0), device="cuda")
    w2 = torch.tensor(synth.random_weight(quant_type, rows, hidden, seed=8, d_scale=2.0 ** -4 if int(quant_type) >= 10 else 1.0), device="cuda")
    torch.manual_seed(3)
    x = torch.randn((num_tokens, hidden), device="cuda")
    bias = torch.randn(rows, device="cuda")
    qa = linear.QuantizedActivations(x, quant_type in NEED_SUM)
    acc = qa.matmul(w, quant_type, rows)                       # fp32 accumulators as written without an epilogue
    yb = linear._matmul_epi(qa, w, quant_type, rows, linear.EPI_BIAS, bias)
    assert torch.equal(yb, acc + bias)
    gate = qa.matmul(w2, quant_type, rows)
    yg = linear._matmul_epi(qa, w, quant_type, rows, linear.EPI_SILU_MUL, gate)
    want = acc * (gate / (1.0 + torch.exp(-gate)))
    torch.testing.assert_close(yg, want, rtol=2e-6, atol=1e-6 * float(want.abs().max()))
    # fp16: module level, against the fp32 path rounded once
    lin = linear.QuantLinear(w.cpu(), quant_type, hidden, rows, bias=bias.half().cpu()).cuda()
    y16 = lin(x.half())
    qa16 = linear.QuantizedActivations(x.half().float().contiguous(), quant_type in NEED_SUM)
    ref16 = (qa16.matmul(w, quant_type, rows) + bias.half().float())
    torch.testing.assert_close(y16.float(), ref16, rtol=2e-3, atol=2e-3 * float(ref16.abs().max()))


@torch.inference_mode()
def test_gated_ffn_module(ops):
    from ggq import linear
    hidden, inter, t = 512, 768, GGMLType.Q4_K
    ws = [torch.tensor(synth.random_weight(t, r, k, seed=s, d_scale=2.0 ** -6)) for r, k, s in ((inter, hidden, 1), (inter, hidden, 2), (hidden, inter, 3))]
    ffn = linear.QuantGatedFFN(ws[0], ws[1], ws[2], t, hidden, inter).cuda()
    x = torch.randn((2, 20, hidden), device="cuda", dtype=torch.float16)
    y = ffn(x)
    assert y.shape == (2, 20, hidden) and torch.isfinite(y).all()
    x2 = x.reshape(-1, hidden)
    g = ops.ggml_mul_mat_a8(ffn.w_gate, x2, t, inter).float()
    u = ops.ggml_mul_mat_a8(ffn.w_up, x2, t, inter).float()
    h = (torch.nn.functional.silu(g) * u).half()
    ref = ops.ggml_mul_mat_a8(ffn.down.weight, h, t, hidden).reshape(2, 20, hidden)
    torch.testing.assert_close(y.float(), ref.float(), rtol=2e-2, atol=2e-2 * float(ref.float().abs().max()))
