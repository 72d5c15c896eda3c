"""GPU parity tests proper: the hand-written HIP path, called through the C ABI
(include/ggq.h), against the oracle on the same seeded inputs.

  integer / byte work (dequantise bit patterns, Q8_1 bytes)  -> bit-exact
  fp accumulate (MMVQ / MMQ)                                  -> 1e-3 relative (north_star)
"""
import ctypes

import numpy as np
import pytest
import torch

from ggq import synth
from ggq import lib as ggqlib
from ggq.formats import GGMLType, BLOCK, WEIGHT_TYPES, NEED_SUM, IQ_TYPES, row_bytes
import util

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16, torch.float32]


def _x(shape, dtype, seed=0, kind="randn"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(shape, generator=g) if kind == "randn" else torch.rand(shape, generator=g)
    return x.to(dtype).cuda()


# ---------------------------------------------------------------- dequantise
@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
def test_dequantize_bit_exact(oracle, t):
    qk, bs = BLOCK[t]
    blocks = np.concatenate([synth.random_blocks(t, 777, seed=11), synth.edge_blocks(t)])
    nb = blocks.shape[0]
    got = util.gpu_dequant(blocks, t, 1, nb * qk).reshape(-1)
    ref = oracle.dequantize_f16(blocks, t, nb * qk)
    assert util.same_nan(got, ref), f"{t.name}: fp16 bit patterns differ from the oracle"


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q5_0, GGMLType.Q8_0], ids=lambda t: t.name)
def test_dequantize_vs_reference_golden(t):
    """The HIP kernel against output the REFERENCE itself produced (no oracle in between): the blocks of
    tests/golden/reference_cpu_dequant.npz were decoded by the reference's compiled ggml-cpu op
    (ggml-cpu/ggml-quants.hpp:4-112, fp32).  For these three formats the GPU result `__hmul(d, q)`
    (HK/ggml/dequantize.cuh:3-78) is one IEEE rounding of the exact product d·q, and so is fp16(reference fp32):
    the fp32 product of an 11-bit and an at most 8-bit significand is exact.  (Q4_1 / Q5_1 round twice on the GPU,
    hmul then hadd, and are compared with the oracle only.)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_cpu_dequant.npz"))
    blocks, bits = g[f"{t.name}_blocks"], g[f"{t.name}_f32_bits"]
    with np.errstate(over="ignore"):
        want = bits.view(np.float32).astype(np.float16)   # round-to-nearest-even; overflow -> inf, NaN stays NaN
    got = util.gpu_dequant(blocks, t, 1, blocks.shape[0] * 32).reshape(-1)
    assert util.same_nan(got, want), f"{t.name}: HIP fp16 != RN_fp16(reference ggml-cpu fp32)"


@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
def test_dequantize_ragged_and_tiny(oracle, t):
    qk, _ = BLOCK[t]
    for nb in (1, 2, 3, 65):  # fewer chunks than one workgroup / odd counts
        blocks = synth.random_blocks(t, nb, seed=nb)
        got = util.gpu_dequant(blocks, t, nb, qk).reshape(-1)
        assert util.same_nan(got, oracle.dequantize_f16(blocks, t, nb * qk))


def test_dequantize_empty_and_errors():
    from ggq import lib
    L = lib.hip()
    assert L.ggq_dequantize_f16(None, None, 2, 0, 0, None) == 0
    assert L.ggq_dequantize_f16(None, None, 5, 1, 32, None) == -1   # unsupported type
    assert L.ggq_dequantize_f16(None, None, 12, 1, 32, None) == -2  # 32 % 256 != 0
    assert L.ggq_dequantize_f16(None, None, 2, 1, 32, None) == -4   # null pointers


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q8_0, GGMLType.Q4_K, GGMLType.Q3_K, GGMLType.Q6_K, GGMLType.IQ4_XS],
                         ids=lambda t: t.name)
def test_dequantize_full_size(oracle, t):
    """BASELINE config 2 shape: 11008 x 4096 (45 M elements), checked bit-exact in full."""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=5)
    got = util.gpu_dequant(w, t, n_rows, k)
    ref = oracle.dequantize_f16(w, t, n_rows * k).reshape(n_rows, k)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))


# ---------------------------------------------------------------- Q8_1 quantiser
def _edge_rows(k, dtype):
    x = torch.zeros((6, k), dtype=torch.float32)
    x[1, 5] = 3.0                      # single spike
    x[2, :] = 1.0                      # constant
    x[3, :32] = torch.arange(32) - 15.5  # ties at .5 after scaling
    x[4, :] = torch.linspace(-1, 1, k)
    n5 = min(64, k)
    x[5, :n5] = torch.tensor([127.0, -127.0] * (n5 // 2))
    return x.to(dtype).cuda()


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("k", [32, 96, 1000, 4096])
def test_quantize_q8_1_bit_exact(oracle, dtype, k):
    for x in (_x((5, k), dtype, seed=k), _edge_rows(k, dtype)):
        got = util.gpu_quantize_q8_1(x)
        ref = oracle.quantize_q8_1(x.float().cpu().numpy())
        assert np.array_equal(got, ref), "block_q8_1 bytes differ from the oracle"


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0], ids=lambda t: t.name)  # need_sum / not
@pytest.mark.parametrize("batch,k", [(1, 256), (7, 1024), (130, 4096), (3, 992)])
def test_quantize_q8_1_mmq_bit_exact(oracle, dtype, t, batch, k):
    x = _x((batch, k), dtype, seed=batch + k)
    got = util.gpu_quantize_q8_1_mmq(x, t)
    ref = oracle.quantize_q8_1_mmq(x.float().cpu().numpy(), t)
    # (the scratch is sized for whole 32-token tiles; the reference layout uses its first batch*... bytes)
    assert np.array_equal(got[:ref.size], ref.reshape(-1)), "block_q8_1_mmq bytes differ from the oracle"
    assert not got[ref.size:].any(), "bytes beyond the reference layout were written"


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0], ids=lambda t: t.name)  # need_sum / not
@pytest.mark.parametrize("batch,k", [(1, 256), (31, 512), (32, 256), (33, 1024), (130, 4096), (70, 768)])
def test_quantize_q8_1_tiled_bit_exact(oracle, dtype, t, batch, k):
    """fragment-major scratch = the oracle's block_q8_1_mmq bytes, regrouped (tokens past the batch in the
    last 32-token tile are never written: the test buffer is zero-initialised, so they compare as zeros)"""
    x = _x((batch, k), dtype, seed=batch + k)
    got = util.gpu_quantize_q8_1_tiled(x, t)
    ref, n_tt = util.retile_q8_1_mmq(oracle.quantize_q8_1_mmq(x.float().cpu().numpy(), t), batch, k)
    assert np.array_equal(got[:ref.size].reshape(ref.shape), ref), "fragment-major bytes differ from the re-tiled oracle"


# ---------------------------------------------------------------- MMVQ
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("k,n_rows", [(256, 37), (1024, 64), (4096, 130)])
def test_mmvq_vs_oracle(oracle, dtype, t, k, n_rows):
    w = synth.random_weight(t, n_rows, k, seed=k + n_rows)
    x = _x((1, k), dtype, seed=1)
    y = util.gpu_mmvq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_vec_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref.reshape(1, -1), yabs.reshape(1, -1), dtype, f"mmvq {t.name}")


@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
def test_mmvq_more_rows_than_waves(oracle, t):
    """the fused GEMV has two row loops: K-split over the in-flight slots when the launch has at most one row per wave (every
    small case above), rows in flight otherwise — this is the otherwise, with a ragged last row group per wave"""
    k, n_rows = 512, 2 * 4096 + 5
    w = synth.random_weight(t, n_rows, k, seed=9)
    x = _x((1, k), torch.float16, seed=8)
    y = util.gpu_mmvq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_vec_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref.reshape(1, -1), yabs.reshape(1, -1), torch.float16, f"mmvq rows>waves {t.name}")


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q4_K, GGMLType.Q6_K, GGMLType.IQ3_S], ids=lambda t: t.name)
def test_mmvq_long_rows_few_of_them(oracle, t):
    """the down-projection shape (4096 x 11008: one row per wave, K-split): oracle on a sample of rows, bit-reproducible,
    and a row's result does not depend on where the row sits (the split is decided per launch, not per row group)"""
    n_rows, k = 4096, 11008 if BLOCK[t][0] == 32 else 11008 // 256 * 256
    w = synth.random_weight(t, n_rows, k, seed=41)
    x = _x((1, k), torch.float32, seed=42)
    y = util.gpu_mmvq(w, x, t, n_rows)
    rows = np.r_[0:40, 2000:2040, n_rows - 40:n_rows]
    ref, yabs = oracle.mul_mat_vec_q(w[rows], x.cpu().numpy(), t, len(rows))
    util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref.reshape(1, -1), yabs.reshape(1, -1), torch.float32, f"mmvq long {t.name}")
    assert torch.equal(util.gpu_mmvq(w, x, t, n_rows), y), "not reproducible run to run"
    perm = np.random.default_rng(5).permutation(n_rows)
    assert torch.equal(util.gpu_mmvq(np.ascontiguousarray(w[perm]), x, t, n_rows), y[:, torch.from_numpy(perm).cuda()]), "y(P W) != y(W) P"
    sub = 1500   # fewer rows than waves: some waves idle, same per-row arithmetic
    assert torch.equal(util.gpu_mmvq(np.ascontiguousarray(w[:sub]), x, t, sub), y[:, :sub])


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q5_1, GGMLType.Q8_0], ids=lambda t: t.name)
def test_mmvq_k_not_multiple_of_256(oracle, t):
    k, n_rows = 32 * 37, 19
    w = synth.random_weight(t, n_rows, k, seed=3)
    x = _x((1, k), torch.float32, seed=2)
    y = util.gpu_mmvq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_vec_q(w, x.cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref.reshape(1, -1), yabs.reshape(1, -1), torch.float32, f"mmvq {t.name}")


# ---------------------------------------------------------------- MMQ
@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", WEIGHT_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k,n_rows", [(1, 256, 33), (7, 1024, 64), (33, 256, 70), (83, 1024, 40),
                                            (128, 512, 96), (200, 256, 32)])
def test_mmq_vs_oracle(oracle, dtype, t, batch, k, n_rows):
    w = synth.random_weight(t, n_rows, k, seed=batch + k)
    x = _x((batch, k), dtype, seed=4)
    y = util.gpu_mmq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"mmq {t.name} b={batch}")


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q5_0, GGMLType.Q8_0, GGMLType.Q4_1], ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k,n_rows", [(9, 32 * 21, 45), (3, 32 * 21, 45), (70, 32 * 5, 33), (40, 32, 64), (33, 32 * 13, 31)])
def test_mmq_k_not_multiple_of_256(oracle, t, batch, k, n_rows):
    w = synth.random_weight(t, n_rows, k, seed=8)
    x = _x((batch, k), torch.float32, seed=6)
    y = util.gpu_mmq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, torch.float32, f"mmq {t.name}")


STREAM_TYPES = WEIGHT_TYPES


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", STREAM_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k,n_rows", [(5, 256, 33), (17, 768, 31), (32, 1024, 64), (33, 256, 1), (64, 2304, 70), (48, 1024, 100),
                                            (65, 1280, 95), (128, 4096, 160), (129, 512, 32), (300, 256, 257)])
def test_mmq_streamed_vs_oracle(oracle, dtype, t, batch, k, n_rows):
    """The streamed kernel (32- and 64-token units, ragged row / token tiles, 1..18 K stages per wave,
    K-slices of unequal length, fewer stages than K-slices) against the oracle."""
    assert ggqlib.hip().ggq_mmq_tiled_supported(int(t), k) == 1
    w = synth.random_weight(t, n_rows, k, seed=batch + k)
    x = _x((batch, k), dtype, seed=14)
    y = util.gpu_mmq_pretiled(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"streamed mmq {t.name} b={batch}")


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q5_K], ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [33, 128])
def test_mmq_streamed_min_scale_range(oracle, t, batch):
    """The per-super-block min term of the streamed kernel is one fp16 MFMA over an exact (hi, lo) split of dmin * m:
    cover the fp16 subnormal range (the remainder is subnormal for every realistic dmin), values next to the 1024
    threshold, and the scaled cold branch for |dmin * m| beyond the fp16 range (dmin up to 65504)."""
    k, n_rows = 1024, 96
    w = synth.random_weight(t, n_rows, k, seed=3).reshape(n_rows, k // 256, -1).copy()
    specials = np.array([6e-8, -6e-8, 3.0517578125e-05, 6.103515625e-05, -0.000244140625, 0.333251953125, 1.0, 1023.5, 1024.0,
                         -1024.0, 1025.0, 2048.0, -30000.0, 65504.0, 0.0, -0.0], dtype=np.float16)
    rng = np.random.default_rng(9)
    dmin = specials[rng.integers(0, len(specials), size=(n_rows, k // 256))]
    dmin[:32, :] = np.where(np.abs(dmin[:32, :].astype(np.float32)) > 1024, np.float16(0.0078125), dmin[:32, :])   # one tile stays on the hot path
    w[:, :, 2:4] = dmin.view(np.uint8).reshape(n_rows, k // 256, 2)
    w = w.reshape(n_rows, -1)
    x = _x((batch, k), torch.float32, seed=21)   # fp32 in / out: dmin = 65504 does not overflow the result
    y = util.gpu_mmq_pretiled(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.cpu().numpy(), t, n_rows)
    assert np.isfinite(ref).all()
    util.assert_fp_accumulate(y, ref, yabs, torch.float32, f"streamed mmq min range {t.name} b={batch}")
    # a row's result does not depend on which rows share its 32-row tile: the |dmin| > 1024 rows take the scaled pass alone
    perm = np.random.default_rng(4).permutation(n_rows)
    yp = util.gpu_mmq_pretiled(np.ascontiguousarray(w[perm]), x, t, n_rows)
    assert torch.equal(yp, y[:, torch.from_numpy(perm).cuda()]), "Y(P W) != Y(W) P with mixed |dmin| ranges in one tile"


def test_mmq_streamed_ldy_and_errors(oracle):
    """row pitch (the multi-GPU slab write) + the argument checks of the fragment-major entry points"""
    L = ggqlib.hip()
    t, batch, k, n_rows, ldy = GGMLType.Q4_K, 70, 512, 40, 104
    w = synth.random_weight(t, n_rows, k, seed=5)
    x = _x((batch, k), torch.float16, seed=15)
    y = util.gpu_mmq_pretiled(w, x, t, n_rows, ldy=ldy)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y[:, :n_rows], ref, yabs, torch.float16, "streamed mmq ldy")
    assert torch.count_nonzero(y[:, n_rows:]) == 0, "columns beyond n_rows were written"
    assert L.ggq_mmq_tiled_supported(int(GGMLType.Q4_0), 4096) == 1
    assert L.ggq_mmq_tiled_supported(int(GGMLType.Q4_K), 4096 + 32) == 0
    assert L.ggq_mmq_tiled_supported(1, 4096) == 0
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    wd = util.dev_bytes(w)
    st = util.stream_ptr()
    assert L.ggq_mul_mat_q_pretiled(util.vp(wd), util.vp(q), util.vp(y), 1, 1, batch, k, n_rows, ldy, st) == -1
    assert L.ggq_mul_mat_q_pretiled(util.vp(wd), util.vp(q), util.vp(y), int(t), 7, batch, k, n_rows, ldy, st) == -3
    assert L.ggq_mul_mat_q_pretiled(util.vp(wd), util.vp(q), util.vp(y), int(t), 1, batch, k, n_rows, n_rows - 1, st) == -4
    assert L.ggq_mul_mat_q_pretiled(ctypes.c_void_p(wd.data_ptr() + 1), util.vp(q), util.vp(y), int(t), 1, batch, k, n_rows, ldy, st) == -6
    assert L.ggq_mul_mat_q_pretiled(util.vp(wd), util.vp(q), util.vp(y), int(t), 1, 0, k, n_rows, ldy, st) == 0
    assert L.ggq_quantize_q8_1_tiled(util.vp(x), 1, ctypes.c_void_p(q.data_ptr() + 4), batch, k, int(t), st) == -6


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q6_K, GGMLType.Q8_0], ids=lambda t: t.name)
def test_mmq_weight_pointer_not_16_aligned(oracle, t):
    """GGUF tensors are 32-byte aligned, but the ABI only asks for 2: the 16-byte row chunks of the streamed
    kernel must work from any even address."""
    L = ggqlib.hip()
    batch, k, n_rows = 40, 512, 48
    w = synth.random_weight(t, n_rows, k, seed=6)
    x = _x((batch, k), torch.float32, seed=16)
    buf = torch.zeros(w.size + 64, dtype=torch.uint8, device="cuda")
    off = (-buf.data_ptr()) % 16 + 2
    buf[off:off + w.size] = torch.from_numpy(w.reshape(-1)).cuda()
    y = torch.empty((batch, n_rows), dtype=torch.float32, device="cuda")
    scratch = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_mul_mat_q(ctypes.c_void_p(buf.data_ptr() + off), util.vp(x), util.vp(y), int(t), 0, batch, k,
                                 n_rows, util.vp(scratch), util.stream_ptr()), "ggq_mul_mat_q")
    torch.cuda.synchronize()
    ref, yabs = oracle.mul_mat_q(w, x.cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, torch.float32, f"mmq unaligned w {t.name}")


def test_mmq_integer_exact(oracle):
    """Integer unpack + int8 MFMA contraction, bit-exact: with power-of-two scales and
    integer-valued activations every product and partial sum is exactly representable,
    so the GPU result must equal the oracle to the last bit."""
    for t in WEIGHT_TYPES:
        qk, bs = BLOCK[t]
        n_rows, k, batch = 64, 512, 40
        w = synth.random_weight(t, n_rows, k, seed=9).reshape(-1, bs)
        from ggq.synth import _F16_FIELDS
        d_off, m_off = _F16_FIELDS[t]
        w[:, d_off:d_off + 2] = np.array([2.0 ** -4], np.float16).view(np.uint8)
        if m_off is not None:
            w[:, m_off:m_off + 2] = np.array([2.0 ** -3], np.float16).view(np.uint8)
        rng = np.random.default_rng(1)
        xi = rng.integers(-8, 9, size=(batch, k)).astype(np.float32)
        xi[:, ::32] = 127.0  # amax = 127 in every group -> d8 = 1 exactly, q8 = x
        x = torch.from_numpy(xi).cuda()
        y = util.gpu_mmq(w.reshape(n_rows, -1), x, t, n_rows).cpu().numpy()
        ref, _ = oracle.mul_mat_q(w.reshape(n_rows, -1), xi, t, n_rows)
        assert np.array_equal(y, ref), f"{t.name}: exact-integer MMQ differs"


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q6_K, GGMLType.Q8_0], ids=lambda t: t.name)
def test_mmq_full_size_rows_sample(oracle, t):
    """BASELINE config 4 shape (K=4096, N=11008, batch 128): oracle on a sample of rows +
    the row-permutation property (permuting W's rows permutes Y's columns bit-exactly)."""
    n_rows, k, batch = 11008, 4096, 128
    w = synth.random_weight(t, n_rows, k, seed=21)
    x = _x((batch, k), torch.float16, seed=22)
    y = util.gpu_mmq(w, x, t, n_rows)
    rows = np.r_[0:40, 5000:5040, n_rows - 40:n_rows]
    ref, yabs = oracle.mul_mat_q(w[rows], x.float().cpu().numpy(), t, len(rows))
    util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref, yabs, torch.float16, f"mmq full {t.name}")
    perm = np.random.default_rng(0).permutation(n_rows)
    y2 = util.gpu_mmq(w[perm], x, t, n_rows)
    assert torch.equal(y2, y[:, torch.from_numpy(perm).cuda()])


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q4_1, GGMLType.Q8_0, GGMLType.Q2_K, GGMLType.Q4_0], ids=lambda t: t.name)
@pytest.mark.parametrize("n_rows", [4096, 4128, 8224, 11008, 12320])
def test_mmq_op_across_the_shape_terms_of_the_routing(oracle, t, n_rows):
    """ggq_mul_mat_q on both sides of every row-count threshold of ggq_mmq_route / ggq_mmq_stream_unit_tokens (4096 rows; the
    8192 < rows <= 12288 band; Q8_0's and Q2_K's own terms), at the batches where they switch kernels or unit sizes: oracle on a
    sample of rows (first / middle / last tiles), whatever kernel the table picks."""
    k = 512
    L = ggqlib.hip()
    w = synth.random_weight(t, n_rows, k, seed=n_rows % 97)
    rows = np.r_[0:20, n_rows // 2 - 10:n_rows // 2 + 10, n_rows - 20:n_rows]
    seen = set()
    for batch in (3, 16, 17, 32, 33, 48, 64, 65):
        x = _x((batch, k), torch.float16, seed=batch)
        y = util.gpu_mmq(w, x, t, n_rows)
        ref, yabs = oracle.mul_mat_q(w[rows], x.float().cpu().numpy(), t, len(rows))
        util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref, yabs, torch.float16, f"mmq {t.name} b={batch} rows={n_rows}")
        seen.add((L.ggq_mmq_route(int(t), batch, k, n_rows), L.ggq_mmq_stream_unit_tokens(int(t), batch, n_rows)))
    assert len(seen) >= 2   # several (kernel, unit size) pairs were exercised (Q2_K: two kernels, 32-token units only)


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q5_1], ids=lambda t: t.name)
def test_mmq_transposed_shape_rows_sample(oracle, t):
    """SURVEY §8 secondary shape: K = 11008 (11008 % 512 = 256 exercises the scratch padding; 43 / 86 K stages
    split unevenly over the four K-slices), N = 4096, batch 128 and a ragged 77."""
    n_rows, k = 4096, 11008
    w = synth.random_weight(t, n_rows, k, seed=31)
    rows = np.r_[0:24, 2000:2024, n_rows - 24:n_rows]
    for batch in (128, 77):
        x = _x((batch, k), torch.float16, seed=32)
        y = util.gpu_mmq(w, x, t, n_rows)
        ref, yabs = oracle.mul_mat_q(w[rows], x.float().cpu().numpy(), t, len(rows))
        util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref, yabs, torch.float16, f"mmq transposed {t.name} b={batch}")


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q6_K, GGMLType.Q4_0, GGMLType.Q5_K], ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [1, 8, 128])
def test_power_of_two_scaling_full_size(t, batch):
    """Size-independent exactness property at the BASELINE shape (no oracle needed): scaling X by 2^k scales every
    Q8_1 block scale and block sum by exactly 2^k and leaves the int8 codes unchanged, so Y scales by exactly 2^k
    (fp32 in/out; all values stay far from fp16's range limits for k = +-3).  Covers the GEMV, small-batch and
    64-token-unit launches on all 11008 x 4096 outputs.  (Not a property of Q4_1 / Q5_1: the reference multiplies their
    scales in fp16, `__hmul2(dm, ds8)`, and those products sit at the edge of fp16's subnormal range.)"""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=51)
    x = _x((batch, k), torch.float32, seed=52)
    f = util.gpu_mmvq if batch == 1 else util.gpu_mmq
    y = f(w, x, t, n_rows)
    assert torch.isfinite(y).all()
    for kk in (3, -3):
        ys = f(w, x * (2.0 ** kk), t, n_rows)
        assert torch.equal(ys, y * (2.0 ** kk)), f"{t.name} batch {batch}: Y(2^{kk} X) != 2^{kk} Y(X)"


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q8_0, GGMLType.Q5_0, GGMLType.Q6_K, GGMLType.Q4_1], ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [32, 128])
def test_mmq_run_to_run_bitwise_reproducible(t, batch):
    """Eight launches of the same full-size matmul give the same bits.  (Round 2 met two kernel variants — parked — whose
    results changed from run to run in the last lanes of single accumulator registers; every shipped kernel has a
    fixed summation order, so any difference here is a hazard, not rounding.)"""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=71)
    x = _x((batch, k), torch.float16, seed=72)
    y0 = util.gpu_mmq(w, x, t, n_rows)
    assert torch.isfinite(y0).all()
    for _ in range(7):
        assert torch.equal(util.gpu_mmq(w, x, t, n_rows), y0)


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q4_1, GGMLType.Q6_K, GGMLType.Q5_0], ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [8, 128])
def test_weight_row_permutation_full_size(t, batch):
    """Permuting the rows of W permutes the columns of Y bit-exactly: a weight row's result may not depend on its place
    in the 32-row tile (accumulator register / lane, and — since all four K-slice waves finish a unit, each one a
    register quarter — on which wave reduced and stored it), nor on the unit or CU it lands on.  Full shape, no oracle."""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=81)
    x = _x((batch, k), torch.float16, seed=82)
    y = util.gpu_mmq(w, x, t, n_rows)
    perm = np.random.default_rng(5).permutation(n_rows)
    yp = util.gpu_mmq(np.ascontiguousarray(w[perm]), x, t, n_rows)
    assert torch.equal(yp, y[:, torch.from_numpy(perm).cuda()]), f"{t.name} batch {batch}: Y(P W) != Y(W) P"


@pytest.mark.parametrize("t", [GGMLType.Q4_K, GGMLType.Q8_0, GGMLType.Q5_1], ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [8, 16, 77, 128])
def test_token_permutation_full_size(t, batch):
    """Permuting the rows of X permutes the rows of Y bit-exactly: a token's result may not depend on its lane,
    accumulator register (batch <= 16 variants), token block or 64-token unit.  Full BASELINE shape, no oracle."""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=61)
    x = _x((batch, k), torch.float16, seed=62)
    y = util.gpu_mmq(w, x, t, n_rows)
    perm = torch.from_numpy(np.random.default_rng(3).permutation(batch)).cuda()
    y2 = util.gpu_mmq(w, x[perm].contiguous(), t, n_rows)
    assert torch.equal(y2, y[perm]), f"{t.name} batch {batch}: Y(P X) != P Y(X)"


@pytest.mark.parametrize("batch", [1, 8, 128])
def test_config5_shard_shape(oracle, batch):
    """BASELINE configs[4]: Q4_K 8192 x 28672 sharded over 8 GPUs = 3584 rows x K 8192 per rank, batch 1 / 8 / 128
    (batch 1 through the GEMV op): oracle on a sample of rows; exercises the eight-K-slice launches (112 / 224 units)."""
    t, n_rows, k = GGMLType.Q4_K, 3584, 8192
    w = synth.random_weight(t, n_rows, k, seed=41)
    x = _x((batch, k), torch.float16, seed=42)
    rows = np.r_[0:20, 1790:1810, n_rows - 20:n_rows]
    sel = torch.from_numpy(rows).cuda()
    if batch == 1:
        y = util.gpu_mmvq(w, x, t, n_rows)
        ref, yabs = oracle.mul_mat_vec_q(w[rows], x.float().cpu().numpy(), t, len(rows))
        util.assert_fp_accumulate(y[:, sel], ref.reshape(1, -1), yabs.reshape(1, -1), torch.float16, "config5 mmvq")
    else:
        y = util.gpu_mmq(w, x, t, n_rows)
        ref, yabs = oracle.mul_mat_q(w[rows], x.float().cpu().numpy(), t, len(rows))
        util.assert_fp_accumulate(y[:, sel], ref, yabs, torch.float16, f"config5 mmq b={batch}")


@pytest.mark.parametrize("batch", [1, 8, 128])
def test_row_sharded_equals_unsharded_config5(batch):
    """BASELINE configs[4] at its real shape (Q4_K 28672 x 8192), the property the multi-GPU path rests on: the slab a rank computes
    from its row shard equals the same columns of the one-GPU result.  The kernel and its K-slice count are chosen from the whole
    (shape, batch) — 3584-row shards take eight K-slices / the GEMV K-split, the full matrix does not (include/ggq.h) — so the fp32
    partial sums are added in a different order: equal within the canon's tolerance (1e-3 relative to the result's scale, one fp16
    rounding), bit-for-bit only where both sides take the same route.  P = 2 and P = 8, first / middle / last shard.  No oracle."""
    t, n_rows, k = GGMLType.Q4_K, 28672, 8192
    w = synth.random_weight(t, n_rows, k, seed=71)
    x = _x((batch, k), torch.float16, seed=72)
    run = util.gpu_mmvq if batch == 1 else util.gpu_mmq
    full = run(w, x, t, n_rows).float()
    scale = float(full.abs().mean())
    for world in (2, 8):
        per = n_rows // world
        for rank in sorted({0, world // 2, world - 1}):
            slab = run(np.ascontiguousarray(w[rank * per:(rank + 1) * per]), x, t, per).float()
            ref = full[:, rank * per:(rank + 1) * per]
            err = float(((slab - ref).abs() / (ref.abs() + scale)).max())
            assert err <= 1e-3, f"P={world} rank {rank} batch {batch}: shard differs from the unsharded result by {err:.2e}"


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q4_K], ids=lambda t: t.name)
def test_mmvq_full_size(oracle, t):
    """BASELINE config 3 shape: batch 1, K=4096, N=11008."""
    n_rows, k = 11008, 4096
    w = synth.random_weight(t, n_rows, k, seed=31)
    x = _x((1, k), torch.float16, seed=32)
    y = util.gpu_mmvq(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_vec_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref.reshape(1, -1), yabs.reshape(1, -1), torch.float16, f"mmvq full {t.name}")
