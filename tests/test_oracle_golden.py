"""CPU suite: the oracle against the golden vectors and against an independent derivation.

Pinned (bit-exact vs the reference's own compiled ggml-cpu op): fp32 dequantise of
Q4_0/Q4_1/Q5_0/Q5_1/Q8_0 — tests/golden/reference_cpu_dequant.npz.
Unpinned by any reference executable (CUDA-only paths): everything else; cross-checked here by
the independent numpy derivation (oracle/ggq_numpy.py) and by exact-arithmetic identities.
"""
import os

import numpy as np
import pytest

from ggq import synth
from ggq.formats import GGMLType, BLOCK, WEIGHT_TYPES, NEED_SUM, IQ_TYPES
from oracle import ggq_numpy as N

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LEGACY = [GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0]


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64), equal_nan=True)


@pytest.mark.parametrize("t", LEGACY, ids=lambda t: t.name)
def test_oracle_matches_reference_cpu_golden(oracle, t):
    g = np.load(os.path.join(GOLD, "reference_cpu_dequant.npz"))
    blocks, bits = g[f"{t.name}_blocks"], g[f"{t.name}_f32_bits"]
    y = oracle.dequantize_f32(blocks, t, blocks.shape[0] * 32)
    assert np.array_equal(y.view(np.uint32), bits), "oracle fp32 dequantise != reference ggml-cpu op (bit patterns)"


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q5_0, GGMLType.Q8_0], ids=lambda t: t.name)
def test_oracle_fp16_path_is_one_rounding_of_the_reference_fp32(oracle, t):
    """the oracle's fp16 (GPU-semantics) dequantise of Q4_0 / Q5_0 / Q8_0 == RN_fp16(reference fp32 golden):
    pins the fp16 path of these formats to reference-held output (same argument as the -m gpu twin in
    tests/test_gpu_parity.py::test_dequantize_vs_reference_golden)."""
    g = np.load(os.path.join(GOLD, "reference_cpu_dequant.npz"))
    blocks, bits = g[f"{t.name}_blocks"], g[f"{t.name}_f32_bits"]
    with np.errstate(over="ignore"):
        want = bits.view(np.float32).astype(np.float16)
    got = oracle.dequantize_f16(blocks, t, blocks.shape[0] * 32)
    an, bn = np.isnan(got), np.isnan(want)
    assert np.array_equal(an, bn) and np.array_equal(got.view(np.uint16)[~an], want.view(np.uint16)[~bn])


@pytest.mark.parametrize("t", LEGACY, ids=lambda t: t.name)
def test_oracle_matches_live_reference_when_present(oracle, t):
    ref = oracle.load_reference_cpu_op()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    import torch
    blocks = synth.random_blocks(t, 2048, seed=17)
    y = ref.ggml_dequantize(torch.from_numpy(blocks.reshape(1, -1).copy()), int(t), 1, 2048 * 32).numpy().reshape(-1)
    assert np.array_equal(oracle.dequantize_f32(blocks, t, 2048 * 32).view(np.uint32), y.view(np.uint32))


def test_reference_cpu_op_rejects_kquants_silently_oracle_flags_them(oracle):
    with pytest.raises(ValueError):
        oracle.dequantize_f32(synth.random_blocks(GGMLType.Q4_K, 1), GGMLType.Q4_K, 256)


def test_iq_mmvq_oracle_equals_dequantised_float64_product(oracle):
    """The IQ formats' MMVQ (vecdotq.cuh:607-888) is an exact integer dot times float scales: equal to
    dequant(W) · dequant_q8(x) in float64 up to fp32 rounding (IQ1_S uses the s8 = Σx shortcut for its delta term, like
    Q4_0's offset: bounded instead); the reference has no MMQ for them."""
    for t in IQ_TYPES:
        n_rows, k = 9, 1024
        w = synth.random_weight(t, n_rows, k, seed=12)
        x = np.random.default_rng(2).standard_normal((1, k)).astype(np.float32)
        W = N.dequantize_exact(w.reshape(-1, BLOCK[t][1]), t).reshape(n_rows, k)
        q8 = oracle.quantize_q8_1(x).reshape(1, -1, 36)[:, :k // 32]
        d8 = q8[:, :, 0:2].copy().view(np.float16)[..., 0].astype(np.float64)
        xq = (q8[:, :, 4:].view(np.int8).astype(np.float64) * d8[..., None]).reshape(1, k)
        yv, yabs = oracle.mul_mat_vec_q(w, x, t, n_rows)
        assert np.all(np.abs(yv - (xq @ W.T)[0]) <= (3e-2 if t == GGMLType.IQ1_S else 2e-5) * yabs + 1e-6), t.name
        with pytest.raises(ValueError):
            oracle.mul_mat_q(w, np.repeat(x, 4, axis=0), t, n_rows)


@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
def test_oracle_vs_independent_numpy(oracle, t):
    qk, _ = BLOCK[t]
    blocks = np.concatenate([synth.random_blocks(t, 500, seed=3), synth.edge_blocks(t)])
    k = blocks.shape[0] * qk
    assert _eq_nan(oracle.dequantize_f64(blocks, t, k), N.dequantize_exact(blocks, t).reshape(-1))
    assert _eq_nan(oracle.dequantize_f16(blocks, t, k).astype(np.float32),
                   N.dequantize_f16(blocks, t).reshape(-1).astype(np.float32))


@pytest.mark.parametrize("t", WEIGHT_TYPES, ids=lambda t: t.name)
def test_oracle_pins(oracle, t):
    g = np.load(os.path.join(GOLD, "oracle_pins.npz"))
    blocks = g[f"{t.name}_blocks"]
    qk, _ = BLOCK[t]
    f16 = oracle.dequantize_f16(blocks, t, blocks.shape[0] * qk)
    a, b = f16.view(np.uint16), g[f"{t.name}_f16_bits"]
    nan = np.isnan(f16)
    assert np.array_equal(a[~nan], b[~nan]) and np.array_equal(nan, np.isnan(b.view(np.float16)))
    w, x = g[f"{t.name}_mm_w"], g["mm_x"]
    yv, _ = oracle.mul_mat_vec_q(w, x[:1], t, 12)
    ym, _ = oracle.mul_mat_q(w, x, t, 12)
    assert np.array_equal(yv, g[f"{t.name}_mmvq_y"]) and np.array_equal(ym, g[f"{t.name}_mmq_y"])


@pytest.mark.parametrize("t", IQ_TYPES, ids=lambda t: t.name)
def test_oracle_pins_iq(oracle, t):
    g = np.load(os.path.join(GOLD, "oracle_pins_iq.npz"))
    blocks = g[f"{t.name}_blocks"]
    f16 = oracle.dequantize_f16(blocks, t, blocks.shape[0] * BLOCK[t][0])
    a, b = f16.view(np.uint16), g[f"{t.name}_f16_bits"]
    nan = np.isnan(f16)
    assert np.array_equal(a[~nan], b[~nan]) and np.array_equal(nan, np.isnan(b.view(np.float16)))
    yv, _ = oracle.mul_mat_vec_q(g[f"{t.name}_mm_w"], g["mm_x"], t, 12)
    assert np.array_equal(yv, g[f"{t.name}_mmvq_y"])


def test_oracle_quantizer_pins_and_numpy(oracle):
    g = np.load(os.path.join(GOLD, "oracle_pins.npz"))
    x = g["q8_x"]
    q = oracle.quantize_q8_1(x)
    assert np.array_equal(q, g["q8_1_bytes"])
    assert np.array_equal(oracle.quantize_q8_1_mmq(x, GGMLType.Q4_K), g["q8_1_mmq_sum_bytes"])
    assert np.array_equal(oracle.quantize_q8_1_mmq(x, GGMLType.Q8_0), g["q8_1_mmq_nosum_bytes"])
    # independent numpy restatement of the 32-lane butterfly / amax / roundf
    padded = 512
    xp = np.zeros((x.shape[0], padded), np.float32)
    xp[:, :x.shape[1]] = x
    qi, d, s = N.quantize_q8_1_groups(xp.reshape(x.shape[0], padded // 32, 32))
    blk = q.reshape(x.shape[0], padded // 32, 36)
    assert np.array_equal(blk[:, :, 4:].view(np.int8), qi)
    assert np.array_equal(blk[:, :, 0:2].copy().view(np.float16)[..., 0], d.astype(np.float16))
    assert np.array_equal(blk[:, :, 2:4].copy().view(np.float16)[..., 0], s.astype(np.float16))


@pytest.mark.parametrize("t", WEIGHT_TYPES, ids=lambda t: t.name)
def test_integer_unpack_ranges(t):
    q = N.unpack_ints(synth.random_blocks(t, 400, seed=5), t)
    lo, hi = {GGMLType.Q4_0: (0, 15), GGMLType.Q4_1: (0, 15), GGMLType.Q5_0: (0, 31), GGMLType.Q5_1: (0, 31),
              GGMLType.Q8_0: (-128, 127), GGMLType.Q2_K: (0, 3), GGMLType.Q3_K: (-4, 3), GGMLType.Q4_K: (0, 15),
              GGMLType.Q5_K: (0, 31), GGMLType.Q6_K: (-32, 31)}[t]
    assert q.min() >= lo and q.max() <= hi and q.min() == lo and q.max() == hi


@pytest.mark.parametrize("t", WEIGHT_TYPES + IQ_TYPES, ids=lambda t: t.name)
def test_fp16_dequant_within_reference_test_tolerance_of_gguf(oracle, t):
    """what the reference's own tests check (atol=1e-2, rtol=4e-2 vs gguf.dequantize) — on |w|<~1 data"""
    qk, _ = BLOCK[t]
    blocks = synth.random_blocks(t, 300, seed=9, d_scale=2.0 ** -4 if int(t) >= 10 else 1.0)
    got = oracle.dequantize_f16(blocks, t, 300 * qk).astype(np.float32)
    ref = N.gguf_dequantize(blocks, t).reshape(-1)
    assert np.all(np.abs(got - ref) <= 1e-2 + 4e-2 * np.abs(ref))


@pytest.mark.parametrize("t", WEIGHT_TYPES, ids=lambda t: t.name)
def test_matmul_oracles_agree_with_dequantised_float64_product(oracle, t):
    """MMQ (exact-integer canon) == dequant(W) . dequant_q8(X) in float64 up to fp32 rounding, for the
    formats without the s8 / d8*sum shortcut; for the others the shortcut error bound is checked."""
    n_rows, k, batch = 9, 1024, 6
    w = synth.random_weight(t, n_rows, k, seed=12)
    x = np.random.default_rng(2).standard_normal((batch, k)).astype(np.float32)
    W = N.dequantize_exact(w.reshape(-1, BLOCK[t][1]), t).reshape(n_rows, k)
    q8 = oracle.quantize_q8_1(x).reshape(batch, -1, 36)[:, :k // 32]
    d8 = q8[:, :, 0:2].copy().view(np.float16)[..., 0].astype(np.float64)
    xq = q8[:, :, 4:].view(np.int8).astype(np.float64) * d8[..., None]
    ref = xq.reshape(batch, k) @ W.T
    ym, yabs = oracle.mul_mat_q(w, x, t, n_rows)
    yv, yabsv = oracle.mul_mat_vec_q(w, x[:1], t, n_rows)
    exact_mmq = t in (GGMLType.Q4_0, GGMLType.Q5_0, GGMLType.Q8_0, GGMLType.Q2_K, GGMLType.Q3_K, GGMLType.Q6_K)
    tol_mmq = (2e-3 if not exact_mmq else 2e-5) * yabs + 1e-6   # fp16 d8/s8 + s8-vs-sum(q8) shortcut
    # MMQ uses fp32 d8 for non-need_sum formats, the identity above used fp16 d8: allow the 2^-11 scale error
    if exact_mmq and t not in NEED_SUM:
        tol_mmq = 1.5e-3 * yabs + 1e-6
    assert np.all(np.abs(ym - ref) <= tol_mmq)
    exact_mmvq = t in (GGMLType.Q8_0, GGMLType.Q2_K, GGMLType.Q3_K, GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q6_K)
    tol_v = (2e-5 if exact_mmvq else 3e-2) * yabsv + 1e-6        # 8*s8 offset shortcut of Q4_0/Q5_0, fp16 products
    assert np.all(np.abs(yv - ref[0]) <= tol_v)


def test_oracle_fp16_conversion_exhaustive(oracle):
    L = oracle.lib()
    hs = np.arange(65536, dtype=np.uint16)
    f = hs.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 257):  # sample + the full table through float16 round trip below
        assert np.float32(L.oracle_f16_to_f32(int(h))).tobytes() == f[h].tobytes() or np.isnan(f[h])
    xs = np.random.default_rng(0).standard_normal(20000).astype(np.float32) * np.float32(10.0) ** np.random.default_rng(1).integers(-9, 6, 20000).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.oracle_f32_to_f16(float(v)) for v in xs], np.uint16)
    assert np.array_equal(got, want)
