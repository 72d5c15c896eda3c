"""The 16-token-tile quantised GEMM (mmq_t16.hip, what ggq_mul_mat_q runs for the HBM-bound batches): its activation
layout against the re-tiled oracle bytes (bit-exact), the kernel against the oracle (1e-3 relative, north_star), and the
size-independent exactness properties at the BASELINE shapes."""
import numpy as np
import pytest
import torch

from ggq import synth
from ggq import lib as ggqlib
from ggq.formats import GGMLType
import util

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16, torch.float32]
T16_TYPES = [t for t in (GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0,
                         GGMLType.Q6_K, GGMLType.Q2_K, GGMLType.Q3_K)
             if ggqlib.hip().ggq_mmq_t16_type_supported(int(t))] if torch.cuda.is_available() else []


def _x(shape, dtype, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(dtype).cuda()


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k", [(1, 256), (5, 512), (16, 256), (17, 1024), (33, 768), (64, 4096)])
def test_quantize_q8_1_t16_bit_exact(oracle, dtype, t, batch, k):
    x = _x((batch, k), dtype, seed=batch + k)
    got = util.gpu_quantize_q8_1_t16(x, t)
    ref = util.retile_q8_1_t16(oracle.quantize_q8_1_mmq(x.float().cpu().numpy(), t), batch, k, t)
    assert np.array_equal(got[:ref.size].reshape(ref.shape), ref), "16-token-tile bytes differ from the re-tiled oracle"


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k,n_rows", [(1, 256, 33), (5, 256, 16), (8, 1024, 64), (16, 4096, 130), (17, 768, 31), (32, 2304, 70),
                                            (33, 1280, 95), (64, 4096, 48), (9, 11008, 40), (24, 16384, 20), (7, 20480, 17)])
def test_mmq_t16_vs_oracle(oracle, dtype, t, batch, k, n_rows):
    """ragged row tiles and token tiles, 1 .. 12 K-slices of equal and unequal length, slices of several LDS rounds (K > 12288)"""
    from ggq.formats import BLOCK
    if (BLOCK[t][0] == 32 or t in (GGMLType.Q6_K, GGMLType.Q3_K)) and batch > 16:   # the 32-element-block formats and Q6_K have no two-token-tile instance
        assert ggqlib.hip().ggq_mmq_t16_supported(int(t), k, batch) == 0
        q = torch.zeros(int(ggqlib.hip().ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
        assert ggqlib.hip().ggq_mul_mat_q_t16(util.vp(q), util.vp(q), util.vp(q), int(t), 1, batch, k, n_rows, n_rows, 0, None, util.stream_ptr()) == -2
        return
    assert ggqlib.hip().ggq_mmq_t16_supported(int(t), k, batch) == 1
    w = synth.random_weight(t, n_rows, k, seed=batch + k)
    x = _x((batch, k), dtype, seed=14)
    y = util.gpu_mmq_t16(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"t16 mmq {t.name} b={batch}")


@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
def test_mmq_t16_integer_exact(oracle, t):
    """power-of-two scales + integer activations: every product and partial sum is exact, so the result must equal the
    oracle's to the last bit (integer unpack, operand masking and the int8 MFMA contraction)"""
    from ggq.formats import BLOCK
    from ggq.synth import _F16_FIELDS
    qk, bs = BLOCK[t]
    n_rows, k, batch = 64, 512, (24 if t in (GGMLType.Q4_K, GGMLType.Q5_K) else 14)
    w = synth.random_weight(t, n_rows, k, seed=9).reshape(-1, bs)
    d_off, m_off = _F16_FIELDS[t]
    w[:, d_off:d_off + 2] = np.array([2.0 ** -4], np.float16).view(np.uint8)
    if m_off is not None:
        w[:, m_off:m_off + 2] = np.array([2.0 ** -3], np.float16).view(np.uint8)
    rng = np.random.default_rng(1)
    xi = rng.integers(-8, 9, size=(batch, k)).astype(np.float32)
    xi[:, ::32] = 127.0
    y = util.gpu_mmq_t16(w.reshape(n_rows, -1), torch.from_numpy(xi).cuda(), t, n_rows).cpu().numpy()
    ref, _ = oracle.mul_mat_q(w.reshape(n_rows, -1), xi, t, n_rows)
    assert np.array_equal(y, ref), f"{t.name}: exact-integer t16 MMQ differs"


def test_mmq_t16_ldy_epilogues_unaligned_and_errors(oracle):
    L = ggqlib.hip()
    t, batch, k, n_rows, ldy = GGMLType.Q4_K, 20, 512, 40, 104
    w = synth.random_weight(t, n_rows, k, seed=5)
    x = _x((batch, k), torch.float16, seed=15)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    y = util.gpu_mmq_t16(w, x, t, n_rows, ldy=ldy)
    util.assert_fp_accumulate(y[:, :n_rows], ref, yabs, torch.float16, "t16 ldy")
    assert torch.count_nonzero(y[:, n_rows:]) == 0, "columns beyond n_rows were written"
    # weight pointer at an odd multiple of 2 (the ABI asks for 2-byte alignment only)
    buf = torch.zeros(w.size + 64, dtype=torch.uint8, device="cuda")
    off = (-buf.data_ptr()) % 16 + 2
    buf[off:off + w.size] = torch.from_numpy(w.reshape(-1)).cuda()
    y2 = util.gpu_mmq_t16(w, x, t, n_rows, w_dev=buf[off:])
    assert torch.equal(y2, y[:, :n_rows].contiguous())
    # epilogues on the fp32 accumulator
    bias = torch.randn(n_rows, generator=torch.Generator().manual_seed(2)).half().cuda()
    yb = util.gpu_mmq_t16(w, x, t, n_rows, epilogue=1, aux=bias)
    util.assert_fp_accumulate(yb, ref + bias.float().cpu().numpy()[None, :], yabs + np.abs(bias.float().cpu().numpy())[None, :], torch.float16, "t16 bias")
    gate = torch.randn((batch, n_rows), generator=torch.Generator().manual_seed(3)).half().cuda()
    yg = util.gpu_mmq_t16(w, x, t, n_rows, epilogue=2, aux=gate)
    gf = gate.float().cpu().numpy().astype(np.float64)
    sil = gf / (1.0 + np.exp(-gf))
    util.assert_fp_accumulate(yg, (ref * sil).astype(np.float32), (yabs * np.abs(sil)).astype(np.float32), torch.float16, "t16 silu_mul")
    # argument checks
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    wd, st = util.dev_bytes(w), util.stream_ptr()
    args = lambda **kw: [kw.get("w", util.vp(wd)), kw.get("q", util.vp(q)), util.vp(y), kw.get("t", int(t)), kw.get("dt", 1), kw.get("b", batch),
                         kw.get("k", k), n_rows, kw.get("ldy", ldy), kw.get("epi", 0), None, st]
    assert L.ggq_mul_mat_q_t16(*args(t=1)) == -1
    assert L.ggq_mul_mat_q_t16(*args(dt=7)) == -3
    assert L.ggq_mul_mat_q_t16(*args(ldy=n_rows - 1)) == -4
    assert L.ggq_mul_mat_q_t16(*args(epi=1)) == -4      # epilogue without aux
    assert L.ggq_mul_mat_q_t16(*args(b=0)) == 0
    import ctypes
    assert L.ggq_mul_mat_q_t16(*args(w=ctypes.c_void_p(wd.data_ptr() + 1))) == -6
    assert L.ggq_quantize_q8_1_t16(util.vp(x), 1, ctypes.c_void_p(q.data_ptr() + 4), batch, k, int(t), st) == -6
    assert L.ggq_mmq_t16_supported(int(t), 4096 + 32, 8) == 0
    assert L.ggq_mmq_t16_supported(1, 4096, 8) == 0


@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
def test_mmq_t16_random_shapes(oracle, t):
    """seeded sweep over ragged shapes: every combination of short / long K (one to several LDS rounds per wave), partial last
    row tile, partial token tile and both tile forms, against the oracle"""
    from ggq.formats import BLOCK
    rng = np.random.default_rng(1000 + int(t))
    bmax = 32 if t in (GGMLType.Q4_K, GGMLType.Q5_K) else 16
    for case in range(14):
        batch = int(rng.integers(1, bmax + 1))
        k = 256 * int(rng.choice([1, 2, 3, 5, 7, 12, 17, 24, 41, 64]))
        n_rows = int(rng.integers(5, 75))   # Q6_K / Q3_K need a tensor of at least 1 KB: 5 rows x 110 bytes x (k / 256) >= 1 KB from k = 512
        if t in (GGMLType.Q6_K, GGMLType.Q3_K) and n_rows * (k // 256) * BLOCK[t][1] < 1024:
            n_rows = 12
        w = synth.random_weight(t, n_rows, k, seed=case)
        x = _x((batch, k), torch.float32, seed=100 + case)
        y = util.gpu_mmq_t16(w, x, t, n_rows)
        ref, yabs = oracle.mul_mat_q(w, x.cpu().numpy(), t, n_rows)
        util.assert_fp_accumulate(y, ref, yabs, torch.float32, f"t16 random {t.name} b={batch} k={k} n={n_rows}")


@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch", [3, 12])
def test_mmq_t16_weights_at_any_2_byte_alignment(oracle, t, batch):
    """the ABI asks for 2-byte aligned weights: the LDS-DMA image (global_load_lds / buffer_load ... lds) and the unaligned LDS
    reads of the 18 .. 34-byte blocks and of Q6_K / Q3_K must give the same bits wherever the tensor starts"""
    k, n_rows = 1280, 37
    w = synth.random_weight(t, n_rows, k, seed=6)
    x = _x((batch, k), torch.float16, seed=16)
    y = util.gpu_mmq_t16(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, torch.float16, f"t16 {t.name}")
    for shift in (2, 6, 14):
        buf = torch.zeros(w.size + 64, dtype=torch.uint8, device="cuda")
        off = (-buf.data_ptr()) % 16 + shift
        buf[off:off + w.size] = torch.from_numpy(w.reshape(-1)).cuda()
        assert torch.equal(util.gpu_mmq_t16(w, x, t, n_rows, w_dev=buf[off:off + w.size]), y), (t.name, shift)


@pytest.mark.parametrize("t", T16_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("n_rows,k", [(11008, 4096), (4096, 11008), (3584, 8192)])
@pytest.mark.parametrize("batch", [8, 32])
def test_mmq_t16_full_size_properties(oracle, t, n_rows, k, batch):
    """BASELINE shapes (gate/up, down, one rank of configs[4]): oracle on a sample of rows, then the size-independent
    properties on every output — bit-reproducible run to run, weight-row permutation, token permutation, exact scaling of
    X by a power of two (fp32 in / out)."""
    from ggq.formats import BLOCK
    if (BLOCK[t][0] == 32 or t in (GGMLType.Q6_K, GGMLType.Q3_K)) and batch > 16:
        batch = 16
    w = synth.random_weight(t, n_rows, k, seed=21)
    wd = util.dev_bytes(w)
    x = _x((batch, k), torch.float32, seed=22)
    y = util.gpu_mmq_t16(w, x, t, n_rows, w_dev=wd)
    rows = np.r_[0:24, n_rows // 2:n_rows // 2 + 24, n_rows - 24:n_rows]
    ref, yabs = oracle.mul_mat_q(w[rows], x.cpu().numpy(), t, len(rows))
    util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref, yabs, torch.float32, f"t16 full {t.name}")
    for _ in range(3):
        assert torch.equal(util.gpu_mmq_t16(w, x, t, n_rows, w_dev=wd), y), "not reproducible run to run"
    perm = np.random.default_rng(0).permutation(n_rows)
    assert torch.equal(util.gpu_mmq_t16(np.ascontiguousarray(w[perm]), x, t, n_rows), y[:, torch.from_numpy(perm).cuda()]), "Y(P W) != Y(W) P"
    tp = torch.from_numpy(np.random.default_rng(3).permutation(batch)).cuda()
    assert torch.equal(util.gpu_mmq_t16(w, x[tp].contiguous(), t, n_rows, w_dev=wd), y[tp]), "Y(P X) != P Y(X)"
    if t not in (GGMLType.Q4_1, GGMLType.Q5_1):   # their fp16 products d d8, m s8 (mmq.cuh:527-529) reach the fp16 subnormal range,
        assert torch.equal(util.gpu_mmq_t16(w, x * 8.0, t, n_rows, w_dev=wd), y * 8.0), "Y(8 X) != 8 Y(X)"   # where rounding is not scale-invariant
