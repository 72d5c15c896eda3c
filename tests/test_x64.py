"""The 64 x 64 wave-tile quantised GEMM (mmq_x64.hip, hand-scheduled K loop; what ggq_mul_mat_q runs from 33 tokens up for the formats it
serves): its activation layout against the re-tiled oracle bytes (bit-exact), the kernel against the oracle (1e-3 relative,
north_star) incl. ragged row / token tiles and unequal K-slices, the exact-integer case, the min-term scale range, the
size-independent exactness properties at the BASELINE shape, and the reference benchmark's own batch (2048 tokens,
HK/tests/kernels/test_cuda_kernels.py:14)."""
import numpy as np
import pytest
import torch

from ggq import synth
from ggq import lib as ggqlib
from ggq.formats import GGMLType, BLOCK
import util

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16, torch.float32]
X64_TYPES = [t for t in (GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0,
                         GGMLType.Q6_K, GGMLType.Q2_K, GGMLType.Q3_K)
             if ggqlib.hip().ggq_mmq_x64_type_supported(int(t))] if torch.cuda.is_available() else []


# formats with 64- and 96-row units (Q5_K has the one-row-tile loop only: 32-row units at every size)
X64_TYPES_64 = [t for t in X64_TYPES if ggqlib.hip().ggq_mmq_x64_unit_rows(int(t), 128, 4096, 11008) == 96] if torch.cuda.is_available() else []


def _x(shape, dtype, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(dtype).cuda()


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k", [(1, 256), (33, 512), (64, 256), (65, 1024), (100, 768), (128, 4096)])
def test_quantize_q8_1_x64_bit_exact(oracle, dtype, t, batch, k):
    x = _x((batch, k), dtype, seed=batch + k)
    got = util.gpu_quantize_q8_1_x64(x, t)
    ref, mask = util.retile_q8_1_x64(oracle.quantize_q8_1_mmq(x.float().cpu().numpy(), t), batch, k, t)
    got = got[:ref.size].reshape(ref.shape)
    assert np.array_equal(got[mask], ref[mask]), "x64 scratch bytes differ from the re-tiled oracle"
    assert not got[~mask].any(), "bytes outside the layout were written"


@pytest.mark.parametrize("dtype", DTYPES, ids=str)
@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("batch,k,n_rows", [(1, 256, 33), (33, 256, 64), (64, 1024, 64), (40, 4096, 130), (65, 768, 31), (128, 2304, 70),
                                            (100, 1280, 95), (129, 512, 200), (70, 11008, 40), (36, 16384, 65),
                                            # batches up to 32 tokens with several super-blocks per K-slice (the one-tile loops where they exist)
                                            (17, 2048, 100), (32, 4096, 70), (24, 1280, 40), (9, 8192, 64), (32, 11008, 33)])
def test_mmq_x64_vs_oracle(oracle, dtype, t, batch, k, n_rows):
    """ragged row tiles and token tiles; 1, 2, 3, 4 ... 64 super-blocks: K-slices of zero, equal and unequal length"""
    assert ggqlib.hip().ggq_mmq_x64_supported(int(t), k, batch) == 1
    w = synth.random_weight(t, n_rows, k, seed=batch + k)
    x = _x((batch, k), dtype, seed=14)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"x64 mmq {t.name} b={batch}")


@pytest.mark.parametrize("t", X64_TYPES_64, ids=lambda t: t.name)
@pytest.mark.parametrize("dtype,batch,k,n_rows", [(torch.float16, 128, 1024, 8230), (torch.float32, 100, 1280, 8257), (torch.bfloat16, 256, 1024, 4100),
                                                  (torch.float16, 2048, 1024, 600), (torch.float16, 128, 4096, 8200)])
def test_mmq_x64_96_row_units(oracle, t, dtype, batch, k, n_rows):
    """launches that take the 96-row units (four two-row-tile waves + four one-row-tile waves per workgroup): against the oracle,
    with a last unit of 6 / 1 / 68 / 24 / 40 rows (the one-row-tile waves partly or not at all in the tensor); and a row's bits do not
    depend on the unit shape — the first rows equal the 64-row-unit launch of those rows alone"""
    L = ggqlib.hip()
    assert L.ggq_mmq_x64_unit_rows(int(t), batch, k, n_rows) == 96
    w = synth.random_weight(t, n_rows, k, seed=batch + k)
    # every seventh row: dmin over the whole range, incl. the 2^-8-scaled cold pass of both wave kinds (test_mmq_x64_min_scale_range)
    from ggq.synth import _F16_FIELDS
    bs, m_off = BLOCK[t][1], _F16_FIELDS[t][1]
    wb = w.reshape(n_rows, -1, bs)
    vals = np.array([6e-8, 1.0, 1023.5, 1024.5, 65504.0, -65504.0, -3.0, 0.0, -2000.0], np.float16)
    for r in range(0, n_rows if (dtype != torch.float16 and m_off is not None) else 0, 7):   # (fp16 outputs would overflow)
        for b in range(wb.shape[1]):
            wb[r, b, m_off:m_off + 2] = vals[(r + 5 * b) % len(vals)].reshape(1).view(np.uint8)
    w = wb.reshape(n_rows, -1)
    x = _x((batch, k), dtype, seed=17)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"x64 96-row units b={batch}")
    if k < 2048:   # (four K-slices whatever the unit count)
        sub = 200
        assert L.ggq_mmq_x64_unit_rows(int(t), batch, k, sub) in (32, 64)   # (32-row units below 160 units of 64 rows)
        ys = util.gpu_mmq_x64(np.ascontiguousarray(w[:sub]), x, t, sub)
        assert torch.equal(ys, y[:, :sub].contiguous()), "a row's result depends on the unit shape"
    else:          # a taller matrix with the same rows first: 64-row units, four K-slices (more than 256 units)
        big = 12352
        assert L.ggq_mmq_x64_unit_rows(int(t), batch, k, big) == 64 and L.ggq_mmq_x64_k_slices(batch, k, big) == 4
        yb = util.gpu_mmq_x64(np.concatenate([w, w[: big - n_rows]]), x, t, big)
        assert torch.equal(yb[:, :n_rows].contiguous(), y), "a row's result depends on the unit shape"
    assert torch.equal(y, util.gpu_mmq_x64(w, x, t, n_rows)), "two launches differ"


@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("dtype,batch,k,n_rows", [(torch.float16, 128, 1024, 4100), (torch.float32, 70, 768, 2500), (torch.bfloat16, 200, 4096, 1030)])
def test_mmq_x64_32_row_units(oracle, t, dtype, batch, k, n_rows):
    """launches below 160 units of 64 rows take 32-row units (every wave a one-row-tile wave; four K-slices, eight at K = 4096): against
    the oracle (ragged last units of 4 / 4 / 6 rows), and — at four K-slices — bit-equal to the 64-row-unit launch of a taller matrix
    that starts with the same rows"""
    L = ggqlib.hip()
    assert L.ggq_mmq_x64_unit_rows(int(t), batch, k, n_rows) == 32
    w = synth.random_weight(t, n_rows, k, seed=batch + k + 1)
    from ggq.synth import _F16_FIELDS
    bs, m_off = BLOCK[t][1], _F16_FIELDS[t][1]
    if m_off is not None and dtype != torch.float16:   # every seventh row: dmin over the whole range incl. the 2^-8-scaled cold pass
        wb = w.reshape(n_rows, -1, bs)
        vals = np.array([6e-8, 1.0, 1023.5, 1024.5, 65504.0, -65504.0, -3.0, 0.0, -2000.0], np.float16)
        for r in range(0, n_rows, 7):
            for b in range(wb.shape[1]):
                wb[r, b, m_off:m_off + 2] = vals[(r + 5 * b) % len(vals)].reshape(1).view(np.uint8)
        w = wb.reshape(n_rows, -1)
    x = _x((batch, k), dtype, seed=19)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"x64 32-row units b={batch}")
    assert torch.equal(y, util.gpu_mmq_x64(w, x, t, n_rows)), "two launches differ"
    if k < 2048 and t in X64_TYPES_64:
        big = 4 * n_rows
        assert L.ggq_mmq_x64_unit_rows(int(t), batch, k, big) == 64 and L.ggq_mmq_x64_k_slices(batch, k, big) == 4
        yb = util.gpu_mmq_x64(np.concatenate([w] * 4), x, t, big)
        assert torch.equal(yb[:, :n_rows].contiguous(), y), "a row's result depends on the unit shape"


@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
@pytest.mark.parametrize("dtype,batch,k,n_rows", [(torch.float16, 1024, 512, 8200), (torch.float32, 2100, 1280, 4100)])
def test_mmq_x64_one_k_slice(oracle, t, dtype, batch, k, n_rows):
    """from 2048 units of 64 rows a launch takes one-wave workgroups (one K-slice: the large batches — 129 x 16 and 65 x 33 units here; Q5_K, on
    32-row units, keeps four-wave workgroups at the same shapes): against the oracle with ragged last units (8 / 4 rows, 52 tokens in the last token tile of the second case) and rows
    that take the 2^-8-scaled cold pass, reproducible run to run, and the integer-exact case"""
    L = ggqlib.hip()
    assert L.ggq_mmq_x64_k_slices(batch, k, n_rows) == 1
    w = synth.random_weight(t, n_rows, k, seed=batch + k + 2)
    from ggq.synth import _F16_FIELDS
    bs, m_off = BLOCK[t][1], _F16_FIELDS[t][1]
    if m_off is not None and dtype != torch.float16:
        wb = w.reshape(n_rows, -1, bs)
        vals = np.array([6e-8, 1.0, 1023.5, 1024.5, 65504.0, -65504.0, -3.0, 0.0, -2000.0], np.float16)
        for r in range(0, n_rows, 7):
            for b in range(wb.shape[1]):
                wb[r, b, m_off:m_off + 2] = vals[(r + 5 * b) % len(vals)].reshape(1).view(np.uint8)
        w = wb.reshape(n_rows, -1)
    x = _x((batch, k), dtype, seed=23)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, dtype, f"x64 one K-slice b={batch}")
    for _ in range(4):
        assert torch.equal(y, util.gpu_mmq_x64(w, x, t, n_rows)), "two launches differ"
    if dtype == torch.float16:
        _integer_exact(oracle, t, 4096, 256, 2100)   # 33 x 64 units


@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
def test_mmq_x64_integer_exact(oracle, t):
    """power-of-two scales + integer activations: every product and partial sum is exact, so the result must equal the oracle's to
    the last bit (integer unpack, operand order and the int8 MFMA contraction)"""
    from ggq.synth import _F16_FIELDS
    qk, bs = BLOCK[t]
    _integer_exact(oracle, t, 96, 1024, 70)
    _integer_exact(oracle, t, 70, 2048, 20)   # (up to 32 tokens: the one-tile loops, two super-blocks per K-slice)


def _integer_exact(oracle, t, n_rows, k, batch):
    from ggq.synth import _F16_FIELDS
    qk, bs = BLOCK[t]
    w = synth.random_weight(t, n_rows, k, seed=9).reshape(-1, bs)
    d_off, m_off = _F16_FIELDS[t]
    w[:, d_off:d_off + 2] = np.array([2.0 ** -4], np.float16).view(np.uint8)
    if m_off is not None:
        w[:, m_off:m_off + 2] = np.array([2.0 ** -3], np.float16).view(np.uint8)
    rng = np.random.default_rng(1)
    xi = rng.integers(-8, 9, size=(batch, k)).astype(np.float32)
    xi[:, ::32] = 127.0
    y = util.gpu_mmq_x64(w.reshape(n_rows, -1), torch.from_numpy(xi).cuda(), t, n_rows).cpu().numpy()
    ref, _ = oracle.mul_mat_q(w.reshape(n_rows, -1), xi, t, n_rows)
    assert np.array_equal(y, ref), f"{t.name}: exact-integer x64 MMQ differs"


@pytest.mark.parametrize("batch", [40, 20, 128])   # 20: the one-tile loops; 128 with 768 rows: 64-row... 32-row units (both row-tile kinds via 96-row tests)
@pytest.mark.parametrize("t", [t for t in X64_TYPES if t in (GGMLType.Q4_K, GGMLType.Q5_K)], ids=lambda t: t.name)
def test_mmq_x64_min_scale_range(oracle, t, batch):
    """the min term's exact hi + lo fp16 split over the whole range of dmin: subnormal remainders, the 1024 threshold of the
    2^-8-scaled cold pass (rows above and below it in ONE tile), fp16 max, negative and zero dmin"""
    from ggq.synth import _F16_FIELDS
    qk, bs = BLOCK[t]
    n_rows, k = 70, 768 if batch != 20 else 2048
    w = synth.random_weight(t, n_rows, k, seed=3).reshape(n_rows, -1, bs)
    _, m_off = _F16_FIELDS[t]
    vals = np.array([6e-8, 6.1e-5, 1.0, 1023.5, 1024.0, 1024.5, 65504.0, -65504.0, -3.0, 0.0, 2.0 ** -14, -2000.0], np.float16)
    for r in range(n_rows):
        for b in range(w.shape[1]):
            w[r, b, m_off:m_off + 2] = vals[(r + 5 * b) % len(vals)].reshape(1).view(np.uint8)
    w = w.reshape(n_rows, -1)
    x = _x((batch, k), torch.float32, seed=4)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    util.assert_fp_accumulate(y, ref, yabs, torch.float32, f"x64 min-term range {t.name}")
    if batch == 20:   # the one-tile loops against the two-tile loops: the same tokens inside a 100-token launch (same units, same K-slices), bit for bit
        assert ggqlib.hip().ggq_mmq_x64_tile_tokens(int(t), 20, k, n_rows) == 32 and ggqlib.hip().ggq_mmq_x64_tile_tokens(int(t), 100, k, n_rows) == 64
        x100 = torch.cat([x, _x((80, k), torch.float32, seed=5)])
        assert torch.equal(util.gpu_mmq_x64(w, x100, t, n_rows)[:20], y), "one-tile and two-tile loops differ"
        x50 = x100[:50].contiguous()   # 33 - 64 tokens on few rows: two one-tile units per 32 rows
        assert ggqlib.hip().ggq_mmq_x64_tile_tokens(int(t), 50, k, n_rows) == 32
        assert torch.equal(util.gpu_mmq_x64(w, x50, t, n_rows), util.gpu_mmq_x64(w, x100, t, n_rows)[:50]), "two one-tile units and one two-tile unit differ"


def test_mmq_x64_ldy_epilogues_unaligned_and_errors(oracle):
    L = ggqlib.hip()
    t, batch, k, n_rows, ldy = GGMLType.Q4_K, 50, 512, 72, 104
    w = synth.random_weight(t, n_rows, k, seed=5)
    x = _x((batch, k), torch.float16, seed=15)
    ref, yabs = oracle.mul_mat_q(w, x.float().cpu().numpy(), t, n_rows)
    y = util.gpu_mmq_x64(w, x, t, n_rows, ldy=ldy)
    util.assert_fp_accumulate(y[:, :n_rows], ref, yabs, torch.float16, "x64 ldy")
    assert torch.count_nonzero(y[:, n_rows:]) == 0, "columns beyond n_rows were written"
    # an odd row pitch takes the scalar stores
    y3 = util.gpu_mmq_x64(w, x, t, n_rows, ldy=77)
    assert torch.equal(y3[:, :n_rows], y[:, :n_rows])
    # weight pointer at an odd multiple of 2 (the ABI asks for 2-byte alignment only)
    buf = torch.zeros(w.size + 64, dtype=torch.uint8, device="cuda")
    off = (-buf.data_ptr()) % 16 + 2
    buf[off:off + w.size] = torch.from_numpy(w.reshape(-1)).cuda()
    y2 = util.gpu_mmq_x64(w, x, t, n_rows, w_dev=buf[off:])
    assert torch.equal(y2, y[:, :n_rows].contiguous())
    bias = torch.randn(n_rows, generator=torch.Generator().manual_seed(2)).half().cuda()
    yb = util.gpu_mmq_x64(w, x, t, n_rows, epilogue=1, aux=bias)
    util.assert_fp_accumulate(yb, ref + bias.float().cpu().numpy()[None, :], yabs + np.abs(bias.float().cpu().numpy())[None, :], torch.float16, "x64 bias")
    gate = torch.randn((batch, n_rows), generator=torch.Generator().manual_seed(3)).half().cuda()
    yg = util.gpu_mmq_x64(w, x, t, n_rows, epilogue=2, aux=gate)
    gf = gate.float().cpu().numpy().astype(np.float64)
    sil = gf / (1.0 + np.exp(-gf))
    util.assert_fp_accumulate(yg, (ref * sil).astype(np.float32), (yabs * np.abs(sil)).astype(np.float32), torch.float16, "x64 silu_mul")
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    wd, st = util.dev_bytes(w), util.stream_ptr()
    args = lambda **kw: [kw.get("w", util.vp(wd)), kw.get("q", util.vp(q)), util.vp(y), kw.get("t", int(t)), kw.get("dt", 1), kw.get("b", batch),
                         kw.get("k", k), n_rows, kw.get("ldy", ldy), kw.get("epi", 0), None, st]
    assert L.ggq_mul_mat_q_x64(*args(t=1)) == -1
    assert L.ggq_mul_mat_q_x64(*args(dt=7)) == -3
    assert L.ggq_mul_mat_q_x64(*args(ldy=n_rows - 1)) == -4
    assert L.ggq_mul_mat_q_x64(*args(epi=1)) == -4      # epilogue without aux
    assert L.ggq_mul_mat_q_x64(*args(b=0)) == 0
    assert L.ggq_mul_mat_q_x64(*args(k=k + 32)) == -2
    import ctypes
    assert L.ggq_mul_mat_q_x64(*args(w=ctypes.c_void_p(wd.data_ptr() + 1))) == -6
    assert L.ggq_quantize_q8_1_x64(util.vp(x), 1, ctypes.c_void_p(q.data_ptr() + 4), batch, k, int(t), st) == -6
    assert L.ggq_mmq_x64_supported(int(t), 4096 + 32, 64) == 0
    assert L.ggq_mmq_x64_supported(1, 4096, 64) == 0


@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
def test_mmq_x64_full_size_properties(oracle, t):
    """BASELINE configs[3] shape (11008 x 4096, batch 128), size-independent properties: repeated launches, a row permutation
    and a token permutation are bit-exact; a sample of rows agrees with the oracle"""
    n_rows, k, batch = 11008, 4096, 128
    w = synth.random_weight(t, n_rows, k, seed=21)
    x = _x((batch, k), torch.float16, seed=22)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    assert torch.equal(y, util.gpu_mmq_x64(w, x, t, n_rows)), "two launches differ"
    rng = np.random.default_rng(5)
    perm = rng.permutation(n_rows)
    yp = util.gpu_mmq_x64(np.ascontiguousarray(w[perm]), x, t, n_rows)
    assert torch.equal(yp, y[:, torch.from_numpy(perm).cuda()]), "row permutation changed bits"
    tperm = torch.from_numpy(rng.permutation(batch)).cuda()
    yt = util.gpu_mmq_x64(w, x[tperm].contiguous(), t, n_rows)
    assert torch.equal(yt, y[tperm]), "token permutation changed bits"
    rows = np.sort(rng.choice(n_rows, 96, replace=False))
    ref, yabs = oracle.mul_mat_q(np.ascontiguousarray(w[rows]), x.float().cpu().numpy(), t, len(rows))
    util.assert_fp_accumulate(y[:, torch.from_numpy(rows).cuda()], ref, yabs, torch.float16, f"x64 full size {t.name}")


@pytest.mark.parametrize("t", X64_TYPES, ids=lambda t: t.name)
def test_mmq_x64_reference_benchmark_batch(oracle, t):
    """2048 tokens (the reference's test batch, HK/tests/kernels/test_cuda_kernels.py:14; its benchmark defaults to 4096): a row
    sample against the oracle, and the first 128 tokens bit-equal to the batch-128 launch of the same rows (a token's result
    does not depend on which tokens share its launch)"""
    n_rows, k, batch = 512, 4096, 2048
    w = synth.random_weight(t, n_rows, k, seed=31)
    x = _x((batch, k), torch.float16, seed=32)
    y = util.gpu_mmq_x64(w, x, t, n_rows)
    y128 = util.gpu_mmq_x64(w, x[:128].contiguous(), t, n_rows)
    if t in X64_TYPES_64:
        assert torch.equal(y[:128], y128)
    else:   # Q5_K: 32-row units at both sizes, but 512 of them take four K-slices and 32 take eight: same sum, another order (include/ggq.h)
        assert torch.allclose(y[:128].float(), y128.float(), rtol=2e-3, atol=2e-2)
    rows = np.arange(0, n_rows, 16)
    toks = np.arange(0, batch, 37)
    ref, yabs = oracle.mul_mat_q(np.ascontiguousarray(w[rows]), x.float().cpu().numpy()[toks], t, len(rows))
    sub = y[torch.from_numpy(toks).cuda()][:, torch.from_numpy(rows).cuda()]
    util.assert_fp_accumulate(sub, ref, yabs, torch.float16, f"x64 batch 2048 {t.name}")
