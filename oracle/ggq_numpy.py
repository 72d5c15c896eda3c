"""Independent numpy derivation of the block formats — TEST INFRASTRUCTURE ONLY.

Two things live here, both written from the wire-format tables (SURVEY.md §2.2 /
HK/ggml/ggml-common.h:17-108), *not* from oracle/ggq_oracle.c, so that two
independently written decoders have to agree bit-for-bit:

1. ``unpack_ints(blocks, type)`` / ``scales16(blocks, type)`` — integer unpack of
   every format into the integers the dot products use, vectorised per block
   (layout derived element-by-element, no thread emulation).
2. ``gguf_dequantize(blocks, type)`` — restatement of the published
   ``gguf.quants.dequantize`` algorithm (gguf-py, floor ``gguf>=0.10.0`` in the
   reference's requirements-dev.txt:4 and HK/requirements-test.txt:4; the package is
   absent from this image).  It is the ground truth the reference's own tests use
   (tests/test_dequantize.py:66, HK/tests/kernels/test_cuda_kernels.py:52) at
   atol=1e-2, rtol=4e-2.  fp32 arithmetic, same operation order as gguf-py.
3. ``dequantize_f16(blocks, type)`` — the fp16-arithmetic sequence of
   HK/ggml/dequantize.cuh:3-254 expressed with numpy float16 ops.

Nothing in the product path imports this module.
"""
import numpy as np

Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K = 10, 11, 12, 13, 14
IQ4_NL, IQ4_XS = 20, 23
IQ2_XXS, IQ2_XS, IQ3_XXS, IQ1_S, IQ3_S, IQ2_S, IQ1_M = 16, 17, 18, 19, 21, 22, 29
GRID_IQ = (IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S)
KVALUES_IQ4NL = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.int32)

BLOCK_ELEMS = {Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32,
               Q2_K: 256, Q3_K: 256, Q4_K: 256, Q5_K: 256, Q6_K: 256, IQ4_NL: 32, IQ4_XS: 256,
               IQ2_XXS: 256, IQ2_XS: 256, IQ2_S: 256, IQ3_XXS: 256, IQ3_S: 256, IQ1_S: 256, IQ1_M: 256}
BLOCK_BYTES = {Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 36,
               Q2_K: 84, Q3_K: 110, Q4_K: 144, Q5_K: 176, Q6_K: 210, IQ4_NL: 18, IQ4_XS: 136,
               IQ2_XXS: 66, IQ2_XS: 74, IQ2_S: 82, IQ3_XXS: 98, IQ3_S: 110, IQ1_S: 50, IQ1_M: 56}
NAMES = {Q4_0: "Q4_0", Q4_1: "Q4_1", Q5_0: "Q5_0", Q5_1: "Q5_1", Q8_0: "Q8_0",
         Q2_K: "Q2_K", Q3_K: "Q3_K", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", IQ4_NL: "IQ4_NL", IQ4_XS: "IQ4_XS",
         IQ2_XXS: "IQ2_XXS", IQ2_XS: "IQ2_XS", IQ2_S: "IQ2_S", IQ3_XXS: "IQ3_XXS", IQ3_S: "IQ3_S", IQ1_S: "IQ1_S", IQ1_M: "IQ1_M"}
WEIGHT_TYPES = [Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K]


def as_blocks(data, t):
    a = np.ascontiguousarray(np.asarray(data, dtype=np.uint8)).reshape(-1, BLOCK_BYTES[t])
    return a


def _f16(bytes2):
    """[nb,2] uint8 -> [nb] float16"""
    return np.ascontiguousarray(bytes2).view(np.float16).reshape(-1)


def _bits(byte_arr, nbits=8):
    """[..., n] uint8 -> [..., n, nbits] of 0/1 (LSB first)"""
    return (byte_arr[..., None] >> np.arange(nbits, dtype=np.uint8)) & 1


def _k4_scales(sc12):
    """Q4_K/Q5_K 12-byte scale field -> (sc[nb,8], mn[nb,8]); 6-bit values.
    bytes 0-3: low 6 bits = sc0..3, top 2 bits = high bits of sc4..7
    bytes 4-7: low 6 bits = mn0..3, top 2 bits = high bits of mn4..7
    bytes 8-11: low nibble = low 4 bits of sc4..7, high nibble = low 4 bits of mn4..7"""
    s = sc12.astype(np.int32)
    sc = np.empty(s.shape[:-1] + (8,), np.int32)
    mn = np.empty_like(sc)
    sc[..., 0:4] = s[..., 0:4] & 63
    mn[..., 0:4] = s[..., 4:8] & 63
    sc[..., 4:8] = (s[..., 8:12] & 15) | ((s[..., 0:4] >> 6) << 4)
    mn[..., 4:8] = (s[..., 8:12] >> 4) | ((s[..., 4:8] >> 6) << 4)
    return sc, mn


def _q3k_scales(sc12):
    """Q3_K 12-byte field -> [nb,16] signed 6-bit scales minus 32.
    scale i: low 4 bits = nibble (i//8) of byte i%8; high 2 bits = bit-pair (i//4) of byte 8 + i%4"""
    s = sc12.astype(np.int32)
    i = np.arange(16)
    lo = (s[..., i % 8] >> (4 * (i // 8))) & 15
    hi = (s[..., 8 + i % 4] >> (2 * (i // 4))) & 3
    return (lo | (hi << 4)) - 32


def unpack_ints(blocks, t):
    """Integers multiplied with q8 in the reference's dot products, element order.
    Raw unsigned for Q4_0/Q4_1/Q5_0/Q5_1/Q2_K/Q4_K/Q5_K; signed for Q8_0/Q3_K/Q6_K."""
    b = as_blocks(blocks, t)
    nb = b.shape[0]
    if t in (Q4_0, Q4_1):
        qs = b[:, 2:18] if t == Q4_0 else b[:, 4:20]
        return np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.int32)
    if t in (Q5_0, Q5_1):
        qh = b[:, 2:6] if t == Q5_0 else b[:, 4:8]
        qs = b[:, 6:22] if t == Q5_0 else b[:, 8:24]
        hb = _bits(qh).reshape(nb, 32).astype(np.int32)  # element j <- bit j of the 32-bit qh
        lo4 = np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.int32)
        return lo4 | (hb << 4)
    if t == Q8_0:
        return b[:, 2:34].view(np.int8).astype(np.int32)
    if t == Q2_K:
        qs = b[:, 16:80].reshape(nb, 2, 1, 32)
        sh = (2 * np.arange(4, dtype=np.uint8)).reshape(1, 1, 4, 1)
        return ((qs >> sh) & 3).reshape(nb, 256).astype(np.int32)
    if t == Q3_K:
        qs = b[:, 32:96].reshape(nb, 2, 1, 32)
        sh = (2 * np.arange(4, dtype=np.uint8)).reshape(1, 1, 4, 1)
        lo = ((qs >> sh) & 3).astype(np.int32)                       # [nb,2,4,32]
        hm = _bits(b[:, 0:32])                                        # [nb,32(l),8(bit)]
        hbit = hm.transpose(0, 2, 1).reshape(nb, 2, 4, 32).astype(np.int32)  # bit 4n+j of hmask[l]
        return (lo - 4 * (1 - hbit)).reshape(nb, 256)
    if t == Q4_K:
        qs = b[:, 16:144].reshape(nb, 4, 1, 32)
        sh = np.array([0, 4], dtype=np.uint8).reshape(1, 1, 2, 1)
        return ((qs >> sh) & 15).reshape(nb, 256).astype(np.int32)
    if t == Q5_K:
        qs = b[:, 48:176].reshape(nb, 4, 1, 32)
        sh = np.array([0, 4], dtype=np.uint8).reshape(1, 1, 2, 1)
        lo = ((qs >> sh) & 15).astype(np.int32)                       # [nb,4,2,32]
        hb = _bits(b[:, 16:48]).transpose(0, 2, 1).reshape(nb, 4, 2, 32).astype(np.int32)
        return (lo + 16 * hb).reshape(nb, 256)
    if t == Q6_K:
        ql = b[:, 0:128].reshape(nb, 2, 2, 32)      # [ip][half 0/1 of the 64 bytes][l]
        qh = b[:, 128:192].reshape(nb, 2, 32)
        out = np.empty((nb, 2, 4, 32), np.int32)
        for j in range(4):                            # 32-element group j inside the 128 half
            nib = (ql[:, :, j % 2, :] >> (4 * (j // 2))) & 15
            hi2 = (qh >> (2 * j)) & 3
            out[:, :, j, :] = (nib | (hi2 << 4)).astype(np.int32) - 32
        return out.reshape(nb, 256)
    raise ValueError(t)


def iq4_codes(blocks, t):
    """IQ4_NL / IQ4_XS (block layouts HK/ggml/ggml-common.h:176-191): codebook values in element order
    [nb, qk] and the integer scale of each 32-element sub-block [nb, qk/32] (1 for IQ4_NL, ls - 32 for IQ4_XS)."""
    b = as_blocks(blocks, t)
    nb = b.shape[0]
    if t == IQ4_NL:
        qs = b[:, 2:18]
        return KVALUES_IQ4NL[np.concatenate([qs & 15, qs >> 4], axis=1)], np.ones((nb, 1), np.int32), _f16(b[:, 0:2])
    qs = b[:, 8:136].reshape(nb, 8, 16)
    idx = np.concatenate([qs & 15, qs >> 4], axis=2).reshape(nb, 256)      # sub-block ib: low nibbles then high nibbles
    sh = b[:, 2].astype(np.int32) | (b[:, 3].astype(np.int32) << 8)
    ib = np.arange(8)
    lo = (b[:, 4:8].astype(np.int32)[:, ib // 2] >> (4 * (ib % 2))) & 15
    hi = (sh[:, None] >> (2 * ib)) & 3
    return KVALUES_IQ4NL[idx], (lo | (hi << 4)) - 32, _f16(b[:, 0:2])


_GRIDS = None


def iq_grids():
    """the codebook grids as numpy arrays, read from the data header the kernels use (constant data transcribed from the
    reference's HK/ggml/ggml-common.h:193-1011): name -> uint8 [n, 8 | 4] magnitudes (iq1s: uint32 [2048])"""
    global _GRIDS
    if _GRIDS is None:
        import os
        import re
        here = os.path.dirname(os.path.abspath(__file__))
        txt = open(os.path.join(here, "..", "ggml-libtorch_amd", "csrc", "hip", "iq_tables.h")).read()
        g = {}
        for m in re.finditer(r"ggq_(\w+)\[(\d+)\] = \{(.*?)\};", txt, re.S):
            vals = [int(v.rstrip("ul"), 16) for v in re.findall(r"0x[0-9a-f]+u?l?l?", m.group(3))]
            assert len(vals) == int(m.group(2))
            a = np.array(vals, dtype=np.uint64)
            if m.group(1).startswith("iq1s"):
                g[m.group(1)] = a.astype(np.uint32)
            else:
                nbytes = 8 if m.group(1).startswith("iq2") else 4
                g[m.group(1)] = ((a[:, None] >> (8 * np.arange(nbytes, dtype=np.uint64))) & np.uint64(0xFF)).astype(np.uint8)
        _GRIDS = g
    return _GRIDS


def _ksigns(i7):
    """ksigns_iq2xs: the 7-bit value with an 8th bit that makes the popcount even -> 8 sign bits"""
    i7 = i7.astype(np.int64)
    par = np.zeros_like(i7)
    for b in range(7):
        par ^= (i7 >> b) & 1
    return i7 | (par << 7)


def iq_grid_codes(blocks, t):
    """IQ2_XXS / IQ2_XS / IQ2_S / IQ3_XXS / IQ3_S (block layouts HK/ggml/ggml-common.h:108-149), element order:
    signed grid values int32 [nb, 256], the float scale factor of each 16 elements float32 [nb, 16] (0.5 + scale nibble),
    the format's constant (0.25 | 0.5) and d [nb] — derived per field from the struct layouts, vectorised over blocks."""
    b = as_blocks(blocks, t)
    nb = b.shape[0]
    G = iq_grids()
    bi = b.astype(np.int64)
    d = _f16(b[:, 0:2])
    ib = np.arange(8)
    if t == IQ2_XXS:      # per sub-block: 4 index bytes, then a uint32: 4 x 7 sign bits | scale << 28
        sub = bi[:, 2:66].reshape(nb, 8, 8)
        idx = sub[:, :, 0:4]
        aux = sub[:, :, 4] | (sub[:, :, 5] << 8) | (sub[:, :, 6] << 16) | (sub[:, :, 7] << 24)
        mag = G["iq2xxs_grid"][idx]                                          # [nb, 8, 4, 8]
        sg = _ksigns((aux[:, :, None] >> (7 * np.arange(4))) & 127)           # [nb, 8, 4]
        mul = np.repeat((0.5 + (aux >> 28)).astype(np.float32), 2, axis=1)    # per sub-block -> per 16
        post = 0.25
    elif t == IQ2_XS:     # uint16 qs[32]: 9-bit grid index | 7 sign bits; uint8 scales[8]: two nibbles per sub-block
        q2 = (bi[:, 2:66:2] | (bi[:, 3:66:2] << 8)).reshape(nb, 8, 4)
        mag = G["iq2xs_grid"][q2 & 511]
        sg = _ksigns(q2 >> 9)
        sc = bi[:, 66:74]
        mul = (0.5 + np.stack([sc & 15, sc >> 4], axis=2).reshape(nb, 16)).astype(np.float32)
        post = 0.25
    elif t == IQ2_S:      # qs[0..31] low index bytes, qs[32..63] sign bytes, qh[8] two high index bits per run, scales[8]
        lo = bi[:, 2:34].reshape(nb, 8, 4)
        qh = bi[:, 66:74]
        hi = (qh[:, :, None] >> (2 * np.arange(4))) & 3
        mag = G["iq2s_grid"][lo | (hi << 8)]
        sg = bi[:, 34:66].reshape(nb, 8, 4)
        sc = bi[:, 74:82]
        mul = (0.5 + np.stack([sc & 15, sc >> 4], axis=2).reshape(nb, 16)).astype(np.float32)
        post = 0.25
    elif t == IQ3_XXS:    # qs[0..63]: one grid index per 4 elements; then 8 x uint32: 4 x 7 sign bits | scale << 28
        idx = bi[:, 2:66].reshape(nb, 8, 4, 2)
        gas = bi[:, 66:98].reshape(nb, 8, 4)
        aux = gas[:, :, 0] | (gas[:, :, 1] << 8) | (gas[:, :, 2] << 16) | (gas[:, :, 3] << 24)
        mag = G["iq3xxs_grid"][idx].reshape(nb, 8, 4, 8)
        sg = _ksigns((aux[:, :, None] >> (7 * np.arange(4))) & 127)
        mul = np.repeat((0.5 + (aux >> 28)).astype(np.float32), 2, axis=1)
        post = 0.5
    else:                 # IQ3_S: qs[64] low index bytes, qh[8] one high bit per index, signs[32], scales[4] (nibble per sub-block)
        lo = bi[:, 2:66].reshape(nb, 8, 8)
        qh = bi[:, 66:74]
        hi = (qh[:, :, None] >> np.arange(8)) & 1
        mag = G["iq3xs_grid"][lo | (hi << 8)].reshape(nb, 8, 4, 8)
        sg = bi[:, 74:106].reshape(nb, 8, 4)
        sc = bi[:, 106:110]
        nib = np.stack([sc & 15, sc >> 4], axis=2).reshape(nb, 8)
        mul = np.repeat((0.5 + nib).astype(np.float32), 2, axis=1)
        post = 0.5
    neg = (sg[..., None] >> np.arange(8)) & 1                                 # [nb, 8, 4, 8]
    vals = np.where(neg == 1, -mag.astype(np.int32), mag.astype(np.int32)).reshape(nb, 256)
    return vals, mul, np.float32(post), d


def iq1_codes(blocks, t):
    """IQ1_S / IQ1_M (HK/ggml/ggml-common.h:151-174): q in {0,1,2} int32 [nb, 256], delta per 8 elements float32 [nb, 32]
    (-1 +- 0.125), the integer scale 2 s + 1 per 16 elements int32 [nb, 16] and the fp16 super-block scale [nb]."""
    b = as_blocks(blocks, t)
    nb = b.shape[0]
    bi = b.astype(np.int64)
    grid = iq_grids()["iq1s_grid_gpu"].astype(np.int64)
    il = np.arange(4)
    if t == IQ1_S:
        d = _f16(b[:, 0:2])
        qs = bi[:, 2:34].reshape(nb, 8, 4)
        qh = bi[:, 34:50:2] | (bi[:, 35:50:2] << 8)                           # [nb, 8]
        idx = qs | (((qh[:, :, None] >> (3 * il)) & 7) << 8)
        delta = np.where(qh & 0x8000, -1.125, -0.875).astype(np.float32)
        delta = np.repeat(delta, 4, axis=1)
        sc = np.repeat(2 * ((qh >> 12) & 7) + 1, 2, axis=1).astype(np.int32)
    else:
        qs = bi[:, 0:32].reshape(nb, 8, 4)
        qh = bi[:, 32:48].reshape(nb, 8, 2)                                   # one byte per two runs
        nibq = (qh[:, :, il // 2] >> (4 * (il % 2))) & 15                     # [nb, 8, 4]: 3 index bits + the delta bit
        idx = qs | ((nibq & 7) << 8)
        delta = np.where(nibq & 8, -1.125, -0.875).astype(np.float32).reshape(nb, 32)
        sc16 = bi[:, 48:56:2] | (bi[:, 49:56:2] << 8)                         # [nb, 4] uint16
        d = np.ascontiguousarray(((sc16[:, 0] >> 12) | ((sc16[:, 1] >> 8) & 0xF0) | ((sc16[:, 2] >> 4) & 0xF00) | (sc16[:, 3] & 0xF000)).astype(np.uint16)).view(np.float16)
        i16 = np.arange(16)
        sc = (2 * ((sc16[:, i16 // 4] >> (3 * (i16 % 4))) & 7) + 1).astype(np.int32)
    g = grid[idx]                                                              # [nb, 8, 4]
    q = np.concatenate([(g[..., None] >> (8 * np.arange(4))) & 15, (g[..., None] >> (8 * np.arange(4) + 4)) & 15], axis=-1)
    return q.reshape(nb, 256).astype(np.int32), delta, sc, d


def scales16(blocks, t):
    """(d, dmin_or_m, sc16, mn16): fp16 block scales and per-16-element integer scale/min.
    Legacy formats: sc16/mn16 are None."""
    b = as_blocks(blocks, t)
    if t in (Q4_0, Q5_0, Q8_0):
        return _f16(b[:, 0:2]), None, None, None
    if t in (Q4_1, Q5_1):
        return _f16(b[:, 0:2]), _f16(b[:, 2:4]), None, None
    if t == Q2_K:
        sc = b[:, 0:16].astype(np.int32)
        return _f16(b[:, 80:82]), _f16(b[:, 82:84]), sc & 15, sc >> 4
    if t == Q3_K:
        return _f16(b[:, 108:110]), None, _q3k_scales(b[:, 96:108]), None
    if t in (Q4_K, Q5_K):
        sc, mn = _k4_scales(b[:, 4:16])
        return _f16(b[:, 0:2]), _f16(b[:, 2:4]), np.repeat(sc, 2, axis=1), np.repeat(mn, 2, axis=1)
    if t == Q6_K:
        return _f16(b[:, 208:210]), None, b[:, 192:208].view(np.int8).astype(np.int32), None
    raise ValueError(t)


def dequantize_exact(blocks, t):
    """float64 mathematical definition, no intermediate rounding."""
    if t in (IQ4_NL, IQ4_XS):
        v, ls, d = iq4_codes(blocks, t)
        return d.astype(np.float64)[:, None] * np.repeat(ls, 32, axis=1) * v
    if t in GRID_IQ:
        v, mul, post, d = iq_grid_codes(blocks, t)
        return d.astype(np.float64)[:, None] * np.repeat(mul, 16, axis=1).astype(np.float64) * float(post) * v
    if t in (IQ1_S, IQ1_M):
        q, delta, sc, d = iq1_codes(blocks, t)
        return d.astype(np.float64)[:, None] * np.repeat(sc, 16, axis=1) * (q + np.repeat(delta, 8, axis=1).astype(np.float64))
    q = unpack_ints(blocks, t).astype(np.float64)
    d, m, sc, mn = scales16(blocks, t)
    d = d.astype(np.float64)[:, None]
    if t == Q4_0:
        return d * (q - 8)
    if t == Q5_0:
        return d * (q - 16)
    if t == Q8_0:
        return d * q
    if t in (Q4_1, Q5_1):
        return d * q + m.astype(np.float64)[:, None]
    s = np.repeat(sc, 16, axis=1).astype(np.float64)
    if t in (Q3_K, Q6_K):
        return d * s * q
    return d * s * q - m.astype(np.float64)[:, None] * np.repeat(mn, 16, axis=1)


def gguf_dequantize(blocks, t):
    """gguf-py ``quants.dequantize`` restated (fp32, gguf-py operation order)."""
    f32 = np.float32
    if t in (IQ4_NL, IQ4_XS):   # gguf-py: d * kvalues (IQ4_NL); dl = d * (scales - 32), dl * kvalues (IQ4_XS)
        v, ls, dd = iq4_codes(blocks, t)
        dl = dd.astype(f32)[:, None] * np.repeat(ls, 32, axis=1).astype(f32) if t == IQ4_XS else dd.astype(f32)[:, None]
        return dl * v.astype(f32)
    if t in GRID_IQ:   # gguf-py: db = d * (0.5 + scales) * 0.25 | 0.5; db * grid * signs
        v, mul, post, dd = iq_grid_codes(blocks, t)
        return (dd.astype(f32)[:, None] * np.repeat(mul, 16, axis=1) * post) * v.astype(f32)
    if t in (IQ1_S, IQ1_M):   # gguf-py: dl = d * (2 * scale + 1); dl * (grid + delta)
        qq, delta, sc1, dd = iq1_codes(blocks, t)
        return (dd.astype(f32)[:, None] * np.repeat(sc1, 16, axis=1).astype(f32)) * (qq.astype(f32) + np.repeat(delta, 8, axis=1))
    q = unpack_ints(blocks, t)
    d, m, sc, mn = scales16(blocks, t)
    d = d.astype(f32)[:, None]
    if t == Q4_0:
        return d * (q - 8).astype(f32)
    if t == Q5_0:
        return d * (q - 16).astype(f32)
    if t == Q8_0:
        return q.astype(f32) * d
    if t in (Q4_1, Q5_1):
        return d * q.astype(f32) + m.astype(f32)[:, None]
    nb = q.shape[0]
    if t == Q2_K:
        dl = (d * sc.astype(f32)).reshape(nb, 16, 1)
        ml = (m.astype(f32)[:, None] * mn.astype(f32)).reshape(nb, 16, 1)
        return (dl * q.reshape(nb, 16, 16).astype(f32) - ml).reshape(nb, 256)
    if t == Q3_K:
        dl = (d * sc.astype(f32)).reshape(nb, 16, 1)
        return (dl * q.reshape(nb, 16, 16).astype(f32)).reshape(nb, 256)
    if t in (Q4_K, Q5_K):
        dl = (d * sc[:, ::2].astype(f32)).reshape(nb, 8, 1)
        ml = (m.astype(f32)[:, None] * mn[:, ::2].astype(f32)).reshape(nb, 8, 1)
        return (dl * q.reshape(nb, 8, 32).astype(f32) - ml).reshape(nb, 256)
    if t == Q6_K:
        dl = (d * sc.astype(f32)).reshape(nb, 16, 1)
        return (dl * q.reshape(nb, 16, 16).astype(f32)).reshape(nb, 256)
    raise ValueError(t)


def dequantize_f16(blocks, t):
    """fp16-arithmetic sequence of HK/ggml/dequantize.cuh, numpy float16 ops
    (numpy evaluates each float16 op in float32 and rounds once = IEEE fp16)."""
    h = np.float16
    if t in (IQ4_NL, IQ4_XS):   # fp32: (d * (ls - 32)) * kvalue, one rounding to fp16 (dequantize.cuh:411-415, 428-432)
        v, ls, d = iq4_codes(blocks, t)
        with np.errstate(all="ignore"):
            dl = d.astype(np.float32)[:, None] * np.repeat(ls, 32, axis=1).astype(np.float32) if t == IQ4_XS else d.astype(np.float32)[:, None]
            return (dl * v.astype(np.float32)).astype(h)
    if t in GRID_IQ:   # fp32: ((half2float(d) * (0.5f + s)) * post) * grid * (+-1), one rounding to fp16 (dequantize.cuh:256-352)
        v, mul, post, d = iq_grid_codes(blocks, t)
        with np.errstate(all="ignore"):
            dd = (d.astype(np.float32)[:, None] * np.repeat(mul, 16, axis=1)) * post
            return (dd * v.astype(np.float32)).astype(h)
    if t in (IQ1_S, IQ1_M):   # fp32: d * (q + delta), d = half2float(d) * (2 s + 1) (dequantize.cuh:354-398)
        q, delta, sc, d = iq1_codes(blocks, t)
        with np.errstate(all="ignore"):
            dd = d.astype(np.float32)[:, None] * np.repeat(sc, 16, axis=1).astype(np.float32)
            return (dd * (q.astype(np.float32) + np.repeat(delta, 8, axis=1))).astype(h)
    q = unpack_ints(blocks, t)
    d, m, sc, mn = scales16(blocks, t)
    d = d[:, None]
    old = np.seterr(all="ignore")
    try:
        if t == Q4_0:       # hmul2(hsub2(v, 8), d)  (dequantize.cuh:14-15)
            return (q.astype(h) - h(8)) * d
        if t == Q5_0:       # :48-49
            return (q.astype(h) - h(16)) * d
        if t == Q8_0:       # :77
            return q.astype(h) * d
        if t in (Q4_1, Q5_1):  # hadd2(hmul2(v, d), m)  (:30-31, :67-68)
            return q.astype(h) * d + m[:, None]
        s = np.repeat(sc, 16, axis=1)
        if t == Q2_K:       # hsub(hmul(dall, i2h(sc*q)), hmul(dmin, i2h(m)))  (:117-120)
            return d * (s * q).astype(h) - m[:, None] * np.repeat(mn, 16, axis=1).astype(h)
        if t == Q3_K:       # dl = hmul(d, i2h(sc)); hmul(dl, i2h(q))  (:145, :151)
            return (d * s.astype(h)) * q.astype(h)
        if t in (Q4_K, Q5_K):  # d1 = hmul(dall,i2h(sc)); m1 = hmul(dmin,i2h(m)); hsub(hmul(d1,i2h(q)), m1)
            return (d * s.astype(h)) * q.astype(h) - (m[:, None] * np.repeat(mn, 16, axis=1).astype(h))
        if t == Q6_K:       # hmul(d, i2h(sc * q))  (:250-253)
            return d * (s * q).astype(np.float32).astype(h)
    finally:
        np.seterr(**old)
    raise ValueError(t)


# ---------------------------------------------------------------------------
# Q8_1 activation quantiser (HK/ggml/ggml_kernel.cu:13-50) in numpy, tree sum order
# ---------------------------------------------------------------------------

def quantize_q8_1_groups(x32):
    """x32: float32 [..., 32] -> (q int8 [...,32], d float32 [...], s float32 [...]).
    amax/127, roundf(x/d) (half away from zero), xor-butterfly (16,8,4,2,1) fp32 sum."""
    x32 = np.asarray(x32, np.float32)
    amax = np.max(np.abs(x32), axis=-1)
    s = x32.copy()
    for mask in (16, 8, 4, 2, 1):
        idx = np.arange(32) ^ mask
        s = (s + s[..., idx]).astype(np.float32)
    s = s[..., 0]
    d = (amax / np.float32(127)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (x32 / d[..., None]).astype(np.float32)
    r = np.where(amax[..., None] == 0, np.float32(0), r)
    r64 = r.astype(np.float64)  # |r| + 0.5 is exact in float64
    q = (np.sign(r64) * np.floor(np.abs(r64) + 0.5)).astype(np.int8)
    return q, d, s
