"""Synthetic block-valid quantised weights (the measurement/test input recipe).

Recipe: quant / scale / qh / hmask bytes uniform u8; fp16 block scale
d = fp16(U(0.5,2) * 2^-8); m / dmin = fp16(U(-1,1) * 2^-6); Q6_K sub-scales int8 uniform
[-64,63].  All finite, no NaN/Inf/subnormal.  The reference reads its inputs from GGUF
sample files (benchmarks/utils.py:25-31) that are not available offline.
"""
import numpy as np

from .formats import GGMLType, BLOCK

# byte offsets of the fp16 fields, per format: (d_off, m_off or None)
_F16_FIELDS = {
    GGMLType.Q4_0: (0, None), GGMLType.Q4_1: (0, 2), GGMLType.Q5_0: (0, None), GGMLType.Q5_1: (0, 2),
    GGMLType.Q8_0: (0, None), GGMLType.Q2_K: (80, 82), GGMLType.Q3_K: (108, None),
    GGMLType.Q4_K: (0, 2), GGMLType.Q5_K: (0, 2), GGMLType.Q6_K: (208, None),
    GGMLType.IQ4_NL: (0, None), GGMLType.IQ4_XS: (0, None),
    GGMLType.IQ2_XXS: (0, None), GGMLType.IQ2_XS: (0, None), GGMLType.IQ2_S: (0, None), GGMLType.IQ3_XXS: (0, None),
    GGMLType.IQ3_S: (0, None), GGMLType.IQ1_S: (0, None),
    GGMLType.IQ1_M: (None, None),   # no fp16 field: the super-block scale is scattered over the high nibbles of scales[]
}


def _iq1m_put_scale(b, d16):
    """IQ1_M: write the fp16 super-block scale `d16` [n] into bits 12-15 of the four uint16 of scales[] (block bytes 48..55),
    nibble k of the value into word k (HK/ggml/dequantize.cuh:481-482 reads it back as iq1m_scale_t)"""
    u = d16.view(np.uint16).astype(np.uint16)
    for k in range(4):
        nib = ((u >> (4 * k)) & 0xF).astype(np.uint8)
        b[:, 49 + 2 * k] = (b[:, 49 + 2 * k] & 0x0F) | (nib << 4)



def random_blocks(t, n_blocks, seed=0, d_scale=1.0):
    """uint8 [n_blocks, block_bytes] of valid blocks.  d_scale multiplies the fp16 block
    scales (d, m/dmin) — the K-quant recipe yields |w| up to ~16, real checkpoints are ~1e-2."""
    t = GGMLType(int(t))
    _, bs = BLOCK[t]
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, size=(n_blocks, bs), dtype=np.uint8)
    d_off, m_off = _F16_FIELDS[t]
    d = (rng.uniform(0.5, 2.0, n_blocks) * 2.0 ** -8 * d_scale).astype(np.float16)
    if d_off is None:
        _iq1m_put_scale(b, d)
    else:
        b[:, d_off:d_off + 2] = d.view(np.uint8).reshape(n_blocks, 2)
    if m_off is not None:
        m = (rng.uniform(-1.0, 1.0, n_blocks) * 2.0 ** -6 * d_scale).astype(np.float16)
        b[:, m_off:m_off + 2] = m.view(np.uint8).reshape(n_blocks, 2)
    if t == GGMLType.Q6_K:
        sc = rng.integers(-64, 64, size=(n_blocks, 16), dtype=np.int8)
        b[:, 192:208] = sc.view(np.uint8)
    return b


def random_weight(t, n_rows, k, seed=0, d_scale=1.0):
    """uint8 [n_rows, row_bytes] — the shape a GGUF ReaderTensor.data has."""
    qk, bs = BLOCK[GGMLType(int(t))]
    assert k % qk == 0
    return random_blocks(t, n_rows * (k // qk), seed, d_scale).reshape(n_rows, (k // qk) * bs)


def edge_blocks(t):
    """Hand-made corner blocks: all-zero / all-ones payloads x special fp16 scales."""
    t = GGMLType(int(t))
    _, bs = BLOCK[t]
    d_off, m_off = _F16_FIELDS[t]
    specials = np.array([0.0, -0.0, 1.0, -1.0, 6e-8, -6e-8, 65504.0, -65504.0, 0.333251953125],
                        dtype=np.float16)
    out = []
    for fill in (0x00, 0xFF, 0xAA, 0x55, 0x0F, 0xF0, 0x80, 0x7F):
        for d in specials:
            for m in (specials[[0, 2, 3, 8]] if m_off is not None else [None]):
                blk = np.full(bs, fill, np.uint8)
                if d_off is None:
                    _iq1m_put_scale(blk.reshape(1, -1), np.array([d], np.float16))
                else:
                    blk[d_off:d_off + 2] = np.array([d], np.float16).view(np.uint8)
                if m is not None:
                    blk[m_off:m_off + 2] = np.array([m], np.float16).view(np.uint8)
                out.append(blk)
    return np.stack(out)
