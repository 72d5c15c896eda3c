"""Shared helpers of the parity tests: device buffers via torch, calls through the C ABI."""
import ctypes

import numpy as np
import torch

from ggq import lib as ggqlib
from ggq.formats import GGMLType, BLOCK, NEED_SUM

TORCH_DT = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_bytes(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


def gpu_dequant(w_np, t, m, n):
    L = ggqlib.hip()
    w = dev_bytes(w_np)
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    ggqlib.check(L.ggq_dequantize_f16(vp(w), vp(out), int(t), m, n, stream_ptr()), "ggq_dequantize_f16")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def gpu_quantize_q8_1(x):
    L = ggqlib.hip()
    batch, k = x.shape
    q = torch.zeros(int(L.ggq_mmvq_scratch_bytes(k)) * batch, dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_quantize_q8_1(vp(x), ggqlib.dtype_code(x.dtype), vp(q), batch, k, stream_ptr()), "quantize")
    torch.cuda.synchronize()
    return q.cpu().numpy().reshape(batch, -1)


def gpu_quantize_q8_1_mmq(x, t):
    L = ggqlib.hip()
    batch, k = x.shape
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_quantize_q8_1_mmq(vp(x), ggqlib.dtype_code(x.dtype), vp(q), batch, k, int(t), stream_ptr()),
                 "quantize_mmq")
    torch.cuda.synchronize()
    return q.cpu().numpy()


def gpu_quantize_q8_1_tiled(x, t):
    L = ggqlib.hip()
    batch, k = x.shape
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_quantize_q8_1_tiled(vp(x), ggqlib.dtype_code(x.dtype), vp(q), batch, k, int(t), stream_ptr()),
                 "quantize_tiled")
    torch.cuda.synchronize()
    return q.cpu().numpy()


def retile_q8_1_mmq(q_mmq, batch, k):
    """block_q8_1_mmq bytes (index (k/128)*batch + token) -> the fragment-major layout of
    ggq_quantize_q8_1_tiled, built independently in numpy: per (k/128, token/32) a 4608-byte tile
    { qs[4 groups][2 halves][32 tokens][16]; ds[2 pairs][32 tokens][2 groups][4 bytes] }."""
    padded = k - k % 512 + 512
    n_kb, n_tt = padded // 128, (batch + 31) // 32
    blocks = np.asarray(q_mmq, np.uint8)[:n_kb * batch * 144].reshape(n_kb, batch, 144)
    out = np.zeros((n_kb, n_tt, 4608), np.uint8)
    for t in range(batch):
        tt, tl = divmod(t, 32)
        ds = blocks[:, t, :16].reshape(n_kb, 4, 4)          # [kb][group][4 bytes]
        qs = blocks[:, t, 16:].reshape(n_kb, 4, 2, 16)      # [kb][group][half][16]
        for g in range(4):
            for h in range(2):
                o = g * 1024 + h * 512 + tl * 16
                out[:, tt, o:o + 16] = qs[:, g, h]
            o = 4096 + (g >> 1) * 256 + tl * 8 + (g & 1) * 4
            out[:, tt, o:o + 4] = ds[:, g]
    return out, n_tt


def gpu_mmq_pretiled(w_np, x, t, n_rows, ldy=None):
    """quantise once into the fragment-major scratch, then the streamed kernel alone"""
    L = ggqlib.hip()
    batch, k = x.shape
    ldy = n_rows if ldy is None else ldy
    w = dev_bytes(w_np)
    y = torch.zeros((batch, ldy), dtype=x.dtype, device="cuda")
    q = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    dt = ggqlib.dtype_code(x.dtype)
    ggqlib.check(L.ggq_quantize_q8_1_tiled(vp(x), dt, vp(q), batch, k, int(t), stream_ptr()), "quantize_tiled")
    ggqlib.check(L.ggq_mul_mat_q_pretiled(vp(w), vp(q), vp(y), int(t), dt, batch, k, n_rows, ldy, stream_ptr()),
                 "ggq_mul_mat_q_pretiled")
    torch.cuda.synchronize()
    return y


def gpu_mmvq(w_np, x, t, n_rows):
    L = ggqlib.hip()
    k = x.shape[1]
    w = dev_bytes(w_np)
    y = torch.empty((1, n_rows), dtype=x.dtype, device="cuda")
    scratch = torch.empty(int(L.ggq_mmvq_scratch_bytes(k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_mul_mat_vec_q(vp(w), vp(x), vp(y), int(t), ggqlib.dtype_code(x.dtype), k, n_rows,
                                     vp(scratch), stream_ptr()), "ggq_mul_mat_vec_q")
    torch.cuda.synchronize()
    return y


def gpu_mmq(w_np, x, t, n_rows):
    L = ggqlib.hip()
    batch, k = x.shape
    w = dev_bytes(w_np)
    y = torch.empty((batch, n_rows), dtype=x.dtype, device="cuda")
    scratch = torch.empty(max(16, int(L.ggq_mmq_scratch_bytes(batch, k))), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_mul_mat_q(vp(w), vp(x), vp(y), int(t), ggqlib.dtype_code(x.dtype), batch, k, n_rows,
                                 vp(scratch), stream_ptr()), "ggq_mul_mat_q")
    torch.cuda.synchronize()
    return y


def same_nan(a, b):
    """bit-level equality of fp16 arrays where any NaN equals any NaN"""
    a = np.asarray(a, np.float16)
    b = np.asarray(b, np.float16)
    an, bn = np.isnan(a), np.isnan(b)
    if not np.array_equal(an, bn):
        return False
    return np.array_equal(a.view(np.uint16)[~an], b.view(np.uint16)[~bn])


DT_EPS = {torch.float32: 0.0, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def assert_fp_accumulate(y_gpu, y_ref, yabs, dtype, what=""):
    """The north-star tolerance for the fp accumulate: 1e-3 relative.  y_ref is the oracle's
    fp32 value before the cast to the output dtype; yabs = Σ|terms| scales the slack that fp32
    re-association may cause when terms cancel (1e-5 of it), and one rounding step of the
    output dtype is granted because both sides round a slightly different fp32 number."""
    y = y_gpu.detach().float().cpu().numpy().reshape(y_ref.shape).astype(np.float64)
    ref = y_ref.astype(np.float64)
    tol = 1e-3 * np.abs(ref) + DT_EPS[dtype] * np.abs(ref) + 1e-5 * yabs.astype(np.float64) + 1e-30
    bad = np.abs(y - ref) > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} outside 1e-3 rel; worst "
                           f"{np.max(np.abs(y - ref) / tol):.2f}x tol at {np.unravel_index(np.argmax(np.abs(y - ref) / tol), ref.shape)}")


# ---------------------------------------------------------------- 16-token-tile path (mmq_t16.hip)
def t16_perm(t):
    """the 16-byte slot 4 f + c of a tile (fragment f, K-chunk c) -> the two 8-element runs of the 256-element unit it holds, in
    order; written from the weight formats' bit layouts (HK/ggml/ggml-common.h:17-108) and from which raw bytes the weight
    lane (row, chunk c) of MFMA f reads, independently of quantize.hip's tables"""
    t = GGMLType(int(t))
    perm = []
    for f in range(4):
        for c in range(4):
            if t in (GGMLType.Q4_K, GGMLType.Q5_K):
                q, hi = divmod(f, 2)                 # 64-byte half of the nibble field, low / high nibbles
                pair, half = 2 * q + (c >> 1), c & 1   # 32-byte chunk of the pair, its 16-byte half
                e0 = 64 * pair + 32 * hi + 16 * half
                perm.append((e0 // 8, e0 // 8 + 1))
            elif t == GGMLType.Q8_0:                  # the lane's 16 bytes are 16 consecutive elements
                perm.append((2 * (4 * f + c), 2 * (4 * f + c) + 1))
            elif t == GGMLType.Q3_K:                  # element 128 ip + 32 jq + l <- bits 2 jq of qs[32 ip + l]: operand f = 2 ip + m, lane chunk c
                ip, m = divmod(f, 2)                  # reads qs[32 ip + 16 (c & 1) ..] and takes jq = 2 m + (c >> 1)
                e0 = 128 * ip + 32 * (2 * m + (c >> 1)) + 16 * (c & 1)
                perm.append((e0 // 8, e0 // 8 + 1))
            elif t == GGMLType.Q6_K:                  # element 128 ip + 32 jq + l <- nibble (jq >> 1) of ql[64 ip + 32 (jq & 1) + l]:
                ip, hi = divmod(f, 2)                 # operand f = 2 ip + hi, lane chunk c reads ql[64 ip + 16 c ..]: jq & 1 = c >> 1,
                jq, l0 = 2 * hi + (c >> 1), 16 * (c & 1)   # l = 16 (c & 1) .., and the nibble chosen gives jq >> 1 = hi
                e0 = 128 * ip + 32 * jq + l0
                perm.append((e0 // 8, e0 // 8 + 1))
            else:   # 32-element nibble blocks: the lane reads qs[8 h .. 8 h + 7] of block 2 f + (c >> 1): low nibbles = elements
                    # 8 h .., high nibbles = elements 16 + 8 h ..
                b, h = 2 * f + (c >> 1), c & 1
                perm.append(((32 * b + 8 * h) // 8, (32 * b + 16 + 8 * h) // 8))
    assert sorted(r for p in perm for r in p) == list(range(32))
    return perm


def retile_q8_1_t16(q_mmq, batch, k, t):
    """block_q8_1_mmq bytes (index (k/128)*batch + token) -> the 16-token-tile layout of ggq_quantize_q8_1_t16, built in
    numpy: per (k/256, token/16) a 4608-byte tile { frag[4][4 K-chunks][16 tokens][16]; ds[2][4 quads][4 groups][4 tokens] }"""
    padded = k - k % 512 + 512
    n_kb, n_u, n_tt = padded // 128, padded // 256, (batch + 15) // 16
    blocks = np.asarray(q_mmq, np.uint8)[:n_kb * batch * 144].reshape(n_kb, batch, 144)
    perm = t16_perm(t)
    if batch <= 8:   # the 8-token form: 2304-byte tiles { frag[4][4][8 tokens][16]; ds[2][2 quads][4 groups][4 tokens] }
        out = np.zeros((n_u, 2304), np.uint8)
        for tok in range(batch):
            qs = blocks[:, tok, 16:].reshape(n_u, 256)
            ds = blocks[:, tok, :16].reshape(n_u, 8, 4)
            for slot, (v0, v1) in enumerate(perm):
                f, c = divmod(slot, 4)
                o = f * 512 + (c * 8 + tok) * 16
                out[:, o:o + 8] = qs[:, 8 * v0:8 * v0 + 8]
                out[:, o + 8:o + 16] = qs[:, 8 * v1:8 * v1 + 8]
            for g8 in range(8):
                o = 2048 + ((((g8 >> 2) * 2 + (tok >> 2)) * 4 + (g8 & 3)) * 4 + (tok & 3)) * 4
                out[:, o:o + 4] = ds[:, g8]
        return out
    out = np.zeros((n_u, n_tt, 4608), np.uint8)
    for tok in range(batch):
        tt, tl = divmod(tok, 16)
        qs = blocks[:, tok, 16:].reshape(n_u, 256)          # the token's int8 values, unit by unit
        ds = blocks[:, tok, :16].reshape(n_u, 8, 4)         # [unit][group of the unit][4 bytes]
        for slot, (v0, v1) in enumerate(perm):
            f, c = divmod(slot, 4)
            o = f * 1024 + (c * 16 + tl) * 16
            out[:, tt, o:o + 8] = qs[:, 8 * v0:8 * v0 + 8]
            out[:, tt, o + 8:o + 16] = qs[:, 8 * v1:8 * v1 + 8]
        for g8 in range(8):
            o = 4096 + ((((g8 >> 2) * 4 + (tl >> 2)) * 4 + (g8 & 3)) * 4 + (tl & 3)) * 4
            out[:, tt, o:o + 4] = ds[:, g8]
    return out


def gpu_quantize_q8_1_t16(x, t):
    L = ggqlib.hip()
    batch, k = x.shape
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_quantize_q8_1_t16(vp(x), ggqlib.dtype_code(x.dtype), vp(q), batch, k, int(t), stream_ptr()), "quantize_t16")
    torch.cuda.synchronize()
    return q.cpu().numpy()


def gpu_mmq_t16(w_np, x, t, n_rows, ldy=None, epilogue=0, aux=None, w_dev=None):
    """quantise into the 16-token-tile scratch, then the 16-token-tile kernel alone"""
    L = ggqlib.hip()
    batch, k = x.shape
    ldy = n_rows if ldy is None else ldy
    w = dev_bytes(w_np) if w_dev is None else w_dev
    y = torch.zeros((batch, ldy), dtype=x.dtype, device="cuda")
    q = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    dt = ggqlib.dtype_code(x.dtype)
    ggqlib.check(L.ggq_quantize_q8_1_t16(vp(x), dt, vp(q), batch, k, int(t), stream_ptr()), "quantize_t16")
    ggqlib.check(L.ggq_mul_mat_q_t16(vp(w), vp(q), vp(y), int(t), dt, batch, k, n_rows, ldy, epilogue,
                                     None if aux is None else vp(aux), stream_ptr()), "ggq_mul_mat_q_t16")
    torch.cuda.synchronize()
    return y


# ---------------------------------------------------------------- 64 x 64 wave-tile path (mmq_x64.hip)
X64_REC = 10240


def retile_q8_1_x64(q_mmq, batch, k, t):
    """block_q8_1_mmq bytes (index (k/128)*batch + token) -> the x64 layout of ggq_quantize_q8_1_x64, built in numpy from the layout's
    definition (include/ggq.h): per (k/256, token/32) a 10240-byte record { frag[8 groups][2 K-halves][32 tokens][16]; float d8[8 groups][h][qd][e]
    (the fp16 d as fp32 for the need_sum formats, else the fp32 d; token 8 qd + 4 h + e); fp16 s8[2 kh][32 tokens][8] at byte 9216 (need_sum formats) };
    the token tiles of one K step are contiguous and padded to an even count.  Returns (records, mask of the bytes the layout defines)."""
    from ggq.formats import GGMLType as G
    need_sum = G(int(t)) in (G.Q4_0, G.Q4_1, G.Q5_1, G.Q4_K, G.Q5_K)
    assert k % 256 == 0
    n_kb, n_sb, n_tt = k // 128, k // 256, (batch + 63) // 64 * 2
    blocks = np.asarray(q_mmq, np.uint8)[:n_kb * batch * 144].reshape(n_kb, batch, 144)
    out = np.zeros((n_sb, n_tt, X64_REC), np.uint8)
    mask = np.zeros((n_sb, n_tt, X64_REC), bool)
    for tok in range(batch):
        tt, tl = divmod(tok, 32)
        qs = blocks[:, tok, 16:].reshape(n_sb, 8, 2, 16)     # [super-block][group][K-half][16]
        ds = blocks[:, tok, :16].reshape(n_sb, 8, 4)         # [super-block][group][4 bytes]: half2(d, sum) or float d
        h, qd, e = (tl >> 2) & 1, tl >> 3, tl & 3
        idx = h * 16 + qd * 4 + e
        for g8 in range(8):
            for kh in range(2):
                o = g8 * 1024 + kh * 512 + tl * 16
                out[:, tt, o:o + 16] = qs[:, g8, kh]
                mask[:, tt, o:o + 16] = True
            if need_sum:
                o = 8192 + g8 * 128 + idx * 4
                out[:, tt, o:o + 4] = np.ascontiguousarray(ds[:, g8, 0:2]).view(np.float16).astype(np.float32).view(np.uint8)
                mask[:, tt, o:o + 4] = True
                j = g8 & 3
                p0 = (j >> 1) * 4 + (j & 1)
                for p in (p0, p0 + 2):
                    o = 9216 + ((g8 >> 2) * 32 + tl) * 16 + 2 * p
                    out[:, tt, o:o + 2] = ds[:, g8, 2:4]
                    mask[:, tt, o:o + 2] = True
            else:
                o = 8192 + g8 * 128 + idx * 4
                out[:, tt, o:o + 4] = ds[:, g8]
                mask[:, tt, o:o + 4] = True
    return out, mask


def gpu_quantize_q8_1_x64(x, t):
    L = ggqlib.hip()
    batch, k = x.shape
    q = torch.zeros(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    ggqlib.check(L.ggq_quantize_q8_1_x64(vp(x), ggqlib.dtype_code(x.dtype), vp(q), batch, k, int(t), stream_ptr()), "quantize_x64")
    torch.cuda.synchronize()
    return q.cpu().numpy()


def gpu_mmq_x64(w_np, x, t, n_rows, ldy=None, epilogue=0, aux=None, w_dev=None):
    """quantise into the x64 scratch, then the 64 x 64 wave-tile kernel alone"""
    L = ggqlib.hip()
    batch, k = x.shape
    ldy = n_rows if ldy is None else ldy
    w = dev_bytes(w_np) if w_dev is None else w_dev
    y = torch.zeros((batch, ldy), dtype=x.dtype, device="cuda")
    q = torch.empty(int(L.ggq_mmq_scratch_bytes(batch, k)), dtype=torch.uint8, device="cuda")
    dt = ggqlib.dtype_code(x.dtype)
    ggqlib.check(L.ggq_quantize_q8_1_x64(vp(x), dt, vp(q), batch, k, int(t), stream_ptr()), "quantize_x64")
    ggqlib.check(L.ggq_mul_mat_q_x64(vp(w), vp(q), vp(y), int(t), dt, batch, k, n_rows, ldy, epilogue,
                                     None if aux is None else vp(aux), stream_ptr()), "ggq_mul_mat_q_x64")
    torch.cuda.synchronize()
    return y
