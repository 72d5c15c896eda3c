"""ctypes front-end of oracle/libggq_oracle.so (+ loader for oracle/_ref).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_void_p, c_int, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


def build(verbose=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    out = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libggq_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.oracle_f32_to_f16.restype = ctypes.c_uint16
        L.oracle_f32_to_f16.argtypes = [ctypes.c_float]
        L.oracle_f16_to_f32.restype = ctypes.c_float
        L.oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        for name in ("oracle_dequantize_row_f32", "oracle_dequantize_row_f16", "oracle_dequantize_row_f64"):
            f = getattr(L, name)
            f.restype = c_int
            f.argtypes = [c_int, c_void_p, c_void_p, c_int64]
        L.oracle_quantize_q8_1.restype = None
        L.oracle_quantize_q8_1.argtypes = [c_void_p, c_void_p, c_int64, c_int64]
        L.oracle_quantize_q8_1_mmq.restype = None
        L.oracle_quantize_q8_1_mmq.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int]
        L.oracle_mul_mat_vec_q.restype = c_int
        L.oracle_mul_mat_vec_q.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64]
        L.oracle_mul_mat_q.restype = c_int
        L.oracle_mul_mat_q.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64]
        L.oracle_block_elems.restype = c_int
        L.oracle_block_elems.argtypes = [c_int]
        L.oracle_block_bytes.restype = c_int
        L.oracle_block_bytes.argtypes = [c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_void_p)


def block_elems(t):
    return lib().oracle_block_elems(int(t))


def block_bytes(t):
    return lib().oracle_block_bytes(int(t))


def need_sum(t):
    """mmq_need_sum, HK/ggml/mmq.cu:84-106"""
    return int(t) in (2, 3, 7, 12, 13)


def _w(w):
    return np.ascontiguousarray(np.asarray(w, dtype=np.uint8))


def dequantize_f32(w, t, k):
    """ggml-cpu semantics, legacy formats."""
    w = _w(w)
    y = np.empty(k, np.float32)
    rc = lib().oracle_dequantize_row_f32(int(t), _p(w), _p(y), k)
    if rc:
        raise ValueError(f"type {t} not handled by the reference CPU op")
    return y


def dequantize_f16(w, t, k):
    """GPU semantics; returns float16 array of k elements."""
    w = _w(w)
    y = np.empty(k, np.uint16)
    rc = lib().oracle_dequantize_row_f16(int(t), _p(w), _p(y), k)
    if rc:
        raise ValueError(f"unsupported type {t}")
    return y.view(np.float16)


def dequantize_f64(w, t, k):
    w = _w(w)
    y = np.empty(k, np.float64)
    rc = lib().oracle_dequantize_row_f64(int(t), _p(w), _p(y), k)
    if rc:
        raise ValueError(f"unsupported type {t}")
    return y


def quantize_q8_1(x):
    """x float32 [batch,k] -> uint8 [batch, padded/32*36] (block_q8_1 layout)."""
    x = np.ascontiguousarray(x, np.float32)
    batch, k = x.shape
    padded = (k + 511) // 512 * 512
    q = np.zeros((batch, padded // 32 * 36), np.uint8)
    lib().oracle_quantize_q8_1(_p(x), _p(q), batch, k)
    return q


def quantize_q8_1_mmq(x, t):
    """x float32 [batch,k] -> uint8 [(padded/128)*batch*144] (block_q8_1_mmq layout)."""
    x = np.ascontiguousarray(x, np.float32)
    batch, k = x.shape
    padded = k - k % 512 + 512
    q = np.zeros((padded // 128) * batch * 144, np.uint8)
    lib().oracle_quantize_q8_1_mmq(_p(x), _p(q), batch, k, int(need_sum(t)))
    return q


def mul_mat_vec_q(w, x, t, n_rows):
    """MMVQ semantics: returns (y fp32 [n_rows], yabs fp32 [n_rows]). x float32 [1,k] or [k]."""
    x = np.ascontiguousarray(x, np.float32).reshape(1, -1)
    k = x.shape[1]
    q8 = quantize_q8_1(x)
    w = _w(w)
    y = np.empty(n_rows, np.float32)
    ya = np.empty(n_rows, np.float32)
    rc = lib().oracle_mul_mat_vec_q(int(t), _p(w), _p(q8), _p(y), _p(ya), k, n_rows)
    if rc:
        raise ValueError(f"mul_mat_vec_q: bad type/shape {t} k={k}")
    return y, ya


def mul_mat_q(w, x, t, n_rows):
    """MMQ semantics: returns (y fp32 [batch,n_rows], yabs). x float32 [batch,k]."""
    x = np.ascontiguousarray(x, np.float32)
    batch, k = x.shape
    q8 = quantize_q8_1_mmq(x, t)
    w = _w(w)
    y = np.empty((batch, n_rows), np.float32)
    ya = np.empty((batch, n_rows), np.float32)
    rc = lib().oracle_mul_mat_q(int(t), _p(w), _p(q8), _p(y), _p(ya), batch, k, n_rows)
    if rc:
        raise ValueError(f"mul_mat_q: bad type/shape {t} k={k}")
    return y, ya


def load_reference_cpu_op():
    """The reference's own compiled ggml-cpu op (oracle/_ref), or None when absent."""
    ref_dir = os.path.join(HERE, "_ref")
    if not os.path.isdir(ref_dir):
        return None
    for f in os.listdir(ref_dir):
        if f.startswith("custom_ops") and f.endswith(".so"):
            import torch  # noqa: F401  (the op links libtorch)
            spec = importlib.util.spec_from_file_location("custom_ops", os.path.join(ref_dir, f))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
    return None
