"""ctypes loader of the C-ABI libraries (include/ggq.h).

The parity tests call the GPU path *through this ABI* (plain pointers + sizes),
exactly what a non-Python host would bind.  Missing library => loud failure.
"""
import ctypes
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIBDIR = os.path.join(_ROOT, "lib")

c_void_p, c_int, c_int64, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/ggq.h declares
HIP_SYMBOLS = {
    "ggq_abi_version": (c_int, []),
    "ggq_strerror": (ctypes.c_char_p, [c_int]),
    "ggq_block_elems": (c_int, [c_int]),
    "ggq_block_bytes": (c_int, [c_int]),
    "ggq_row_bytes": (c_int64, [c_int, c_int64]),
    "ggq_type_supported": (c_int, [c_int]),
    "ggq_mmq_type_supported": (c_int, [c_int]),
    "ggq_mmq_need_sum": (c_int, [c_int]),
    "ggq_mmvq_padded_k": (c_int64, [c_int64]),
    "ggq_mmq_padded_k": (c_int64, [c_int64]),
    "ggq_mmvq_scratch_bytes": (c_size_t, [c_int64]),
    "ggq_mmq_scratch_bytes": (c_size_t, [c_int64, c_int64]),
    "ggq_dequantize_f16": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "ggq_quantize_q8_1": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "ggq_quantize_q8_1_mmq": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ggq_mul_mat_vec_q": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "ggq_mul_mat_q": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "ggq_mul_mat_q_ld": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "ggq_mul_mat_q_prequant": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "ggq_mmq_tiled_supported": (c_int, [c_int, c_int64]),
    "ggq_quantize_q8_1_tiled": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ggq_mul_mat_q_pretiled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "ggq_mul_mat_q_pretiled_epi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "ggq_mul_mat_q_epi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ggq_mmq_route": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_mmq_stream_unit_tokens": (c_int, [c_int, c_int64, c_int64]),
    "ggq_mmq_t16_type_supported": (c_int, [c_int]),
    "ggq_mmq_t16_supported": (c_int, [c_int, c_int64, c_int64]),
    "ggq_mmq_x64_type_supported": (c_int, [c_int]),
    "ggq_mmq_x64_supported": (c_int, [c_int, c_int64, c_int64]),
    "ggq_mmq_x64_k_slices": (c_int, [c_int64, c_int64, c_int64]),
    "ggq_mmq_x64_unit_rows": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_mmq_x64_tile_tokens": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_quantize_q8_1_x64": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ggq_mul_mat_q_x64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "ggq_quantize_q8_1_t16": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ggq_mul_mat_q_t16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "ggq_mul_mat_vec_q_prequant": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p]),
    "ggq_peer_export": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    "ggq_peer_import": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_void_p)]),
    "ggq_peer_close": (c_int, [c_void_p, c_int64]),
    "ggq_peer_write_2d": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "ggq_peer_scatter": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, c_int64, c_int64, c_int64,
                                 ctypes.c_uint32, c_void_p, c_void_p]),
    "ggq_peer_wait": (c_int, [c_void_p, c_int, ctypes.c_uint32, c_void_p, c_void_p]),
    "ggq_mul_mat_vec_q_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_uint32, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p]),
    "ggq_mul_mat_q_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_uint32, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
}
CPU_SYMBOLS = {
    "ggq_mmq_route": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_mmq_stream_unit_tokens": (c_int, [c_int, c_int64, c_int64]),
    "ggq_mmq_x64_supported": (c_int, [c_int, c_int64, c_int64]),
    "ggq_mmq_x64_k_slices": (c_int, [c_int64, c_int64, c_int64]),
    "ggq_mmq_x64_unit_rows": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_mmq_x64_tile_tokens": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "ggq_cpu_dequantize_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int]),
    "ggq_cpu_dequantize_f32_ex": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int]),
    "ggq_cpu_simd_name": (ctypes.c_char_p, []),
    "ggq_cpu_quantize_q8_1_mmq": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int]),
    "ggq_cpu_mul_mat_q": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_int]),
    "ggq_cpu_mmq_simd_name": (ctypes.c_char_p, []),
}

_hip = None
_cpu = None


def _bind(lib, table):
    for name, (res, args) in table.items():
        f = getattr(lib, name)  # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    return lib


def hip_library_path():
    return os.path.join(_LIBDIR, "libggq_hip.so")


def cpu_library_path():
    return os.path.join(_LIBDIR, "libggq_cpu.so")


def hip():
    """libggq_hip.so (import torch first so that both share one HIP runtime instance)."""
    global _hip
    if _hip is None:
        p = hip_library_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing — build it with ggml-libtorch_amd/build.py; no fallback exists")
        import torch  # noqa: F401  (loads libamdhip64 with the SONAME our library needs)
        _hip = _bind(ctypes.CDLL(p), HIP_SYMBOLS)
    return _hip


def cpu():
    global _cpu
    if _cpu is None:
        p = cpu_library_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing — build it with ggml-libtorch_amd/build.py")
        _cpu = _bind(ctypes.CDLL(p), CPU_SYMBOLS)
    return _cpu


DTYPE_CODE = {"float32": 0, "float16": 1, "bfloat16": 2}


def dtype_code(torch_dtype):
    return DTYPE_CODE[str(torch_dtype).replace("torch.", "")]


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what}: ggq error {rc}: {hip().ggq_strerror(rc).decode()}")
