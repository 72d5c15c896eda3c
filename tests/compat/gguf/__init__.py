"""TEST INFRASTRUCTURE — not part of the product, never on sys.path unless a test puts it there.

A stand-in for the four names the reference's harness imports from the third-party ``gguf`` package (gguf-py, not installed in
this image and not vendored by the reference):

    from gguf import GGMLQuantizationType, GGUFReader, ReaderTensor, dequantize
    (reference: tests/test_dequantize.py:6, benchmarks/utils.py:22, hf-kernels/ggml-kernels/tests/kernels/test_cuda_kernels.py:3)

so that a test written exactly as the reference writes its own — that import line, ``import custom_ops as ops`` /
``import ggml as ops``, the same calls and tolerances — runs against this build.  ``GGMLQuantizationType`` is the product's own
enum (same names and ids as ggml's), ``GGUFReader`` / ``ReaderTensor`` the product's GGUF reader (ggq/gguf_io.py); ``dequantize``
is the CHECKER: the numpy restatement of gguf-py's ``quants.dequantize`` under oracle/ (parity unpinned beyond the legacy formats:
no gguf-py output is held by the reference).  tests/test_reference_harness.py uses it only when the real package is absent."""
import numpy as np

from ggq.formats import GGMLType as GGMLQuantizationType   # noqa: F401
from ggq.gguf_io import GGUFReader, ReaderTensor           # noqa: F401
from oracle import ggq_numpy as _N


def dequantize(data, qtype):
    """gguf-py's ``dequantize(data, qtype)``: uint8 ``[..., bytes_per_row]`` -> float32 ``[..., elements_per_row]``"""
    data = np.asarray(data, dtype=np.uint8)
    out = _N.gguf_dequantize(data, GGMLQuantizationType(int(qtype)))
    return np.ascontiguousarray(out, dtype=np.float32).reshape(*data.shape[:-1], -1)
