"""Loads the compiled operator library and exposes it as ``ops`` — the role the
kernel-builder-generated ``_ops.py`` plays in the reference package
(HK/torch-ext/ggml/__init__.py:3-12).  Fails loudly when the extension has not been
built: there is no eager / CPU fallback for the GPU ops."""
import glob
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    cands = sorted(glob.glob(os.path.join(_HERE, "_ggml*.so")))
    if not cands:
        raise ImportError(
            "ggml: the native extension ggml/_ggml*.so is missing — run "
            "`python ggml-libtorch_amd/build.py` (or __graft_entry__.build()) first; "
            "there is no fallback path for the gfx950 kernels")
    torch.ops.load_library(cands[0])
    return cands[0]


LIBRARY_PATH = _load()
ops = torch.ops._ggml


def add_op_namespace_prefix(op_name: str) -> str:
    return f"_ggml::{op_name}"
