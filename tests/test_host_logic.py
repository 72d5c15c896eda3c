"""CPU suite: host-side helpers (format table, synthetic inputs, build recipe, oracle isolation)."""
import os
import re

import numpy as np

from ggq import synth
from ggq.formats import GGMLType, BLOCK, WEIGHT_TYPES, row_bytes, weight_bytes
from oracle import ggq_numpy as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baseline_byte_counts():
    # BASELINE.md §3 algorithmic bytes at 4096 x 11008
    assert weight_bytes(GGMLType.Q4_0, 11008, 4096) == 25362432 == weight_bytes(GGMLType.Q4_K, 11008, 4096)
    assert weight_bytes(GGMLType.Q8_0, 11008, 4096) == 47906816
    assert weight_bytes(GGMLType.Q5_K, 11008, 4096) == 30998528
    assert weight_bytes(GGMLType.Q6_K, 11008, 4096) == 36986880
    assert weight_bytes(GGMLType.Q4_K, 11008, 4096) + 128 * 4096 * 2 + 128 * 11008 * 2 == 29229056
    assert row_bytes(GGMLType.Q4_0, 11008) == 6192


def test_synthetic_blocks_are_valid_and_finite():
    for t in WEIGHT_TYPES:
        b = synth.random_blocks(t, 200, seed=1)
        assert b.shape == (200, BLOCK[t][1]) and b.dtype == np.uint8
        w = N.dequantize_exact(b, t)
        assert np.isfinite(w).all() and np.abs(w).max() < 64
        assert np.array_equal(b, synth.random_blocks(t, 200, seed=1))  # seeded
        assert synth.edge_blocks(t).shape[1] == BLOCK[t][1]


def test_product_path_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    pkg = os.path.join(ROOT, "ggml-libtorch_amd")
    for dp, _, files in os.walk(pkg):
        if "_build" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f"{f} imports oracle"
                assert "ggq_oracle" not in txt and "libggq_oracle" not in txt, f"{f} references the oracle library"


def test_no_compat_layers_in_kernels():
    """no hipify output, no CUDA shims, no dual paths (north_star)"""
    hipdir = os.path.join(ROOT, "ggml-libtorch_amd", "csrc", "hip")
    for f in os.listdir(hipdir):
        txt = open(os.path.join(hipdir, f)).read()
        for banned in ("__HIP_PLATFORM_AMD__", "USE_ROCM", "cuda_runtime", "__CUDA_ARCH__", "hipify", "triton"):
            assert banned not in txt, f"{f} contains {banned}"
