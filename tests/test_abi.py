"""CPU suite: the C-ABI shared libraries load and export every symbol include/ggq.h declares;
host-side traits and argument validation work without a GPU (no kernel is launched here)."""
import os
import re

import pytest

from ggq import lib as ggqlib
from ggq.formats import GGMLType, BLOCK, WEIGHT_TYPES, NEED_SUM, IQ_TYPES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ggq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ggq_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_what_we_bind():
    decl = declared_symbols()
    bound = sorted(set(ggqlib.HIP_SYMBOLS) | set(ggqlib.CPU_SYMBOLS))
    assert decl == bound, f"header vs loader mismatch: {set(decl) ^ set(bound)}"


def test_hip_library_exports_every_symbol():
    L = ggqlib.hip()  # raises if the .so is missing or a symbol is not exported
    for name in ggqlib.HIP_SYMBOLS:
        assert getattr(L, name) is not None
    assert L.ggq_abi_version() == 10


def test_cpu_library_exports_every_symbol():
    L = ggqlib.cpu()
    for name in ggqlib.CPU_SYMBOLS:
        assert getattr(L, name) is not None


def test_traits_match_format_table():
    L = ggqlib.hip()
    for t, (qk, bs) in BLOCK.items():
        assert L.ggq_block_elems(int(t)) == qk and L.ggq_block_bytes(int(t)) == bs
        assert L.ggq_row_bytes(int(t), 4096) == 4096 // qk * bs
    for t in WEIGHT_TYPES:
        assert L.ggq_type_supported(int(t)) == 1 and L.ggq_mmq_type_supported(int(t)) == 1
        assert L.ggq_mmq_need_sum(int(t)) == int(t in NEED_SUM)
    for t in IQ_TYPES:   # dequantise + MMVQ only, like the reference (its ggml_mul_mat_a8 switch has no IQ case)
        assert L.ggq_type_supported(int(t)) == 1 and L.ggq_mmq_type_supported(int(t)) == 0
        assert L.ggq_mmq_tiled_supported(int(t), 4096) == 0
    for bad in (0, 1, 4, 5, 9, 15, 24, 28, 30, -1):
        assert L.ggq_type_supported(bad) == 0 and L.ggq_mmq_type_supported(bad) == 0
    assert L.ggq_row_bytes(int(GGMLType.Q4_K), 100) == -2 and L.ggq_row_bytes(99, 256) == -1


def test_padding_rules_of_the_reference():
    L = ggqlib.hip()
    # MMVQ: roundup(k,512) (ggml_kernel.cu:84); MMQ: k - k%512 + 512 (mmq.cu:190-191), SURVEY §9
    assert [L.ggq_mmvq_padded_k(k) for k in (1, 512, 4096, 11008)] == [512, 512, 4096, 11264]
    assert [L.ggq_mmq_padded_k(k) for k in (4096, 8192, 11008, 100)] == [4608, 8704, 11264, 512]
    assert L.ggq_mmvq_scratch_bytes(4096) == 4096 // 32 * 36
    assert L.ggq_mmq_scratch_bytes(128, 4096) == 128 * 4608 // 32 * 36 == 663552


def test_argument_validation_needs_no_gpu():
    L = ggqlib.hip()
    assert L.ggq_dequantize_f16(None, None, 2, 0, 0, None) == 0          # empty is fine
    assert L.ggq_dequantize_f16(None, None, 5, 1, 32, None) == -1         # unsupported type
    assert L.ggq_dequantize_f16(None, None, 12, 1, 32, None) == -2        # not a multiple of 256
    assert L.ggq_dequantize_f16(None, None, 2, 1, 32, None) == -4         # null pointers
    assert L.ggq_dequantize_f16(None, None, 2, -1, 32, None) == -4
    assert L.ggq_quantize_q8_1(None, 7, None, 1, 32, None) == -3          # bad dtype
    assert L.ggq_quantize_q8_1(None, 1, None, 0, 32, None) == 0
    assert L.ggq_quantize_q8_1_mmq(None, 1, None, 1, 32, 9, None) == -1   # Q8_1 is not a weight type
    assert L.ggq_mul_mat_vec_q(None, None, None, 2, 1, 4096, 8, None, None) == -4
    assert L.ggq_mul_mat_q(None, None, None, 2, 1, 8, 4096, 8, None, None) == -4
    assert L.ggq_mul_mat_q_prequant(None, None, None, 2, 1, 8, 100, 8, 8, None) == -2   # k % 32
    assert L.ggq_mul_mat_q_prequant(None, None, None, 2, 1, 8, 4096, 8, 4, None) == -4  # ldy < n_rows
    assert L.ggq_mul_mat_q_prequant(None, None, None, 2, 1, 0, 4096, 8, 8, None) == 0   # empty batch
    assert L.ggq_mul_mat_vec_q_prequant(None, None, None, 77, 1, 4096, 8, None) == -1
    assert b"unsupported" in L.ggq_strerror(-1)


def test_operator_library_loads_and_registers_the_reference_schemas():
    import torch
    import ggml
    ops = torch.ops._ggml
    for name, schema in (("ggml_dequantize", "_ggml::ggml_dequantize(Tensor W, int type, SymInt m, SymInt n) -> Tensor"),
                         ("ggml_mul_mat_vec_a8", "_ggml::ggml_mul_mat_vec_a8(Tensor W, Tensor X, int type, SymInt row) -> Tensor"),
                         ("ggml_mul_mat_a8", "_ggml::ggml_mul_mat_a8(Tensor W, Tensor X, int type, SymInt row) -> Tensor")):
        assert str(getattr(ops, name).default._schema) == schema
        assert callable(getattr(ggml, name))
    # GPU ops have no CPU fallback: a CPU tensor must be refused loudly
    with pytest.raises((RuntimeError, NotImplementedError)):
        ggml.ggml_dequantize(torch.zeros(18, dtype=torch.uint8), 2, 1, 32)
    with pytest.raises(AssertionError):  # HK/torch-ext/ggml/__init__.py:32-33
        ggml.ggml_mul_mat_vec_a8(torch.zeros(18, dtype=torch.uint8), torch.zeros((2, 32)), 2, 1)


def test_scratch_and_tiled_traits():
    """host logic added with the fragment-major scratch: whole 32-token tiles; which (type, k) the streamed
    kernel accepts (every supported format, whole blocks, rows within its 32-bit byte offsets)"""
    L = ggqlib.hip()
    per_token = (4096 - 4096 % 512 + 512) // 32 * 36
    # ... or, where that is more, the x64 layout's 10240-byte records per (256 elements, 32 tokens) with token tiles in pairs (64 tokens)
    for batch, tiles, tiles64 in ((1, 32, 64), (32, 32, 64), (33, 64, 64), (128, 128, 128), (129, 160, 192)):
        assert L.ggq_mmq_scratch_bytes(batch, 4096) == max(tiles * per_token, tiles64 // 32 * 16 * 10240)
    assert L.ggq_mmq_scratch_bytes(128, 4096) == 663552                        # the reference's figure where it suffices
    assert L.ggq_mmq_scratch_bytes(128, 8192) == 4 * 32 * 10240 > 128 * 8704 // 32 * 36
    assert L.ggq_mmq_scratch_bytes(128, 4096 + 32) == 128 * 4608 // 32 * 36    # k not a multiple of 256: no x64 layout
    for t in WEIGHT_TYPES:
        qk = BLOCK[t][0]
        assert L.ggq_mmq_tiled_supported(int(t), 4096) == 1
        assert L.ggq_mmq_tiled_supported(int(t), 4096 + qk) == 1
        assert L.ggq_mmq_tiled_supported(int(t), 4096 + qk // 2) == 0
        assert L.ggq_mmq_tiled_supported(int(t), 0) == 0
        assert L.ggq_mmq_tiled_supported(int(t), 1 << 28) == 0      # 2^28 elements: rows far beyond 32 MiB
    assert L.ggq_mmq_tiled_supported(int(GGMLType.Q8_1), 4096) == 0
    assert L.ggq_mmq_tiled_supported(99, 4096) == 0
