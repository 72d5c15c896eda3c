#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run in the BUILD container only
(/root/reference present):  python tests/golden/gen_golden.py

reference_cpu_dequant.npz   inputs (block bytes) + fp32 outputs of the REFERENCE's own compiled
                            ggml-cpu op (oracle/_ref, built from /root/reference/ggml-cpu by
                            oracle/Makefile) for Q4_0 Q4_1 Q5_0 Q5_1 Q8_0 — random + edge blocks.
                            This is what pins the oracle (tests/test_oracle_golden.py).
oracle_pins.npz             outputs of OUR oracle for what no reference executable can produce
                            here (fp16 GPU-semantics dequantise, K-quants, Q8_1 quantiser, MMVQ,
                            MMQ).  Regression pins only — parity for these is "unpinned"
                            (DESIGN.md §Oracle); they are cross-checked against the independent
                            numpy derivation at generation time.
A fixture is data: inputs and expected outputs.  No reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ggml-libtorch_amd"))

import torch  # noqa: E402
from oracle import oracle as O, ggq_numpy as N  # noqa: E402
from ggq import synth  # noqa: E402
from ggq.formats import GGMLType, BLOCK, WEIGHT_TYPES  # noqa: E402

LEGACY = [GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0]


def main():
    O.build()
    ref = O.load_reference_cpu_op()
    assert ref is not None, "oracle/_ref missing: /root/reference not available?"
    out = {}
    for t in LEGACY:
        qk, bs = BLOCK[t]
        blocks = np.concatenate([synth.random_blocks(t, 64, seed=100 + int(t)), synth.edge_blocks(t)])
        k = blocks.shape[0] * qk
        y = ref.ggml_dequantize(torch.from_numpy(blocks.reshape(1, -1).copy()), int(t), 1, k).numpy().reshape(-1)
        out[f"{t.name}_blocks"] = blocks
        out[f"{t.name}_f32_bits"] = y.view(np.uint32)  # bit patterns (NaN-safe)
    np.savez_compressed(os.path.join(HERE, "reference_cpu_dequant.npz"), **out)

    pins = {}
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 512)).astype(np.float32)
    for t in WEIGHT_TYPES:
        qk, bs = BLOCK[t]
        blocks = np.concatenate([synth.random_blocks(t, 24, seed=200 + int(t)), synth.edge_blocks(t)[:40]])
        k = blocks.shape[0] * qk
        f16 = O.dequantize_f16(blocks, t, k)
        assert np.array_equal(f16.astype(np.float32), N.dequantize_f16(blocks, t).reshape(-1).astype(np.float32),
                              equal_nan=True), t
        pins[f"{t.name}_blocks"] = blocks
        pins[f"{t.name}_f16_bits"] = f16.view(np.uint16)
        # small matmuls: K = 512, N = 12, batch {1, 5}
        w = synth.random_weight(t, 12, 512, seed=300 + int(t))
        yv, _ = O.mul_mat_vec_q(w, x[:1], t, 12)
        ym, _ = O.mul_mat_q(w, x, t, 12)
        pins[f"{t.name}_mm_w"] = w
        pins[f"{t.name}_mmvq_y"] = yv
        pins[f"{t.name}_mmq_y"] = ym
    pins["mm_x"] = x
    xq = np.concatenate([rng.standard_normal((3, 96)).astype(np.float32), np.zeros((1, 96), np.float32)])
    xq[2, :32] = np.arange(32) - 15.5
    pins["q8_x"] = xq
    pins["q8_1_bytes"] = O.quantize_q8_1(xq)
    pins["q8_1_mmq_sum_bytes"] = O.quantize_q8_1_mmq(xq, GGMLType.Q4_K)
    pins["q8_1_mmq_nosum_bytes"] = O.quantize_q8_1_mmq(xq, GGMLType.Q8_0)
    np.savez_compressed(os.path.join(HERE, "oracle_pins.npz"), **pins)
    for f in ("reference_cpu_dequant.npz", "oracle_pins.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


def iq_pins():
    """oracle_pins_iq.npz: the oracle's outputs for the nine IQ formats (dequantise fp16 bits, MMVQ results) — frozen so that
    a later edit of the oracle or of the codebook tables cannot move both sides of the GPU parity tests together.
    `python tests/golden/gen_golden.py iq` (needs no reference build)."""
    from ggq.formats import IQ_TYPES
    O.build()
    pins = {}
    x = np.random.default_rng(11).standard_normal((1, 512)).astype(np.float32)
    for t in IQ_TYPES:
        qk, bs = BLOCK[t]
        blocks = np.concatenate([synth.random_blocks(t, 16, seed=400 + int(t)), synth.edge_blocks(t)[:24]])
        f16 = O.dequantize_f16(blocks, t, blocks.shape[0] * qk)
        assert np.array_equal(f16.astype(np.float32), N.dequantize_f16(blocks, t).reshape(-1).astype(np.float32), equal_nan=True), t
        w = synth.random_weight(t, 12, 512, seed=500 + int(t))
        yv, _ = O.mul_mat_vec_q(w, x, t, 12)
        pins[f"{t.name}_blocks"], pins[f"{t.name}_f16_bits"] = blocks, f16.view(np.uint16)
        pins[f"{t.name}_mm_w"], pins[f"{t.name}_mmvq_y"] = w, yv
    pins["mm_x"] = x
    np.savez_compressed(os.path.join(HERE, "oracle_pins_iq.npz"), **pins)
    print("oracle_pins_iq.npz", os.path.getsize(os.path.join(HERE, "oracle_pins_iq.npz")), "bytes")


if __name__ == "__main__":
    iq_pins() if sys.argv[1:] == ["iq"] else main()
