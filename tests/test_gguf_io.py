"""GGUF container reader/writer: the file format on the input side of the hot path.
The reader is checked against a file assembled here field by field with struct (independent of the writer),
then writer -> reader round trips, then the reference's own test protocol (tests/test_dequantize.py:60-75:
read every tensor of Quant_{TYPE}_{hidden}.gguf, dequantise with the op, compare with a ground truth)
replayed on synthetic sample files with the CPU op; the GPU ops replay lives in test_gpu_ops.py."""
import struct

import numpy as np
import pytest

from ggq import gguf_io, synth
from ggq.formats import BLOCK, GGMLType, WEIGHT_TYPES, row_bytes


def _s(x):
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def test_reader_on_hand_assembled_file(tmp_path):
    rows, cols = 3, 64
    w = synth.random_weight(GGMLType.Q4_0, rows, cols, seed=1)
    f32 = np.arange(8, dtype=np.float32)
    head = struct.pack("<IIQQ", 0x46554747, 3, 2, 3)
    head += _s("general.architecture") + struct.pack("<I", 8) + _s("llama")
    head += _s("general.alignment") + struct.pack("<II", 4, 64)
    head += _s("tokenizer.list") + struct.pack("<IIQ", 9, 8, 2) + _s("a") + _s("bc")
    infos = _s("blk.0.w_3x64") + struct.pack("<IQQIQ", 2, cols, rows, 2, 0)
    infos += _s("bias") + struct.pack("<IQIQ", 1, 8, 0, 128)   # second tensor at the next 64-byte boundary
    blob = head + infos
    blob += b"\0" * (-len(blob) % 64)
    data = bytearray(192)
    data[:w.size] = w.tobytes()
    data[128:128 + 32] = f32.tobytes()
    p = tmp_path / "hand.gguf"
    p.write_bytes(blob + bytes(data))
    r = gguf_io.GGUFReader(p)
    assert r.version == 3 and r.alignment == 64
    assert r.fields["general.architecture"] == "llama" and r.fields["tokenizer.list"] == ["a", "bc"]
    t0, t1 = r.tensors
    assert t0.name == "blk.0.w_3x64" and t0.tensor_type == GGMLType.Q4_0 and t0.shape == (cols, rows)
    assert t0.data.dtype == np.uint8 and t0.data.shape == (rows, row_bytes(GGMLType.Q4_0, cols))
    assert np.array_equal(t0.data, w)
    assert t1.shape == (8,) and np.array_equal(t1.data, f32)
    assert r.get_tensor("bias") is t1


@pytest.mark.parametrize("t", WEIGHT_TYPES, ids=lambda t: t.name)
def test_write_read_round_trip(tmp_path, t):
    hidden = 256
    path = gguf_io.write_sample_file(tmp_path, t, hidden, seed=3)
    assert path.endswith(f"Quant_{t.name}_{hidden}.gguf")
    r = gguf_io.GGUFReader(path)
    assert len(r.tensors) == 3
    for i, tensor in enumerate(r.tensors):
        m, n = map(int, tensor.name.split("_")[-1].split("x"))   # the reference's shape convention
        assert (m, n) == (hidden * (i + 1), hidden) and tensor.shape == (n, m)
        assert tensor.tensor_type == t and int(tensor.tensor_type) == int(t)
        assert tensor.data_offset % 32 == 0
        assert np.array_equal(tensor.data, synth.random_weight(t, m, n, seed=3 + i))


def test_reader_rejects_malformed_files(tmp_path):
    p = tmp_path / "bad.gguf"
    p.write_bytes(b"GGML" + b"\0" * 40)
    with pytest.raises(ValueError, match="magic"):
        gguf_io.GGUFReader(p)
    p.write_bytes(struct.pack("<IIQQ", 0x46554747, 1, 0, 0))
    with pytest.raises(ValueError, match="version"):
        gguf_io.GGUFReader(p)
    # tensor data running past the end of the file
    head = struct.pack("<IIQQ", 0x46554747, 3, 1, 0) + _s("w") + struct.pack("<IQQIQ", 2, 32, 4, 2, 0)
    p.write_bytes(head + b"\0" * (-len(head) % 32) + b"\0" * 10)
    with pytest.raises(ValueError, match="out of bounds"):
        gguf_io.GGUFReader(p)
    # unsupported tensor type (IQ formats are out of scope)
    head = struct.pack("<IIQQ", 0x46554747, 3, 1, 0) + _s("w") + struct.pack("<IQQIQ", 2, 256, 1, 24, 0)
    p.write_bytes(head + b"\0" * 4096)
    with pytest.raises(ValueError, match="outside the supported set"):
        gguf_io.GGUFReader(p)
    with pytest.raises(ValueError, match="does not match"):
        gguf_io.write_gguf(tmp_path / "x.gguf", [("w", np.zeros(10, np.uint8), 2, (1, 32))])


@pytest.mark.parametrize("t", [GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0],
                         ids=lambda t: t.name)
@pytest.mark.parametrize("hidden", [256, 1024])
def test_reference_dequantize_protocol_on_samples(tmp_path, oracle, t, hidden):
    """tests/test_dequantize.py:60-75 with the file read by our reader and the CPU op of this build:
    for every tensor of the sample, custom_ops.ggml_dequantize(torch.tensor(tensor.data), type, m, n)
    must equal the ground truth (there gguf.dequantize, here the oracle) — bit-exact for the CPU op."""
    import torch
    import custom_ops
    path = gguf_io.write_sample_file(tmp_path, t, hidden, seed=11)
    for tensor in gguf_io.GGUFReader(path).tensors:
        m, n = map(int, tensor.name.split("_")[-1].split("x"))
        out = custom_ops.ggml_dequantize(torch.tensor(tensor.data), tensor.tensor_type, m, n)
        ref = oracle.dequantize_f32(np.asarray(tensor.data), t, m * n).reshape(m, n)
        assert out.dtype == torch.float32 and tuple(out.shape) == (m, n)
        assert np.array_equal(out.numpy().view(np.uint32), ref.view(np.uint32))
