"""Minimal GGUF (v2/v3) tensor reader and writer — the on-disk format on the input side of the path.

The reference's tests and benchmark read their weights with gguf-py (`gguf.GGUFReader(path).tensors`,
tests/test_dequantize.py:15-21, benchmarks/utils.py:25-31, HK/tests/utils.py:25-31) from sample files named
`Quant_{TYPE}_{hidden}.gguf`; gguf-py is not installed here and the samples live on the HF hub.  This module
is an independent implementation of the public GGUF container layout (header, metadata key/values, tensor
infos, aligned data section) that exposes the three ReaderTensor attributes those call sites use:

    t.name         str
    t.tensor_type  GGMLType (IntEnum, same ids as gguf.GGMLQuantizationType)
    t.data         numpy array; block-quantised tensors as uint8 [rows, row_bytes] (memory-mapped, zero copy),
                   F32/F16 as typed arrays of the logical shape
    t.shape        logical shape, innermost dimension first as stored in the file (ne[0] = K)

Only what the hot path needs is supported: F32, F16 and the ten block formats of ggq.formats.
"""
import mmap
import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .formats import BLOCK, GGMLType

GGUF_MAGIC = 0x46554747  # "GGUF" little-endian
DEFAULT_ALIGNMENT = 32

# metadata value types of the container
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)
_SCALAR = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f", _BOOL: "<?",
           _U64: "<Q", _I64: "<q", _F64: "<d"}
_F32_TYPE, _F16_TYPE = 0, 1


def _type_size(t: int) -> Tuple[int, int]:
    """(elements per block, bytes per block) of a ggml tensor type id"""
    if t == _F32_TYPE:
        return 1, 4
    if t == _F16_TYPE:
        return 1, 2
    try:
        return BLOCK[GGMLType(t)]
    except (ValueError, KeyError):
        raise ValueError(f"GGUF tensor type {t} is outside the supported set (F32, F16, Q4_0..Q6_K)") from None


@dataclass
class ReaderTensor:
    name: str
    tensor_type: object       # GGMLType for block formats, int 0/1 for F32/F16
    shape: Tuple[int, ...]    # ne[0] (innermost) first
    n_bytes: int
    data_offset: int          # absolute file offset
    data: np.ndarray


class _Cursor:
    def __init__(self, buf, pos=0):
        self.buf, self.pos = buf, pos

    def unpack(self, fmt):
        v = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += struct.calcsize(fmt)
        return v[0] if len(v) == 1 else v

    def string(self, len_fmt="<Q"):
        n = self.unpack(len_fmt)
        if n > len(self.buf) - self.pos:
            raise ValueError("GGUF: string length runs past the end of the file")
        s = bytes(self.buf[self.pos:self.pos + n]).decode("utf-8")
        self.pos += n
        return s

    def value(self, vtype, len_fmt="<Q"):
        if vtype in _SCALAR:
            return self.unpack(_SCALAR[vtype])
        if vtype == _STR:
            return self.string(len_fmt)
        if vtype == _ARR:
            etype = self.unpack("<I")
            count = self.unpack(len_fmt)
            return [self.value(etype, len_fmt) for _ in range(count)]
        raise ValueError(f"GGUF: unknown metadata value type {vtype}")


class GGUFReader:
    """`GGUFReader(path).tensors` — list of ReaderTensor, in file order; `.fields` — metadata dict."""

    def __init__(self, path):
        self.path = os.fspath(path)
        self._file = open(self.path, "rb")
        size = os.fstat(self._file.fileno()).st_size
        if size < 24:
            raise ValueError(f"{self.path}: too short to be a GGUF file")
        self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        cur = _Cursor(self._map)
        magic, self.version = cur.unpack("<I"), cur.unpack("<I")
        if magic != GGUF_MAGIC:
            raise ValueError(f"{self.path}: bad magic 0x{magic:08x} (not a GGUF file)")
        if self.version not in (2, 3):
            raise ValueError(f"{self.path}: GGUF version {self.version} not supported (2 or 3)")
        n_tensors, n_kv = cur.unpack("<Q"), cur.unpack("<Q")
        self.fields: Dict[str, object] = {}
        for _ in range(n_kv):
            key = cur.string()
            self.fields[key] = cur.value(cur.unpack("<I"))
        self.alignment = int(self.fields.get("general.alignment", DEFAULT_ALIGNMENT))
        if self.alignment <= 0 or self.alignment & (self.alignment - 1):
            raise ValueError(f"{self.path}: general.alignment {self.alignment} is not a power of two")
        infos = []
        for _ in range(n_tensors):
            name = cur.string()
            n_dims = cur.unpack("<I")
            if not 1 <= n_dims <= 4:
                raise ValueError(f"{self.path}: tensor {name!r} has {n_dims} dimensions")
            dims = tuple(cur.unpack("<Q") for _ in range(n_dims))
            ttype, offset = cur.unpack("<I"), cur.unpack("<Q")
            infos.append((name, dims, ttype, offset))
        data_start = -(-cur.pos // self.alignment) * self.alignment
        self.tensors: List[ReaderTensor] = []
        for name, dims, ttype, offset in infos:
            qk, bs = _type_size(ttype)
            if dims[0] % qk:
                raise ValueError(f"{self.path}: tensor {name!r}: ne[0]={dims[0]} is not a multiple of {qk}")
            n_elems = int(np.prod(dims, dtype=np.int64))
            n_bytes = n_elems // qk * bs
            start = data_start + offset
            if offset % self.alignment or start + n_bytes > size:
                raise ValueError(f"{self.path}: tensor {name!r}: data offset {offset} / size {n_bytes} out of bounds")
            raw = np.frombuffer(self._map, dtype=np.uint8, count=n_bytes, offset=start)
            if ttype == _F32_TYPE:
                data, tt = raw.view(np.float32).reshape(dims[::-1]), ttype
            elif ttype == _F16_TYPE:
                data, tt = raw.view(np.float16).reshape(dims[::-1]), ttype
            else:  # rows = all outer dimensions, one row = ne[0] elements
                data, tt = raw.reshape(n_elems // dims[0], dims[0] // qk * bs), GGMLType(ttype)
            self.tensors.append(ReaderTensor(name, tt, dims, n_bytes, start, data))

    def get_tensor(self, name: str) -> ReaderTensor:
        for t in self.tensors:
            if t.name == name:
                return t
        raise KeyError(name)

    def close(self):
        self.tensors = []
        try:
            self._map.close()
        except BufferError:   # numpy views still alive: the map is released with them
            pass
        self._file.close()


def _pack_string(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _pack_value(v) -> bytes:
    if isinstance(v, bool):
        return struct.pack("<I?", _BOOL, v)
    if isinstance(v, int):
        return struct.pack("<Iq", _I64, v) if v < 0 or v >= 2 ** 32 else struct.pack("<II", _U32, v)
    if isinstance(v, float):
        return struct.pack("<If", _F32, v)
    if isinstance(v, str):
        return struct.pack("<I", _STR) + _pack_string(v)
    if isinstance(v, (list, tuple)) and all(isinstance(e, str) for e in v):
        return struct.pack("<IIQ", _ARR, _STR, len(v)) + b"".join(_pack_string(e) for e in v)
    if isinstance(v, (list, tuple)) and all(isinstance(e, int) and not isinstance(e, bool) for e in v):
        return struct.pack("<IIQ", _ARR, _I32, len(v)) + b"".join(struct.pack("<i", e) for e in v)
    raise TypeError(f"unsupported GGUF metadata value {v!r}")


def write_gguf(path, tensors: Sequence[Tuple[str, np.ndarray, int, Tuple[int, int]]], metadata: Dict[str, object] = None,
               alignment: int = DEFAULT_ALIGNMENT) -> None:
    """tensors: (name, payload, ggml type id, (rows, cols)); payload = uint8 [rows, row_bytes] for block
    formats, float32/float16 [rows, cols] for F32/F16.  Writes a GGUF v3 file."""
    metadata = dict(metadata or {})
    metadata.setdefault("general.architecture", "ggq-sample")
    if alignment != DEFAULT_ALIGNMENT:
        metadata["general.alignment"] = alignment
    head = struct.pack("<IIQQ", GGUF_MAGIC, 3, len(tensors), len(metadata))
    for k, v in metadata.items():
        head += _pack_string(k) + _pack_value(v)
    infos, blobs, offset = b"", [], 0
    for name, payload, ttype, (rows, cols) in tensors:
        qk, bs = _type_size(int(ttype))
        blob = np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
        if cols % qk or blob.size != rows * (cols // qk) * bs:
            raise ValueError(f"tensor {name!r}: payload of {blob.size} bytes does not match {rows} x {cols} of type {int(ttype)}")
        infos += _pack_string(name) + struct.pack("<IQQIQ", 2, cols, rows, int(ttype), offset)
        blobs.append((offset, blob))
        offset = -(-(offset + blob.size) // alignment) * alignment
    with open(path, "wb") as f:
        f.write(head + infos)
        pad = -f.tell() % alignment
        f.write(b"\0" * pad)
        base = f.tell()
        for off, blob in blobs:
            f.seek(base + off)
            f.write(blob.tobytes())
        end = base + offset
        if f.tell() < end:   # keep the file a whole number of alignment units, as llama.cpp's writer does
            f.seek(end - 1)
            f.write(b"\0")


def sample_filename(quant_type: GGMLType, hidden_size: int) -> str:
    """naming convention of the reference's sample repo (benchmarks/utils.py:29)"""
    return f"Quant_{GGMLType(quant_type).name}_{hidden_size}.gguf"


def write_sample_file(directory, quant_type: GGMLType, hidden_size: int, seed: int = 0, d_scale: float = 1.0,
                      row_multiples: Sequence[int] = (1, 2, 3)) -> str:
    """A synthetic stand-in for `Quant_{TYPE}_{hidden}.gguf`: tensors named `tensor_{rows}x{cols}` (the reference
    parses the shape from the name, benchmark_mmq.py:68-73) with cols = hidden_size and random valid blocks."""
    from . import synth
    os.makedirs(directory, exist_ok=True)
    tensors = []
    for i, mult in enumerate(row_multiples):
        rows = hidden_size * mult
        w = synth.random_weight(quant_type, rows, hidden_size, seed=seed + i, d_scale=d_scale)
        tensors.append((f"tensor_{rows}x{hidden_size}", w, int(quant_type), (rows, hidden_size)))
    path = os.path.join(os.fspath(directory), sample_filename(quant_type, hidden_size))
    write_gguf(path, tensors, {"general.name": f"synthetic {GGMLType(quant_type).name} sample"})
    return path
