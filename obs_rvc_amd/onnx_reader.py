"""Dependency-free reader for the subset of ONNX (protobuf wire format) that a weight importer needs: the graph's
initializers (name, dims, dtype, data) and its node list (op_type, inputs, outputs).  SURVEY.md section 8 row f4: the
reference loads `*.onnx` files through ONNX Runtime (rvc/src/models.rs:48-76, names at models.rs:58-61,72); the native engine
reads its own blob format, so a user's files go through obs_rvc_amd.importers once.

Field numbers follow onnx.proto3 (ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5; NodeProto.input = 1,
.output = 2, .name = 3, .op_type = 4; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int32_data = 5,
.int64_data = 7, .name = 8, .raw_data = 9, .double_data = 10, .data_location = 14).  A small writer for the same subset is
included so that tests can produce files without the onnx package (absent from this image)."""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}
_DT_INV = {np.dtype(np.float32): 1, np.dtype(np.int32): 6, np.dtype(np.int64): 7, np.dtype(np.float16): 10, np.dtype(np.float64): 11}


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    v = s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7
        if s > 70:
            raise ValueError("malformed varint")


def _fields(b: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; length-delimited values are memoryview slices."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]; i += 8
        elif w == 2:
            ln, i = _varint(b, i)
            if i + ln > n:
                raise ValueError("truncated protobuf message")
            v = b[i:i + ln]; i += ln
        elif w == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        yield f, w, v


def _packed_varints(v) -> List[int]:
    out, i = [], 0
    v = bytes(v)
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _signed64(x: int) -> int:
    return x - (1 << 64) if x >= (1 << 63) else x


def _tensor(b: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw = 1, "", None
    floats: List[bytes] = []; i32: List[int] = []; i64: List[int] = []; f64: List[bytes] = []
    for f, w, v in _fields(b):
        if f == 1:
            dims += [_signed64(x) for x in (_packed_varints(v) if w == 2 else [v])]
        elif f == 2:
            dtype = v
        elif f == 4:
            floats.append(bytes(v))
        elif f == 5:
            i32 += _packed_varints(v) if w == 2 else [v]
        elif f == 7:
            i64 += _packed_varints(v) if w == 2 else [v]
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 10:
            f64.append(bytes(v))
        elif f == 14 and v == 1:
            raise ValueError("tensor %r uses external data, which is not supported" % name)
    if dtype not in _DT:
        raise ValueError("tensor %r: unsupported ONNX data_type %d" % (name, dtype))
    dt = np.dtype(_DT[dtype])
    if raw is not None:
        a = np.frombuffer(raw, dtype=dt.newbyteorder("<")).astype(dt)
    elif floats:
        a = np.frombuffer(b"".join(floats), dtype="<f4").astype(dt)
    elif f64:
        a = np.frombuffer(b"".join(f64), dtype="<f8").astype(dt)
    elif i64:
        a = np.array([_signed64(x) for x in i64], dtype=np.int64).astype(dt)
    elif i32:
        if dtype == 10:      # float16 payload travels as uint16 bit patterns in int32_data
            a = np.array(i32, dtype=np.uint16).view(np.float16)
        else:
            a = np.array([x - (1 << 32) if x >= (1 << 31) else x for x in (y & 0xFFFFFFFF for y in i32)], dtype=np.int64).astype(dt)
    else:
        a = np.zeros(0, dt)
    n = int(np.prod(dims)) if dims else a.size
    if a.size != n:
        raise ValueError("tensor %r: %d elements for dims %r" % (name, a.size, dims))
    return name, a.reshape(dims) if dims else a.reshape(())


def _node(b: bytes) -> Dict[str, object]:
    d = {"input": [], "output": [], "name": "", "op_type": ""}
    for f, w, v in _fields(b):
        if f == 1: d["input"].append(bytes(v).decode())
        elif f == 2: d["output"].append(bytes(v).decode())
        elif f == 3: d["name"] = bytes(v).decode()
        elif f == 4: d["op_type"] = bytes(v).decode()
    return d


def read_onnx(path: str) -> Tuple[Dict[str, np.ndarray], List[Dict[str, object]]]:
    """-> ({initializer name: array}, [node dicts in file order])."""
    data = memoryview(open(path, "rb").read())
    graph = None
    for f, w, v in _fields(data):
        if f == 7 and w == 2:
            graph = v
    if graph is None:
        raise ValueError("%s: no GraphProto (not an ONNX model?)" % path)
    inits: Dict[str, np.ndarray] = {}
    nodes: List[Dict[str, object]] = []
    for f, w, v in _fields(graph):
        if f == 5 and w == 2:
            name, a = _tensor(v)
            inits[name] = a
        elif f == 1 and w == 2:
            nodes.append(_node(v))
    return inits, nodes


# ----------------------------------------------------------------------------- writer (tests)
def _enc_varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = x & 0x7F; x >>= 7
        out.append(c | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def write_onnx(path: str, inits: Dict[str, np.ndarray], nodes: List[Dict[str, object]] = (), raw: bool = True) -> None:
    g = b""
    for nd in nodes:
        m = b"".join(_ld(1, s.encode()) for s in nd.get("input", [])) + b"".join(_ld(2, s.encode()) for s in nd.get("output", []))
        m += _ld(3, str(nd.get("name", "")).encode()) + _ld(4, str(nd["op_type"]).encode())
        g += _ld(1, m)
    g += _ld(2, b"graph")
    for name, a in inits.items():
        a = np.asarray(a, order="C")
        t = b"".join(_enc_varint((1 << 3) | 0) + _enc_varint(int(d)) for d in a.shape)
        t += _enc_varint((2 << 3) | 0) + _enc_varint(_DT_INV[a.dtype])
        if raw or a.dtype != np.float32:
            t += _ld(8, name.encode()) + _ld(9, a.astype(a.dtype.newbyteorder("<")).tobytes())
        else:
            t += _ld(4, a.astype("<f4").tobytes()) + _ld(8, name.encode())
        g += _ld(5, t)
    model = _enc_varint((1 << 3) | 0) + _enc_varint(8) + _ld(7, g)
    with open(path, "wb") as f:
        f.write(model)
