"""Test-side writer of frozen GraphDefs (protobuf wire format, no TensorFlow): the only .pb files the
importer (single-shot-detector_amd/pb_import.py) has ever read were written by this file -- no real frozen graph
of the reference is reachable offline (create_pb.py:57-85 needs TF and a checkpoint).  Test infrastructure only."""
import numpy as np

DT_FLOAT = 1


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fn, v):
    return _enc_varint(fn << 3) + _enc_varint(v)


def write_frozen_graph(weights, path=None, extra_nodes=True, use_float_val=()):
    """Serialises {name: float32 ndarray} as a GraphDef of Const nodes (plus, like a real frozen
    graph, `name/read` Identity nodes and a Placeholder that carry no tensor)."""
    out = bytearray()
    if extra_nodes:
        ph = _ld(1, b"images") + _ld(2, b"Placeholder") + _ld(5, _ld(1, b"dtype") + _ld(2, _vi(6, 4)))
        out += _ld(1, ph)
    for name, arr in weights.items():
        a = np.ascontiguousarray(arr, dtype="<f4")
        shape = b"".join(_ld(2, _vi(1, d)) for d in a.shape)
        if name in use_float_val:
            payload = _ld(5, a.tobytes())                               # packed float_val
        else:
            payload = _ld(4, a.tobytes())                               # tensor_content
        tensor = _vi(1, DT_FLOAT) + _ld(2, shape) + payload
        node = (_ld(1, name.encode()) + _ld(2, b"Const") +
                _ld(5, _ld(1, b"dtype") + _ld(2, _vi(6, DT_FLOAT))) +
                _ld(5, _ld(1, b"value") + _ld(2, _ld(8, tensor))))
        out += _ld(1, node)
        if extra_nodes:
            ident = (_ld(1, (name + "/read").encode()) + _ld(2, b"Identity") + _ld(3, name.encode()) +
                     _ld(5, _ld(1, b"T") + _ld(2, _vi(6, DT_FLOAT))))
            out += _ld(1, ident)
    out += _ld(4, _vi(1, 26))                                           # GraphDef.versions.producer
    data = bytes(out)
    if path is not None:
        with open(path, "wb") as f:
            f.write(data)
    return data
