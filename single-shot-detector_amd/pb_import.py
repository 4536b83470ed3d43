"""Reads the weights out of the reference's frozen inference graph (`inference/model.pb`,
written by create_pb.py:57-85) without TensorFlow: a minimal protobuf wire-format reader for
GraphDef -> NodeDef(op == 'Const') -> attr['value'].tensor (TensorProto).

After tf.graph_util.convert_variables_to_constants every variable `scope/name` of the model
is a Const node of that name holding the float32 tensor; inference/detector.py:13-19 imports
the graph under the prefix 'import/', the file itself has no prefix (a prefix is stripped if
present).  Field numbers are those of tensorflow/core/framework/{graph,node_def,attr_value,
tensor,tensor_shape}.proto (TF r1.12, third-party, not vendored by the reference).

No pretrained .pb is reachable offline; tests/test_host.py round-trips a synthetic GraphDef
written by tests/helpers/pb_writer.py (same wire format, plus non-Const nodes to skip).
"""
import struct

import numpy as np

DT_FLOAT, DT_INT32 = 1, 3


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """Yields (field_number, wire_type, value) over one message; value is an int for varint /
    fixed fields and a memoryview for length-delimited ones."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        if pos > n:
            raise ValueError("truncated protobuf message")
        yield fn, wt, v


def _tensor(buf):
    """TensorProto -> ndarray (float32 / int32 only; others return None)."""
    dtype, shape, content, fvals, ivals = None, [], None, [], []
    for fn, wt, v in _fields(buf):
        if fn == 1 and wt == 0:
            dtype = v
        elif fn == 2 and wt == 2:                      # TensorShapeProto
            for f2, w2, v2 in _fields(v):
                if f2 == 2 and w2 == 2:                # Dim
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1 and w3 == 0:
                            shape.append(v3 if v3 < (1 << 63) else v3 - (1 << 64))
        elif fn == 4 and wt == 2:
            content = bytes(v)
        elif fn == 5:                                   # float_val (packed or not)
            if wt == 2:
                fvals.extend(struct.unpack("<%df" % (len(v) // 4), bytes(v)))
            elif wt == 5:
                fvals.append(struct.unpack("<f", struct.pack("<I", v))[0])
        elif fn == 7:                                   # int_val
            if wt == 2:
                p, b = 0, bytes(v)
                while p < len(b):
                    x, p = _varint(b, p)
                    ivals.append(x)
            elif wt == 0:
                ivals.append(v)
    np_dtype = {DT_FLOAT: np.float32, DT_INT32: np.int32}.get(dtype)
    if np_dtype is None:
        return None
    count = int(np.prod(shape)) if shape else 1
    if content is not None and len(content) > 0:
        arr = np.frombuffer(content, dtype=np.dtype(np_dtype).newbyteorder("<")).astype(np_dtype)
    else:
        vals = fvals if dtype == DT_FLOAT else ivals
        if len(vals) == 0:
            arr = np.zeros(count, np_dtype)
        elif len(vals) == 1:                            # TF: a single value fills the tensor
            arr = np.full(count, vals[0], np_dtype)
        else:
            arr = np.asarray(vals, np_dtype)
    if arr.size != count:
        raise ValueError("tensor content has %d values, shape %s needs %d" % (arr.size, shape, count))
    return arr.reshape(shape)


def read_frozen_graph(path_or_bytes, prefix="import/"):
    """Returns {node name: ndarray} for every float/int Const node of a GraphDef."""
    data = path_or_bytes
    if not isinstance(data, (bytes, bytearray, memoryview)):
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    buf = memoryview(bytes(data))
    consts = {}
    for fn, wt, node in _fields(buf):
        if fn != 1 or wt != 2:                          # GraphDef.node
            continue
        name, op, value = None, None, None
        for f2, w2, v2 in _fields(node):
            if f2 == 1 and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif f2 == 2 and w2 == 2:
                op = bytes(v2).decode("utf-8")
            elif f2 == 5 and w2 == 2:                   # attr map entry {1: key, 2: AttrValue}
                key, av = None, None
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1 and w3 == 2:
                        key = bytes(v3)
                    elif f3 == 2 and w3 == 2:
                        av = v3
                if key == b"value" and av is not None:
                    for f4, w4, v4 in _fields(av):
                        if f4 == 8 and w4 == 2:         # AttrValue.tensor
                            value = v4
        if op == "Const" and name is not None and value is not None:
            arr = _tensor(value)
            if arr is not None:
                if prefix and name.startswith(prefix):
                    name = name[len(prefix):]
                consts[name] = arr
    return consts


def load_pb_weights(path_or_bytes, params):
    """The model's variables (variables.variable_shapes(params)) out of a frozen graph."""
    from .variables import variable_shapes
    consts = read_frozen_graph(path_or_bytes)
    W = {}
    for name, shape in variable_shapes(params).items():
        if name not in consts:
            raise KeyError("frozen graph has no Const node %r" % name)
        a = consts[name]
        if tuple(a.shape) != tuple(shape) or a.dtype != np.float32:
            raise ValueError("Const %r has shape %s dtype %s, expected %s float32" % (name, a.shape, a.dtype, shape))
        W[name] = np.ascontiguousarray(a)
    return W
