"""TensorFlow's GraphDef message family, declared at test time with google.protobuf's descriptor API and serialised by
the OFFICIAL protobuf encoder -- an encoder this repository did not write -- so that single-shot-detector_amd/pb_import.py
(a hand-written wire-format reader) is pinned against something other than its own test writer (tests/helpers/pb_writer.py).

Field numbers and types as published in tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,types,
versions}.proto of TF r1.12 (the reference's version, README.md:22; third party, not vendored; the .proto texts are not
reproduced here, only the message shapes needed to encode a frozen inference graph).  Test infrastructure only.

`unpacked=True` builds the same family with `[packed = false]` on TensorProto's repeated scalars: the parser of a proto3
reader must accept both encodings of a repeated field, and old writers emit the unpacked one."""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 4, 7, 9, 10


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    if oneof is not None:
        f.oneof_index = oneof
    return f


def build(unpacked=False):
    """-> dict of message classes: GraphDef, NodeDef, AttrValue, TensorProto, TensorShapeProto, VersionDef."""
    pkg = "tfpin_u" if unpacked else "tfpin"
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = pkg + "/graph_family.proto", pkg, "proto3"
    P = "." + pkg + "."
    pk = False if unpacked else None

    shape = fd.message_type.add()
    shape.name = "TensorShapeProto"
    dim = shape.nested_type.add()
    dim.name = "Dim"
    _field(dim, "size", 1, F.TYPE_INT64)
    _field(dim, "name", 2, F.TYPE_STRING)
    _field(shape, "dim", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, P + "TensorShapeProto.Dim")
    _field(shape, "unknown_rank", 3, F.TYPE_BOOL)

    t = fd.message_type.add()
    t.name = "TensorProto"
    _field(t, "dtype", 1, F.TYPE_INT32)                    # enum DataType on the wire = varint
    _field(t, "tensor_shape", 2, F.TYPE_MESSAGE, type_name=P + "TensorShapeProto")
    _field(t, "version_number", 3, F.TYPE_INT32)
    _field(t, "tensor_content", 4, F.TYPE_BYTES)
    _field(t, "float_val", 5, F.TYPE_FLOAT, F.LABEL_REPEATED, packed=pk)
    _field(t, "double_val", 6, F.TYPE_DOUBLE, F.LABEL_REPEATED, packed=pk)
    _field(t, "int_val", 7, F.TYPE_INT32, F.LABEL_REPEATED, packed=pk)
    _field(t, "string_val", 8, F.TYPE_BYTES, F.LABEL_REPEATED)
    _field(t, "int64_val", 10, F.TYPE_INT64, F.LABEL_REPEATED, packed=pk)
    _field(t, "bool_val", 11, F.TYPE_BOOL, F.LABEL_REPEATED, packed=pk)
    _field(t, "half_val", 13, F.TYPE_INT32, F.LABEL_REPEATED, packed=pk)

    av = fd.message_type.add()
    av.name = "AttrValue"
    lv = av.nested_type.add()
    lv.name = "ListValue"
    _field(lv, "s", 2, F.TYPE_BYTES, F.LABEL_REPEATED)
    _field(lv, "i", 3, F.TYPE_INT64, F.LABEL_REPEATED, packed=pk)
    _field(lv, "f", 4, F.TYPE_FLOAT, F.LABEL_REPEATED, packed=pk)
    _field(lv, "b", 5, F.TYPE_BOOL, F.LABEL_REPEATED, packed=pk)
    _field(lv, "type", 6, F.TYPE_INT32, F.LABEL_REPEATED, packed=pk)
    _field(lv, "shape", 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, P + "TensorShapeProto")
    _field(lv, "tensor", 8, F.TYPE_MESSAGE, F.LABEL_REPEATED, P + "TensorProto")
    av.oneof_decl.add().name = "value"
    _field(av, "list", 1, F.TYPE_MESSAGE, type_name=P + "AttrValue.ListValue", oneof=0)
    _field(av, "s", 2, F.TYPE_BYTES, oneof=0)
    _field(av, "i", 3, F.TYPE_INT64, oneof=0)
    _field(av, "f", 4, F.TYPE_FLOAT, oneof=0)
    _field(av, "b", 5, F.TYPE_BOOL, oneof=0)
    _field(av, "type", 6, F.TYPE_INT32, oneof=0)
    _field(av, "shape", 7, F.TYPE_MESSAGE, type_name=P + "TensorShapeProto", oneof=0)
    _field(av, "tensor", 8, F.TYPE_MESSAGE, type_name=P + "TensorProto", oneof=0)
    _field(av, "placeholder", 9, F.TYPE_STRING, oneof=0)

    nd = fd.message_type.add()
    nd.name = "NodeDef"
    entry = nd.nested_type.add()                           # map<string, AttrValue> attr = 5
    entry.name = "AttrEntry"
    entry.options.map_entry = True
    _field(entry, "key", 1, F.TYPE_STRING)
    _field(entry, "value", 2, F.TYPE_MESSAGE, type_name=P + "AttrValue")
    _field(nd, "name", 1, F.TYPE_STRING)
    _field(nd, "op", 2, F.TYPE_STRING)
    _field(nd, "input", 3, F.TYPE_STRING, F.LABEL_REPEATED)
    _field(nd, "device", 4, F.TYPE_STRING)
    _field(nd, "attr", 5, F.TYPE_MESSAGE, F.LABEL_REPEATED, P + "NodeDef.AttrEntry")

    vd = fd.message_type.add()
    vd.name = "VersionDef"
    _field(vd, "producer", 1, F.TYPE_INT32)
    _field(vd, "min_consumer", 2, F.TYPE_INT32)
    _field(vd, "bad_consumers", 3, F.TYPE_INT32, F.LABEL_REPEATED, packed=pk)

    g = fd.message_type.add()
    g.name = "GraphDef"
    _field(g, "node", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, P + "NodeDef")
    _field(g, "version", 3, F.TYPE_INT32)
    _field(g, "versions", 4, F.TYPE_MESSAGE, type_name=P + "VersionDef")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(pkg + "." + n))
            for n in ("GraphDef", "NodeDef", "AttrValue", "TensorProto", "TensorShapeProto", "VersionDef")}


def _fill_tensor(t, arr, how):
    """how: 'content' (tensor_content bytes, what TF writes for large tensors), 'vals' (float_val / int_val lists),
    'splat' (ONE value for a constant-filled tensor: TF's compact form, e.g. batch-norm vectors of all ones / zeros)."""
    arr = np.asarray(arr)
    t.dtype = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32, np.dtype(np.int64): DT_INT64}[arr.dtype]
    for d in arr.shape:
        t.tensor_shape.dim.add().size = int(d)
    if how == "content":
        t.tensor_content = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    else:
        vals = arr.ravel()[:1] if how == "splat" else arr.ravel()
        field = {DT_FLOAT: t.float_val, DT_INT32: t.int_val, DT_INT64: t.int64_val}[t.dtype]
        field.extend(float(v) if t.dtype == DT_FLOAT else int(v) for v in vals)


def frozen_graph(weights, prefix="", unpacked=False, splat=(), as_vals=(), deterministic=False):
    """A frozen inference graph in the shape create_pb.py:57-85 leaves behind (tf.graph_util.convert_variables_to_constants):
    every variable a Const node `<prefix><name>` followed by its `<name>/read` Identity, between them the nodes a reader has
    to step over -- a uint8 Placeholder with unknown (-1) dimensions, consumers carrying list / shape / string / bool / float
    attributes, int32 / int64 / string Consts -- and the versions record.  Serialised by the official encoder."""
    M = build(unpacked)
    g = M["GraphDef"]()
    n = g.node.add()
    n.name, n.op = prefix + "images", "Placeholder"
    n.attr["dtype"].type = DT_UINT8
    for d in (-1, -1, -1, 3):
        n.attr["shape"].shape.dim.add().size = d                  # [None, None, None, 3]: negative varints, ten bytes each
    n = g.node.add()
    n.name, n.op = prefix + "resize/size", "Const"                 # an int32 Const: not a weight
    n.attr["dtype"].type = DT_INT32
    _fill_tensor(n.attr["value"].tensor, np.array([640, 896], np.int32), "vals")
    n = g.node.add()
    n.name, n.op = prefix + "Assert/data_0", "Const"               # a string Const: no numeric tensor at all
    n.attr["dtype"].type = DT_STRING
    n.attr["value"].tensor.dtype = DT_STRING
    n.attr["value"].tensor.tensor_shape.SetInParent()
    n.attr["value"].tensor.string_val.append(b"image must have 3 channels")
    n = g.node.add()
    n.name, n.op = prefix + "strided_slice/stack", "Const"         # int64, given as int64_val
    n.attr["dtype"].type = DT_INT64
    _fill_tensor(n.attr["value"].tensor, np.array([0, -1], np.int64), "vals")
    prev = prefix + "images"
    for k, (name, arr) in enumerate(weights.items()):
        c = g.node.add()
        c.name, c.op = prefix + name, "Const"
        c.attr["dtype"].type = DT_FLOAT
        how = "splat" if name in splat else ("vals" if name in as_vals else "content")
        _fill_tensor(c.attr["value"].tensor, np.asarray(arr, np.float32), how)
        r = g.node.add()
        r.name, r.op = prefix + name + "/read", "Identity"
        r.input.append(prefix + name)
        r.attr["T"].type = DT_FLOAT
        r.attr["_class"].list.s.append(("loc:@" + name).encode())
        if np.asarray(arr).ndim == 4:                                # the consumer of a kernel, with the attributes TF gives it
            u = g.node.add()
            u.name, u.op = prefix + name.rsplit("/", 1)[0] + "/Conv2D", "Conv2D"
            u.input.extend([prev, prefix + name + "/read"])
            u.attr["T"].type = DT_FLOAT
            u.attr["strides"].list.i.extend([1, 1, 2, 2])
            u.attr["dilations"].list.i.extend([1, 1, 1, 1])
            u.attr["padding"].s = b"SAME"
            u.attr["data_format"].s = b"NCHW"
            u.attr["use_cudnn_on_gpu"].b = True
            prev = u.name
        elif k % 4 == 0:
            u = g.node.add()
            u.name, u.op = prefix + name.rsplit("/", 1)[0] + "/FusedBatchNorm", "FusedBatchNorm"
            u.input.extend([prev, prefix + name + "/read"])
            u.attr["epsilon"].f = 1e-3
            u.attr["is_training"].b = False
    g.versions.producer = 27
    g.versions.min_consumer = 12
    return g.SerializeToString(deterministic=deterministic)
