"""Writes a TensorFlow V2 checkpoint ("tensor bundle": `<prefix>.index` + `<prefix>.data-SSSSS-of-NNNNN`) without TensorFlow,
for the tests of single-shot-detector_amd/ckpt_import.py.  Test infrastructure only, and deliberately NOT sharing code with
the reader: its own table builder (LevelDB table format as TF's lib/io/table writes it: prefix-compressed entries, a restart
point every `restart_interval` keys, block trailer = compression type 0 + masked CRC-32C, an empty metaindex block, an index
block of block handles keyed by the blocks' last keys, 48-byte footer), its own byte-at-a-time CRC-32C, and the two bundle
messages (BundleHeaderProto, BundleEntryProto; field numbers as published in tensorflow/core/protobuf/tensor_bundle.proto,
TF r1.12, third party) declared with google.protobuf's descriptor API and serialised by the official encoder.

No TensorFlow-written checkpoint is reachable offline, so this writer follows the published format, like the reader: what the
pair pins is the reader against an independent second reading of the format, not against TensorFlow's own bytes."""
import struct

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float16): 19}
DT_STRING = 7


def _messages():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "bundlepin/tensor_bundle.proto", "bundlepin", "proto3"

    def field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = msg.field.add()
        f.name, f.number, f.type, f.label = name, number, ftype, label
        if type_name:
            f.type_name = type_name

    shape = fd.message_type.add()
    shape.name = "TensorShapeProto"
    dim = shape.nested_type.add()
    dim.name = "Dim"
    field(dim, "size", 1, F.TYPE_INT64)
    field(dim, "name", 2, F.TYPE_STRING)
    field(shape, "dim", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bundlepin.TensorShapeProto.Dim")
    field(shape, "unknown_rank", 3, F.TYPE_BOOL)
    ver = fd.message_type.add()
    ver.name = "VersionDef"
    field(ver, "producer", 1, F.TYPE_INT32)
    field(ver, "min_consumer", 2, F.TYPE_INT32)
    sl = fd.message_type.add()
    sl.name = "TensorSliceProto"
    ext = sl.nested_type.add()
    ext.name = "Extent"
    field(ext, "start", 1, F.TYPE_INT64)
    field(ext, "length", 2, F.TYPE_INT64)
    field(sl, "extent", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bundlepin.TensorSliceProto.Extent")
    hdr = fd.message_type.add()
    hdr.name = "BundleHeaderProto"
    field(hdr, "num_shards", 1, F.TYPE_INT32)
    field(hdr, "endianness", 2, F.TYPE_INT32)              # enum on the wire = varint
    field(hdr, "version", 3, F.TYPE_MESSAGE, type_name=".bundlepin.VersionDef")
    ent = fd.message_type.add()
    ent.name = "BundleEntryProto"
    field(ent, "dtype", 1, F.TYPE_INT32)
    field(ent, "shape", 2, F.TYPE_MESSAGE, type_name=".bundlepin.TensorShapeProto")
    field(ent, "shard_id", 3, F.TYPE_INT32)
    field(ent, "offset", 4, F.TYPE_INT64)
    field(ent, "size", 5, F.TYPE_INT64)
    field(ent, "crc32c", 6, F.TYPE_FIXED32)
    field(ent, "slices", 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bundlepin.TensorSliceProto")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("bundlepin." + n)) for n in ("BundleHeaderProto", "BundleEntryProto")}


_MSG = None


def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CT = _crc_table()


def crc32c_bytewise(data):
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _CT[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def _v(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.reset()

    def reset(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _v(shared) + _v(len(key) - shared) + _v(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=4096, restart_interval=16):
    """items: [(key bytes, value bytes)] in ascending key order."""
    out = bytearray()
    index = _BlockBuilder(1)
    data = _BlockBuilder(restart_interval)

    def emit(content):
        off = len(out)
        out.extend(content)
        out.extend(b"\x00" + struct.pack("<I", mask(crc32c_bytewise(content + b"\x00"))))
        return _v(off) + _v(len(content))

    nblocks = 0
    for k, v in items:
        data.add(k, v)
        if data.size() >= block_size:
            last = data.last
            index.add(last, emit(data.finish()))
            data.reset()
            nblocks += 1
    if data.count or not nblocks:
        last = data.last
        index.add(last, emit(data.finish()))
    meta = emit(_BlockBuilder(1).finish())
    ix = emit(index.finish())
    foot = meta + ix
    foot += bytes(40 - len(foot)) + struct.pack("<Q", 0xdb4775248b80fb57)
    out.extend(foot)
    with open(path, "wb") as f:
        f.write(bytes(out))
    return nblocks + 1


def write_bundle(prefix, tensors, num_shards=1, block_size=4096, restart_interval=16, strings=(), fast_crc=None, sliced=()):
    """tensors: {name: ndarray}; variables go to the shards round-robin in key order.  strings: names written as DT_STRING entries
    (a Saver's bookkeeping tensors), sliced: names whose entry carries a `slices` list (a partitioned variable's full-shape
    entry) -- both are entries a reader of this model's weights must step over.  fast_crc: a CRC-32C callable for the tensors'
    bytes (default: this file's byte loop, ~5 MB/s).  Returns the number of table blocks of the index file."""
    global _MSG
    if _MSG is None:
        _MSG = _messages()
    crc = fast_crc or crc32c_bytewise
    names = sorted(list(tensors) + list(strings), key=lambda s: s.encode("utf-8"))
    shards = [bytearray() for _ in range(num_shards)]
    hdr = _MSG["BundleHeaderProto"]()
    hdr.num_shards = num_shards
    hdr.endianness = 0
    hdr.version.producer = 1
    items = [(b"", hdr.SerializeToString())]
    for i, name in enumerate(names):
        e = _MSG["BundleEntryProto"]()
        sid = i % num_shards
        if name in strings:
            raw = b"\x03abc"
            e.dtype = DT_STRING
        else:
            a = np.asarray(tensors[name], order="C")          # (ascontiguousarray would turn a scalar into shape (1,))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            e.dtype = DT[a.dtype]
            for d in a.shape:
                e.shape.dim.add().size = d
        e.shard_id = sid
        e.offset = len(shards[sid])
        e.size = len(raw)
        e.crc32c = mask(crc(raw))
        if name in sliced:
            s = e.slices.add()
            s.extent.add().length = 1
        shards[sid] += raw
        items.append((name.encode("utf-8"), e.SerializeToString()))
    for sid, sh in enumerate(shards):
        with open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "wb") as f:
            f.write(bytes(sh))
    return write_table(prefix + ".index", items, block_size, restart_interval)
