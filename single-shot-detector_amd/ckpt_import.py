"""Reads the model's variables out of a TensorFlow checkpoint without TensorFlow.

What the reference leaves on disk before `create_pb.py` has been run -- and what a user of this build, who has no TensorFlow,
cannot turn into a `.pb` any more:

  * `params['model_dir']/model.ckpt-<step>.{index,data-00000-of-00001}` + the state file `checkpoint`, written by the
    Estimator of train.py:33-38,57-66 (tf.train.Saver, format V2);
  * `export/<timestamp>/variables/variables.{index,data-*}` next to `saved_model.pb`, written by create_pb.py:27-52
    (`estimator.export_savedmodel` restores the latest checkpoint of model_dir and saves the PREDICT graph's variables).

Both are "tensor bundles" (tensorflow/core/util/tensor_bundle; third party, TF r1.12, not vendored by the reference): the
`.index` file is an immutable sorted string table in LevelDB's table format (tensorflow/core/lib/io/table*, format.cc: data
blocks of prefix-compressed entries with a restart array, each block followed by a 1-byte compression type and a masked
CRC-32C; an index block of block handles; a 48-byte footer ending in the magic 0xdb4775248b80fb57) whose key "" holds a
BundleHeaderProto and whose other keys are variable names holding a BundleEntryProto {dtype, shape, shard_id, offset, size,
crc32c}; the `.data-SSSSS-of-NNNNN` shards hold the tensors' bytes, little-endian, at those offsets.  The field numbers are
those of tensor_bundle.proto / tensor_shape.proto / types.proto.

Variable names are the frozen graph's (variables.py): a checkpoint also carries `global_step`, the Adam slots and, for every
trainable variable, its moving average `<name>/ExponentialMovingAverage` (model.py:124-127).  create_pb.py restores the RAW
variables (export_savedmodel has no RestoreMovingAverageHook), the evaluation inside train.py the averages (model.py:148-161,
train.py:62-65): `use_ema` selects which, default = what create_pb.py would have frozen.

No TensorFlow-written checkpoint is reachable offline: tests/test_host.py reads bundles written by tests/helpers/
tf_bundle_writer.py (an independent writer: its own table builder and byte-wise CRC, protos through google.protobuf's
official encoder).
"""
import os
import re
import struct

import numpy as np

from .pb_import import _fields, _varint

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_BYTES = 48
BLOCK_TRAILER_BYTES = 5
EMA_SUFFIX = "/ExponentialMovingAverage"
# types.proto DataType -> numpy (the ones a Saver writes for this model and its optimizer; others are listed, not read)
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


# ----------------------------------------------------------------------------- CRC-32C (Castagnoli), masked as LevelDB does
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_T = _make_table()
_TL = [int(v) for v in _T]
_ZCACHE = {}


def _reg_bytes(s, data):
    """The CRC register after `data` (bytes-like), from register state s: the plain byte loop."""
    for b in bytes(data):
        s = _TL[(s ^ b) & 0xFF] ^ (s >> 8)
    return s


def _matvec(cols, v):
    r, k = 0, 0
    while v:
        if v & 1:
            r ^= cols[k]
        v >>= 1
        k += 1
    return r


def _zero_operator(n):
    """Four 256-entry tables of the GF(2)-linear map s -> register after n zero bytes (the update is affine in the register:
    reg(s, A || B) = Z_|B|(reg(s, A)) xor reg(0, B)), by square-and-multiply on 32 x 32 bit matrices."""
    if n in _ZCACHE:
        return _ZCACHE[n]
    one = [_TL[(1 << k) & 0xFF] ^ ((1 << k) >> 8) for k in range(32)]        # one zero byte, columns = images of the unit vectors
    result, power, e = None, one, n
    while e:
        if e & 1:
            result = power if result is None else [_matvec(power, c) for c in result]
        e >>= 1
        if e:
            power = [_matvec(power, c) for c in power]
    if result is None:
        result = [1 << k for k in range(32)]
    tabs = [[_matvec(result, b << (8 * j)) for b in range(256)] for j in range(4)]
    if len(_ZCACHE) > 64:
        _ZCACHE.clear()
    _ZCACHE[n] = tabs
    return tabs


def crc32c(data, crc=0):
    """CRC-32C of `data` (bytes-like or a uint8 array), continuing from `crc`.  Large inputs run as many lanes side by side in
    numpy (lane i = the register from 0 over chunk i), then the lanes are chained with the zero-bytes operator of the chunk
    length: a 9 MB kernel takes ~30 ms instead of ~3 s of byte loop."""
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    n = a.size
    s = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    if n < 1 << 14:
        return _reg_bytes(s, a.tobytes()) ^ 0xFFFFFFFF
    lanes = int(min(8192, max(64, n >> 9)))
    chunk = n // lanes
    body = np.ascontiguousarray(a[:lanes * chunk].reshape(lanes, chunk).T)      # row j = byte j of every lane
    reg = np.zeros(lanes, np.uint32)
    for j in range(chunk):
        idx = (reg ^ body[j]) & np.uint32(0xFF)
        reg = _T[idx] ^ (reg >> np.uint32(8))
    z0, z1, z2, z3 = _zero_operator(chunk)
    for r in reg.tolist():
        s = z0[s & 0xFF] ^ z1[(s >> 8) & 0xFF] ^ z2[(s >> 16) & 0xFF] ^ z3[s >> 24] ^ r
    s = _reg_bytes(s, a[lanes * chunk:].tobytes())
    return s ^ 0xFFFFFFFF


def masked_crc32c(data):
    """crc32c::Mask: what block trailers and BundleEntryProto.crc32c store (a CRC of data that itself embeds CRCs)."""
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- the sorted string table of the .index file
def _block(buf, offset, size, what, verify):
    if offset < 0 or size < 4 or offset + size + BLOCK_TRAILER_BYTES > len(buf):
        raise ValueError("checkpoint index: %s block [%d, +%d] lies outside the file (%d bytes)" % (what, offset, size, len(buf)))
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if masked_crc32c(bytes(buf[offset:offset + size + 1])) != stored:
            raise ValueError("checkpoint index: checksum mismatch in the %s block at offset %d" % (what, offset))
    if ctype == 1:
        raise ValueError("checkpoint index: snappy-compressed %s block (TensorFlow's BundleWriter writes uncompressed tables)" % what)
    if ctype != 0:
        raise ValueError("checkpoint index: unknown block compression type %d" % ctype)
    return buf[offset:offset + size]


def _entries(block):
    """(key, value) pairs of one table block: shared / unshared / value lengths as varint32, key delta, value; the restart
    array and its count at the block's end only bound the entry area."""
    n = len(block)
    restarts = struct.unpack_from("<I", block, n - 4)[0]
    end = n - 4 - 4 * restarts
    if restarts < 1 or end < 0:
        raise ValueError("checkpoint index: malformed block (restart count %d, %d bytes)" % (restarts, n))
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > end:
            raise ValueError("checkpoint index: malformed block entry at byte %d" % pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """{key bytes: value bytes} of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    if len(buf) < FOOTER_BYTES:
        raise ValueError("%s: %d bytes is too short for a checkpoint index" % (path, len(buf)))
    foot = buf[len(buf) - FOOTER_BYTES:]
    if struct.unpack_from("<Q", foot, 40)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (V2 tensor bundle): bad table magic -- a V1 checkpoint "
                         "(one model.ckpt file, tf.train.SaverDef.V1) is not supported" % path)
    pos = 0
    _mi_off, pos = _varint(foot, pos)
    _mi_size, pos = _varint(foot, pos)
    ix_off, pos = _varint(foot, pos)
    ix_size, pos = _varint(foot, pos)
    out = {}
    for _sep, handle in _entries(_block(buf, ix_off, ix_size, "index", verify)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for k, v in _entries(_block(buf, off, size, "data", verify)):
            out[k] = bytes(v)
    return out


# ----------------------------------------------------------------------------- the bundle
def _shape(buf):
    dims = []
    for fn, wt, v in _fields(buf):
        if fn == 2 and wt == 2:                         # TensorShapeProto.Dim
            size = 0
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 0:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def read_checkpoint_index(prefix, verify=True):
    """-> (header dict, {variable name: entry dict}) of the bundle `prefix` (`prefix + '.index'`).  An entry:
    dtype (types.proto number), shape, shard_id, offset, size, crc32c (masked), sliced (a partitioned variable)."""
    table = read_table(prefix + ".index", verify)
    if b"" not in table:
        raise ValueError("%s.index has no bundle header (key \"\")" % prefix)
    header = {"num_shards": 0, "endianness": 0, "producer": None}
    for fn, wt, v in _fields(memoryview(table[b""])):
        if fn == 1 and wt == 0:
            header["num_shards"] = v
        elif fn == 2 and wt == 0:
            header["endianness"] = v
        elif fn == 3 and wt == 2:
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 0:
                    header["producer"] = v2
    if header["endianness"] != 0:
        raise ValueError("%s: big-endian bundle" % prefix)
    entries = {}
    for key, val in table.items():
        if key == b"":
            continue
        e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
        for fn, wt, v in _fields(memoryview(val)):
            if fn == 1 and wt == 0:
                e["dtype"] = v
            elif fn == 2 and wt == 2:
                e["shape"] = _shape(v)
            elif fn == 3 and wt == 0:
                e["shard_id"] = v
            elif fn == 4 and wt == 0:
                e["offset"] = v
            elif fn == 5 and wt == 0:
                e["size"] = v
            elif fn == 6 and wt == 5:
                e["crc32c"] = v
            elif fn == 7 and wt == 2:
                e["sliced"] = True
        entries[key.decode("utf-8", "replace")] = e
    return header, entries


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray} of the bundle `prefix`: the variables `names` (KeyError for a missing one), or every variable of
    a numeric dtype.  verify: check the index blocks' and every read tensor's CRC-32C (what TensorFlow's BundleReader does)."""
    header, entries = read_checkpoint_index(prefix, verify)
    nshard = max(int(header["num_shards"]), 1)
    want = list(entries) if names is None else list(names)
    shards = {}
    out = {}
    for name in want:
        if name not in entries:
            raise KeyError("checkpoint %s has no variable %r" % (prefix, name))
        e = entries[name]
        dt = _DTYPES.get(e["dtype"])
        if dt is None or e["sliced"]:
            if names is None:
                continue
            raise ValueError("variable %r: %s" % (name, "a partitioned variable (slices) is not supported" if e["sliced"]
                                                  else "dtype %d is not numeric" % e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, nshard)
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        data = shards[sid]
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if e["size"] != count * np.dtype(dt).itemsize or e["offset"] < 0 or e["offset"] + e["size"] > data.size:
            raise ValueError("variable %r: entry (offset %d, size %d, shape %s) does not fit its dtype / shard of %d bytes"
                             % (name, e["offset"], e["size"], e["shape"], data.size))
        raw = np.array(data[e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] is not None and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError("variable %r: checksum mismatch in %s (shard %d, offset %d)" % (name, prefix, sid, e["offset"]))
        out[name] = raw.view(np.dtype(dt).newbyteorder("<")).astype(dt, copy=False).reshape(e["shape"])
    return out


def resolve_checkpoint(path):
    """The bundle prefix behind `path`, or None when `path` is no checkpoint: a prefix (`model.ckpt-1234`), one of its files
    (`.index`, `.data-00000-of-00001`), a model_dir with a `checkpoint` state file (its model_checkpoint_path, what
    tf.train.latest_checkpoint returns), or a SavedModel directory (`variables/variables`; the newest timestamp directory
    of create_pb.py's export/ folder when `path` holds such directories)."""
    path = str(path)
    if os.path.isdir(path):
        state = os.path.join(path, "checkpoint")
        if os.path.isfile(state):
            m = re.search(r'^\s*model_checkpoint_path\s*:\s*"((?:[^"\\]|\\.)*)"', open(state).read(), re.M)
            if m:
                p = m.group(1).encode().decode("unicode_escape")
                p = p if os.path.isabs(p) else os.path.join(path, p)
                if os.path.isfile(p + ".index"):
                    return p
                raise FileNotFoundError("%s names the checkpoint %r, whose .index file does not exist" % (state, p))
        if os.path.isfile(os.path.join(path, "variables", "variables.index")):
            return os.path.join(path, "variables", "variables")
        subs = sorted(d for d in os.listdir(path) if os.path.isfile(os.path.join(path, d, "variables", "variables.index")))
        if subs:
            return os.path.join(path, subs[-1], "variables", "variables")
        return None
    if os.path.isfile(path + ".index"):
        return path
    if path.endswith(".index") and os.path.isfile(path):
        return path[:-len(".index")]
    m = re.match(r"^(.*)\.data-\d{5}-of-\d{5}$", path)
    if m and os.path.isfile(m.group(1) + ".index"):
        return m.group(1)
    return None


def load_ckpt_weights(path, params, use_ema=False, verify=True):
    """The model's variables (variables.variable_shapes(params)) out of a checkpoint / model_dir / SavedModel directory.
    use_ema: take `<name>/ExponentialMovingAverage` where the checkpoint has it (every trainable variable of a train.py run;
    batch-norm moving statistics have none and are taken as they are: ema.variables_to_restore(), model.py:154-155)."""
    from .variables import variable_shapes
    prefix = resolve_checkpoint(path)
    if prefix is None:
        raise FileNotFoundError("%s is not a TensorFlow checkpoint prefix, model_dir or SavedModel directory" % path)
    _, entries = read_checkpoint_index(prefix, verify)
    shapes = variable_shapes(params)
    source = {}
    for name in shapes:
        src = name + EMA_SUFFIX if use_ema and name + EMA_SUFFIX in entries else name
        if src not in entries:
            raise KeyError("checkpoint %s has no variable %r" % (prefix, src))
        source[name] = src
    got = read_checkpoint(prefix, sorted(set(source.values())), verify)
    W = {}
    for name, shape in shapes.items():
        a = got[source[name]]
        if tuple(a.shape) != tuple(shape) or a.dtype != np.float32:
            raise ValueError("variable %r has shape %s dtype %s, expected %s float32" % (source[name], a.shape, a.dtype, shape))
        W[name] = np.ascontiguousarray(a)
    return W
