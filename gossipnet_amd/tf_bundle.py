"""TensorFlow-free reader / writer of tf.train.Saver V2 checkpoints ("tensor bundles"): `<prefix>.index` +
`<prefix>.data-00000-of-00001` -- the files the reference writes as `gnet-<iteration>` (train.py:288,337,345) and
restores with `restorer.restore(sess, ...)` (train.py:274-304).  SURVEY.md 8f rank 2: load reference-trained weights
without a TensorFlow installation.

Format, restated from the published TensorFlow sources (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc},
tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/lib/io/{format,block,table}.{h,cc}, which follow
LevelDB's table format):

  .index   an SSTable: [data block]* [metaindex block] [index block] [footer]
           block   = entries + uint32 restart offsets[] + uint32 num_restarts, then a 5-byte trailer
                     (1 byte compression type: 0 = none, 1 = snappy; 4 bytes masked CRC-32C of block + type)
           entry   = varint32 shared_key_len, varint32 non_shared_key_len, varint32 value_len, key suffix, value
           index   block maps a key >= last key of a data block to that block's BlockHandle (varint64 offset, size)
           footer  = metaindex BlockHandle, index BlockHandle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE)
           key ""      -> BundleHeaderProto   {1: num_shards, 2: endianness (0 = little), 3: version}
           key <name>  -> BundleEntryProto    {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset,
                                               5: size, 6: crc32c (fixed32, masked), 7: slices (partitioned variables)}
  .data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset + size) of shard shard_id

NOT VALIDATED AGAINST A REAL TENSORFLOW CHECKPOINT: TensorFlow cannot be installed in this environment (no
network), and the reference ships no checkpoint.  The reader is tested against files produced by the writer below
and against a byte-level fixture assembled by hand in tests/test_tf_bundle.py; snappy-compressed blocks (TensorFlow
writes bundles uncompressed) and partitioned variables (`slices`) are rejected with a clear error.
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}

# ---------------------------------------------------------------- CRC-32C (Castagnoli), masked as in leveldb / TF
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF




def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf wire format
def _get_varint(b, i):
    r, s = 0, 0
    while True:
        x = b[i]; i += 1
        r |= (x & 0x7F) << s
        if not x & 0x80:
            return r, i
        s += 7


def _put_varint(v):
    out = bytearray()
    while True:
        x = v & 0x7F
        v >>= 7
        if v:
            out.append(x | 0x80)
        else:
            out.append(x)
            return bytes(out)


def _pb_fields(b):
    """Yields (field number, wire type, value) of a protobuf message (varint, fixed64, bytes, fixed32)."""
    i = 0
    while i < len(b):
        tag, i = _get_varint(b, i)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _get_varint(b, i)
        elif wt == 1:
            v = struct.unpack_from("<Q", b, i)[0]; i += 8
        elif wt == 2:
            n, i = _get_varint(b, i)
            v = bytes(b[i:i + n]); i += n
        elif wt == 5:
            v = struct.unpack_from("<I", b, i)[0]; i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _parse_shape(b):
    dims = []
    for f, _, v in _pb_fields(b):
        if f == 2:                                   # TensorShapeProto.dim
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
        elif f == 3 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(b):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, _, v in _pb_fields(b):
        if f == 1: e["dtype"] = v
        elif f == 2: e["shape"] = _parse_shape(v)
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
        elif f == 7: e["slices"] += 1
    return e


# ---------------------------------------------------------------- SSTable
def _read_block(buf, offset, size, verify):
    data, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(trailer) != 5:
        raise ValueError("truncated table block")
    if trailer[0] == 1:
        raise ValueError("snappy-compressed table block (TensorFlow writes bundle indexes uncompressed)")
    if trailer[0] != 0:
        raise ValueError("unknown block compression type %d" % trailer[0])
    if verify and struct.unpack("<I", trailer[1:])[0] != mask_crc(crc32c(data + trailer[:1])):
        raise ValueError("table block checksum mismatch")
    return data


def _block_entries(data):
    n_restarts = struct.unpack_from("<I", data, len(data) - 4)[0]
    limit = len(data) - 4 - 4 * n_restarts
    i, key = 0, b""
    while i < limit:
        shared, i = _get_varint(data, i)
        non_shared, i = _get_varint(data, i)
        vlen, i = _get_varint(data, i)
        key = key[:shared] + bytes(data[i:i + non_shared]); i += non_shared
        yield key, bytes(data[i:i + vlen]); i += vlen


def read_index(path, verify=True):
    """{tensor name: entry dict} and the header of `<prefix>.index`."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = buf[-48:]
    i = 0
    _, i = _get_varint(footer, i); _, i = _get_varint(footer, i)            # metaindex handle (unused)
    idx_off, i = _get_varint(footer, i); idx_size, i = _get_varint(footer, i)
    entries, header = {}, None
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        off, j = _get_varint(handle, 0); size, j = _get_varint(handle, j)
        for key, value in _block_entries(_read_block(buf, off, size, verify)):
            if key == b"":
                header = {f: v for f, _, v in _pb_fields(value)}
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise ValueError("checkpoint index without a bundle header")
    if header.get(2, 0) != 0:
        raise ValueError("big-endian checkpoint")
    return entries, {"num_shards": header.get(1, 1)}


def read_bundle(prefix, names=None, verify=True):
    """{variable name: numpy array} of the checkpoint `prefix` (e.g. './gnet-100000').  verify checks the table block
    checksums and the tensors' CRC-32C (pure Python: tensors above 4 MB are not re-hashed)."""
    entries, header = read_index(prefix + ".index", verify)
    shards, out = {}, {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e["slices"]:
            raise ValueError("partitioned variable %s (tensor slices) is not supported" % name)
        if e["dtype"] not in DTYPES:
            raise ValueError("unsupported dtype %d of %s" % (e["dtype"], name))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"]), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("data shard too short for %s" % name)
        if verify and e["crc32c"] is not None and e["size"] <= (4 << 20) and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("tensor checksum mismatch for %s" % name)
        dt = np.dtype(DTYPES[e["dtype"]])
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise ValueError("size of %s does not match its shape" % name)
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e["shape"]).copy()
    return out


# ---------------------------------------------------------------- writer (same format; used to hand weights back and by the tests)
def _pb_varint_field(f, v):
    return _put_varint((f << 3) | 0) + _put_varint(v & ((1 << 64) - 1))


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for n, (key, value) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, block_entries=64):
    """Writes {name: array} as a one-shard V2 checkpoint (`prefix`.index + .data-00000-of-00001), keys sorted as the
    table format requires."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    data = bytearray()
    items = [(b"", _pb_varint_field(1, 1) + _pb_varint_field(2, 0) + _pb_bytes_field(3, _pb_varint_field(1, 1)))]
    for name in names:
        a = np.require(np.asarray(tensors[name]), requirements="C")     # (ascontiguousarray would turn a scalar into [1])
        if a.dtype not in DTYPE_IDS:
            raise ValueError("unsupported dtype %s of %s" % (a.dtype, name))
        raw = a.tobytes()
        shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in a.shape)
        entry = (_pb_varint_field(1, DTYPE_IDS[a.dtype]) + _pb_bytes_field(2, shape) + _pb_varint_field(4, len(data)) +
                 _pb_varint_field(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack("<I", mask_crc(crc32c(raw))))
        items.append((name.encode("utf-8"), entry))
        data += raw
    table, index_items = bytearray(), []

    def emit(block):
        off = len(table)
        table.extend(block)
        table.extend(b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_items.append((chunk[-1][0], emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    table += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    return prefix


def is_bundle(prefix):
    return os.path.exists(prefix + ".index")
