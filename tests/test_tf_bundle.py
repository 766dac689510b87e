"""TensorFlow-free reader / writer of Saver V2 checkpoints (gossipnet_amd/tf_bundle.py).
No TensorFlow and no reference checkpoint exist in this environment: the reader is pinned by (i) the CRC-32C
known answer, (ii) a bundle assembled BYTE BY BYTE below from the published table / proto layout, (iii) round trips
through the writer, (iv) corruption detection."""
import struct

import numpy as np
import pytest

from gossipnet_amd import tf_bundle as tb


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value
    assert tb.crc32c(b"") == 0
    assert tb.mask_crc(0) == 0xa282ead8


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def _trailer(block):
    return b"\x00" + struct.pack("<I", tb.mask_crc(tb.crc32c(block + b"\x00")))


def test_hand_assembled_bundle(tmp_path):
    """One float tensor `gnet/x` [2,3] and a scalar int64 `global_step`, every byte written out here."""
    x = np.arange(6, dtype=np.float32).reshape(2, 3)
    step = np.int64(1234)
    data = x.tobytes() + step.tobytes()
    # BundleEntryProto: dtype (1) varint, shape (2) bytes{dim (2) bytes{size (1) varint}}, offset (4), size (5), crc32c (6) fixed32
    def dim(n): return b"\x12" + _varint(2) + b"\x08" + _varint(n)
    shape_x = dim(2) + dim(3)
    e_x = (b"\x08\x01" + b"\x12" + _varint(len(shape_x)) + shape_x + b"\x20" + _varint(0) + b"\x28" + _varint(24) +
           b"\x35" + struct.pack("<I", tb.mask_crc(tb.crc32c(x.tobytes()))))
    e_s = (b"\x08\x09" + b"\x12\x00" + b"\x20" + _varint(24) + b"\x28" + _varint(8) +
           b"\x35" + struct.pack("<I", tb.mask_crc(tb.crc32c(step.tobytes()))))
    header = b"\x08\x01" + b"\x10\x00" + b"\x1a\x02\x08\x01"          # num_shards 1, little endian, version{producer 1}
    # data block: three entries, second key shares the prefix "g" with the first non-empty key
    k1, k2 = b"global_step", b"gnet/x"
    ent = lambda shared, key, val: _varint(shared) + _varint(len(key) - shared) + _varint(len(val)) + key[shared:] + val
    block = ent(0, b"", header) + ent(0, k1, e_s) + ent(1, k2, e_x)
    block += struct.pack("<I", 0) + struct.pack("<I", 1)               # one restart at offset 0
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                 # empty metaindex block
    table = block + _trailer(block)
    meta_off = len(table)
    table += meta + _trailer(meta)
    handle = _varint(0) + _varint(len(block))
    index = ent(0, b"h", handle) + struct.pack("<I", 0) + struct.pack("<I", 1)     # separator key >= "gnet/x"
    idx_off = len(table)
    table += index + _trailer(index)
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(idx_off) + _varint(len(index))
    table += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    prefix = str(tmp_path / "gnet-1234")
    open(prefix + ".index", "wb").write(table)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = tb.read_bundle(prefix)
    assert set(got) == {"gnet/x", "global_step"}
    assert np.array_equal(got["gnet/x"], x) and got["gnet/x"].dtype == np.float32
    assert got["global_step"].shape == () and int(got["global_step"]) == 1234
    # and the writer produces an index the reader parses to the same content
    tb.write_bundle(str(tmp_path / "w"), {"gnet/x": x, "global_step": np.asarray(step)})
    again = tb.read_bundle(str(tmp_path / "w"))
    assert np.array_equal(again["gnet/x"], x) and int(again["global_step"]) == 1234


def test_round_trip_many_variables_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"gnet/block%d/pw_fc1/weights" % b: rng.normal(size=(96, 64)).astype(np.float32) for b in range(1, 40)}
    tensors.update({"gnet/block%d/pw_fc1/biases" % b: rng.normal(size=64).astype(np.float32) for b in range(1, 40)})
    tensors["global_step"] = np.asarray(np.int64(77))
    tensors["flags"] = np.array([True, False, True])
    prefix = str(tmp_path / "gnet-77")
    tb.write_bundle(prefix, tensors, block_entries=16)        # several data blocks
    got = tb.read_bundle(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert np.array_equal(got[k], tensors[k]) and got[k].dtype == np.asarray(tensors[k]).dtype
    only = tb.read_bundle(prefix, names={"global_step"})
    assert list(only) == ["global_step"]
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[10] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[20] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        tb.read_bundle(prefix)
    with pytest.raises(ValueError, match="magic"):
        open(prefix + ".index", "wb").write(b"\x00" * 100)
        tb.read_bundle(prefix)
