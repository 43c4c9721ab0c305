"""Scope row f3: TensorFlow checkpoint-V2 import (deepsignal_amd/tf_checkpoint.py), CPU only.

No TensorFlow and no reference checkpoint exist in the build container, so the reader is pinned
by published known answers of its primitives (CRC-32C test vectors of RFC 3720, the LevelDB varint / block layout,
the snappy format description) assembled BY HAND below -- not through the module's own writer -- plus a writer
round trip and failure-mode checks. "Parity unpinned" against TensorFlow's own output is stated in DESIGN.md.
"""
import os
import struct

import numpy as np
import pytest

from deepsignal_amd import spec, tf_checkpoint as T, weights as W


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors + the classic check value
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # incremental == one shot; mask / unmask are inverses and masking changes the value
    a, b = b"hello ", b"world"
    assert T.crc32c(b, T.crc32c(a)) == T.crc32c(a + b)
    c = T.crc32c(b"foo")
    assert T.unmask_crc(T.mask_crc(c)) == c and T.mask_crc(c) != c and T.mask_crc(T.mask_crc(c)) != c


def test_native_crc32c_matches_python(tmp_path):
    lib_path = os.path.join(os.path.dirname(T.__file__), "libdeepsignal_hip.so")
    if not os.path.exists(lib_path):
        pytest.skip("native library not built")
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 1000, 4097):
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T._crc32c_large(buf) == T.crc32c(buf)


def _varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def _hand_block(entries, ctype=0, compress=None):
    """entries: (shared, key_delta, value). One restart point at 0."""
    body = b"".join(_varint(s) + _varint(len(k)) + _varint(len(v)) + k + v for s, k, v in entries)
    body += struct.pack("<II", 0, 1)
    stored = compress(body) if compress else body
    trailer = bytes([ctype])
    return stored + trailer + struct.pack("<I", T.mask_crc(T.crc32c(stored + trailer))), len(stored)


def _hand_table(path, data_entries, **kw):
    blk, blk_size = _hand_block(data_entries, **kw)
    meta, meta_size = _hand_block([])
    last_key = b"zzzz"
    index, index_size = _hand_block([(0, last_key, _varint(0) + _varint(blk_size))])
    off_meta = len(blk)
    off_index = off_meta + len(meta)
    footer = _varint(off_meta) + _varint(meta_size) + _varint(off_index) + _varint(index_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(path, "wb") as f:
        f.write(blk + meta + index + footer)


def test_hand_assembled_table_with_prefix_compression(tmp_path):
    p = str(tmp_path / "t.index")
    # keys "dense/kernel", "dense_1/kernel" (shares "dense"), "model" ; LevelDB block entry layout written by hand
    _hand_table(p, [(0, b"dense/kernel", b"A"), (5, b"_1/kernel", b"BB"), (0, b"model", b"")])
    assert T.read_table(p) == [(b"dense/kernel", b"A"), (b"dense_1/kernel", b"BB"), (b"model", b"")]
    # flip one payload byte: the block checksum must catch it
    raw = bytearray(open(p, "rb").read())
    raw[4] ^= 1
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.read_table(p)
    assert len(T.read_table(p, verify=False)) == 3


def test_snappy_block(tmp_path):
    # hand-encoded raw snappy: varint length, literal "abcd" (tag (4-1)<<2), then a 1-byte-offset copy of
    # length 8 at offset 4 (tag kind 1: ((8-4)<<2)|1, offset byte 4) -> "abcd" + "abcdabcd"
    stream = _varint(12) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert T.snappy_uncompress(stream) == b"abcdabcdabcd"
    # 2-byte-offset copy and a long literal (length 61 -> tag 60<<2 with one extra length byte)
    lit = bytes(range(61))
    stream = _varint(61 + 5) + bytes([60 << 2, 60]) + lit + bytes([((5 - 1) << 2) | 2, 61, 0])
    assert T.snappy_uncompress(stream) == lit + lit[:5]

    def compress(body):      # a valid (literal-only) snappy stream of the block body
        out = _varint(len(body))
        for i in range(0, len(body), 60):
            chunk = body[i:i + 60]
            out += bytes([(len(chunk) - 1) << 2]) + chunk
        return out

    p = str(tmp_path / "s.index")
    _hand_table(p, [(0, b"k1", b"v1"), (1, b"2", b"v2")], ctype=1, compress=compress)
    assert T.read_table(p) == [(b"k1", b"v1"), (b"k2", b"v2")]


def test_bundle_entry_proto_bytes():
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:3} dim{size:4}}, offset: 300, size: 48, crc32c: 0x01020304}
    # written out field by field: tags 0x08, 0x12, 0x20, 0x28, 0x35 (fixed32)
    raw = (b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x03" + b"\x12\x02\x08\x04" + b"\x20\xac\x02" + b"\x28\x30"
           + b"\x35\x04\x03\x02\x01")
    e = T.BundleEntry.parse(raw)
    assert (e.dtype, e.shape, e.shard_id, e.offset, e.size, e.crc) == (T.DT_FLOAT, (3, 4), 0, 300, 48, 0x01020304)
    assert T.BundleEntry.parse(e.serialize()).__dict__ == e.__dict__
    scalar = T.BundleEntry.parse(b"\x08\x01\x12\x00\x28\x04\x35\x00\x00\x00\x00")       # beta1_power-like scalar
    assert scalar.shape == () and scalar.size == 4


def test_round_trip_multi_block(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"scope%02d/layer/kernel" % i: rng.normal(size=(3, i + 1)).astype(np.float32) for i in range(40)}
    tensors["global_step"] = np.array(7, dtype=np.int64)
    tensors["beta1_power"] = np.array(0.9, dtype=np.float32)
    prefix = str(tmp_path / "m.ckpt")
    T.write_checkpoint(prefix, tensors, block_size=256)          # forces many data blocks + restarts
    num_shards, entries = T.read_index(prefix)
    assert num_shards == 1 and set(entries) == set(tensors)
    got = T.load_checkpoint(prefix)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v)
    # corrupt one tensor byte in the data shard
    path = T.shard_path(prefix, 0, 1)
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0x40
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.load_checkpoint(prefix)


def test_checkpoint_to_weights_ignores_optimizer_slots(tmp_path):
    """A reference checkpoint holds every global variable: the model's 580 tensors plus two Adam slots per trainable
    variable and the beta powers (train_model.py:33-36). Only Appendix B.7 names are imported."""
    kw = dict(kmer_len=5, signal_len=40)
    w = W.random_weights(seed=5, lstm_bias_std=0.1, **kw)
    tensors = dict(w)
    for name in list(w)[:50]:
        if not name.endswith(("moving_mean", "moving_variance")):
            tensors[name + "/Adam"] = np.zeros_like(w[name])
            tensors[name + "/Adam_1"] = np.ones_like(w[name])
    tensors["beta1_power"] = np.array(0.5, np.float32)
    tensors["beta2_power"] = np.array(0.25, np.float32)
    prefix = str(tmp_path / "bn_5.sn_40.epoch_3.ckpt")
    T.write_checkpoint(prefix, tensors)
    got = T.checkpoint_to_weights(prefix, **kw)
    assert list(got) == [n for n, _ in spec.tensor_table(**kw)]
    for k in w:
        assert np.array_equal(got[k], w[k])
    out = str(tmp_path / "m.dsw")
    T.convert(prefix, out, **kw)
    back = W.load_weights(out)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    # wrong geometry / variant -> loud errors
    with pytest.raises(ValueError, match="shape"):
        T.checkpoint_to_weights(prefix, kmer_len=5, signal_len=80)
    partial = {k: v for k, v in w.items() if "embedding" not in k}
    T.write_checkpoint(str(tmp_path / "p.ckpt"), partial)
    with pytest.raises(KeyError, match="lacks"):
        T.checkpoint_to_weights(str(tmp_path / "p.ckpt"), **kw)


def test_model_path_resolution(tmp_path):
    from deepsignal_amd import call_modifications as cm
    kw = dict(kmer_len=5, signal_len=40)
    w = W.random_weights(seed=6, **kw)
    prefix = str(tmp_path / "x.ckpt")
    T.write_checkpoint(prefix, w)
    got = cm.load_model_weights(prefix, 5, 40, 2)
    assert got is not None and np.array_equal(got["dense/kernel"], w["dense/kernel"])
    dsw = str(tmp_path / "x.dsw")
    W.save_weights(dsw, w)
    assert cm.load_model_weights(dsw, 5, 40, 2) is None          # the engine reads weight files itself
    with pytest.raises(FileNotFoundError):
        cm.load_model_weights(str(tmp_path / "missing.ckpt"), 5, 40, 2)


def test_importer_reads_a_checkpoint_it_did_not_write(tmp_path):
    """tests/hand_checkpoint.py assembles the index table, the proto bytes and the checksums itself (pure Python
    CRC-32C included); the importer must hand back exactly the tensors that went in, ignore the optimizer slots and
    verify every checksum -- a flipped data byte is caught."""
    from hand_checkpoint import hand_checkpoint
    kw = dict(kmer_len=5, signal_len=40)
    w = W.random_weights(seed=9, lstm_bias_std=0.1, **kw)
    small = {k: v for k, v in w.items()}
    tensors = dict(small)
    tensors["dense/kernel/Adam"] = np.zeros_like(w["dense/kernel"])
    tensors["beta1_power"] = np.array(0.9, np.float32)
    prefix = str(tmp_path / "hand.ckpt")
    hand_checkpoint(prefix, tensors, native_crc=False)
    got = T.checkpoint_to_weights(prefix, **kw)
    assert list(got) == [n for n, _ in spec.tensor_table(**kw)]
    assert all(np.array_equal(got[k], w[k]) and got[k].shape == w[k].shape for k in w)
    raw = bytearray(open(T.shard_path(prefix, 0, 1), "rb").read())
    raw[100] ^= 0x10
    open(T.shard_path(prefix, 0, 1), "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.checkpoint_to_weights(prefix, **kw)
