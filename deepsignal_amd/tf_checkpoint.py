"""Dependency-free TensorFlow checkpoint-V2 ("tensor bundle") reader / writer  (SURVEY.md section 8 row f3).

The reference restores its models with `tf.train.Saver().restore(sess, model_path)`
(/root/reference/deepsignal/call_modifications.py:210-211) from the files `train_model.py:33-36,242-243` writes:

    <prefix>.index                      a LevelDB-style sorted string table: tensor name -> BundleEntryProto
    <prefix>.data-00000-of-00001        the raw little-endian tensor bytes
    <prefix>.meta                       the graph (not needed here)

This module restates the published on-disk format (tensorflow/core/lib/io/table*, format.cc, block.cc and
tensorflow/core/util/tensor_bundle/tensor_bundle.cc, tensor_bundle.proto) without importing TensorFlow, so that
`--model_path some.ckpt` keeps working: `checkpoint_to_weights()` maps the variables of SURVEY.md Appendix B.7
one-to-one onto `deepsignal_amd.spec.tensor_table` and ignores the optimizer slots (`/Adam`, `/Adam_1`,
`beta1_power`, ...) that the reference's checkpoints also carry.

VALIDATION STATUS: TensorFlow is not installable in the build container and the reference ships no checkpoint, so the
reader is pinned by (a) its own writer (round trip, tests/test_tf_checkpoint.py), (b) hand-assembled byte-level
fixtures of the table format (prefix-compressed keys, restart arrays, multi-block index, snappy blocks, masked
crc32c known answers) and (c) nothing produced by TensorFlow itself. Every block and tensor checksum is verified
on read, so a layout misunderstanding fails loudly instead of yielding silently wrong weights.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from . import spec

TABLE_MAGIC = 0xDB4775248B80FB57          # table/format.h kTableMagicNumber
FOOTER_LEN = 48                           # two BlockHandles padded to 40 bytes + 8-byte magic
BLOCK_TRAILER_LEN = 5                     # 1-byte compression type + 4-byte masked crc32c
NO_COMPRESSION, SNAPPY_COMPRESSION = 0, 1
CRC_MASK_DELTA = 0xA282EAD8               # lib/hash/crc32c.h

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------
# crc32c (Castagnoli), slicing-by-8 tables
# ------------------------------------------------------------------------------------------------
def _make_tables() -> np.ndarray:
    poly = 0x82F63B78
    t = np.zeros((8, 256), dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[0, i] = c
    for i in range(256):
        c = int(t[0, i])
        for k in range(1, 8):
            c = int(t[0, c & 0xFF]) ^ (c >> 8)
            t[k, i] = c
    return t


_T = _make_tables()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C of `data` (same value as tensorflow::crc32c::Value)."""
    crc ^= 0xFFFFFFFF
    mv = memoryview(data)
    n = len(mv)
    t0, t1, t2, t3, t4, t5, t6, t7 = (_T[k].tolist() for k in range(8))
    i = 0
    end8 = n - (n % 8)
    while i < end8:
        lo = crc ^ (mv[i] | (mv[i + 1] << 8) | (mv[i + 2] << 16) | (mv[i + 3] << 24))
        crc = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24]
               ^ t3[mv[i + 4]] ^ t2[mv[i + 5]] ^ t1[mv[i + 6]] ^ t0[mv[i + 7]])
        i += 8
    while i < n:
        crc = t0[(crc ^ mv[i]) & 0xFF] ^ (crc >> 8)
        i += 1
    return crc ^ 0xFFFFFFFF


def _crc32c_large(buf: bytes) -> int:
    """crc32c for tensor payloads: the native library's table-driven implementation when it is built (the
    pure-Python loop above does ~2 MB/s, the 145 MB dense kernel would take a minute)."""
    try:
        from .engine import load_library
        import ctypes
        lib = load_library()
        fn = lib.ds_crc32c
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        fn.restype = ctypes.c_uint32
        a = np.frombuffer(buf, dtype=np.uint8)
        return int(fn(a.ctypes.data, a.size, 0))
    except (RuntimeError, OSError, AttributeError):
        return crc32c(buf)


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + CRC_MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - CRC_MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# varints / protobuf wire format (only what BundleHeaderProto / BundleEntryProto need)
# ------------------------------------------------------------------------------------------------
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v: int) -> bytes:
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


def _pb_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) of one serialized message."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            if len(v) != ln:
                raise ValueError("truncated length-delimited field")
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


class BundleEntry:
    """tensor_bundle.proto BundleEntryProto: dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32), slices=7."""

    def __init__(self, dtype: int = DT_FLOAT, shape: Tuple[int, ...] = (), shard_id: int = 0, offset: int = 0,
                 size: int = 0, crc: int = 0, sliced: bool = False):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc, self.sliced = \
            dtype, tuple(shape), shard_id, offset, size, crc, sliced

    @classmethod
    def parse(cls, buf: bytes) -> "BundleEntry":
        e = cls()
        e.dtype = 0
        for f, wt, v in _pb_fields(buf):
            if f == 1:
                e.dtype = int(v)
            elif f == 2:                                    # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}
                dims: List[int] = []
                for f2, _, v2 in _pb_fields(v):
                    if f2 == 2:
                        size = 0
                        for f3, _, v3 in _pb_fields(v2):
                            if f3 == 1:
                                size = _signed64(int(v3))
                        dims.append(size)
                    elif f2 == 3 and v2:
                        raise ValueError("tensor of unknown rank in checkpoint")
                e.shape = tuple(dims)
            elif f == 3:
                e.shard_id = int(v)
            elif f == 4:
                e.offset = _signed64(int(v))
            elif f == 5:
                e.size = _signed64(int(v))
            elif f == 6:
                e.crc = int(v)
            elif f == 7:
                e.sliced = True
        return e

    def serialize(self) -> bytes:
        out = bytearray()
        if self.dtype:
            out += b"\x08" + _put_varint(self.dtype)
        shp = bytearray()
        for d in self.shape:
            dim = b"\x08" + _put_varint(d)
            shp += b"\x12" + _put_varint(len(dim)) + dim
        out += b"\x12" + _put_varint(len(shp)) + bytes(shp)
        if self.shard_id:
            out += b"\x18" + _put_varint(self.shard_id)
        if self.offset:
            out += b"\x20" + _put_varint(self.offset)
        if self.size:
            out += b"\x28" + _put_varint(self.size)
        out += b"\x35" + struct.pack("<I", self.crc)
        return bytes(out)


# ------------------------------------------------------------------------------------------------
# snappy block decompression (the format is public: framing-less "raw" snappy). BundleWriter itself writes
# uncompressed index blocks; this is here so a table written with snappy still loads.
# ------------------------------------------------------------------------------------------------
def snappy_uncompress(buf: bytes) -> bytes:
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                       # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                                 # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# table reader
# ------------------------------------------------------------------------------------------------
def _read_block(data: bytes, offset: int, size: int, verify: bool = True) -> bytes:
    end = offset + size + BLOCK_TRAILER_LEN
    if offset < 0 or end > len(data):
        raise ValueError("block handle out of range")
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
        raise ValueError("block checksum mismatch at offset %d" % offset)
    if ctype == NO_COMPRESSION:
        return contents
    if ctype == SNAPPY_COMPRESSION:
        return snappy_uncompress(contents)
    raise ValueError("unknown block compression type %d" % ctype)


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    """Entries of one table block: [shared varint32][non_shared varint32][value_len varint32][key delta][value] ...,
    then uint32 restart offsets and a final uint32 restart count."""
    if len(block) < 4:
        raise ValueError("block too small")
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise ValueError("bad restart array")
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise ValueError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    """All (key, value) pairs of a TF/LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < FOOTER_LEN:
        raise ValueError("%s: too short for a table file" % path)
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise ValueError("%s: not a TensorFlow checkpoint index (bad table magic)" % path)
    pos = 0
    _meta_off, pos = _get_varint(footer, pos)
    _meta_size, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    out: List[Tuple[bytes, bytes]] = []
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


def read_index(prefix: str, verify: bool = True) -> Tuple[int, Dict[str, BundleEntry]]:
    """(num_shards, {tensor name: entry}) of checkpoint `prefix`."""
    path = prefix + ".index"
    if not os.path.exists(path):
        raise FileNotFoundError("%s: no such checkpoint index" % path)
    num_shards = 1
    entries: Dict[str, BundleEntry] = {}
    for key, value in read_table(path, verify):
        if key == b"":                                      # BundleHeaderProto: num_shards=1, endianness=2, version=3
            for f, _, v in _pb_fields(value):
                if f == 1:
                    num_shards = int(v)
                elif f == 2 and int(v) != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        entries[key.decode("utf-8")] = BundleEntry.parse(value)
    return num_shards, entries


def shard_path(prefix: str, shard: int, num_shards: int) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def load_checkpoint(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """Read tensors (all, or just `names`) from a checkpoint-V2 prefix."""
    num_shards, entries = read_index(prefix, verify)
    wanted = list(entries) if names is None else list(names)
    out: Dict[str, np.ndarray] = {}
    files: Dict[int, object] = {}
    try:
        for name in wanted:
            if name not in entries:
                raise KeyError("tensor %s not found in checkpoint %s" % (name, prefix))
            e = entries[name]
            if e.sliced:
                raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
            if e.dtype not in _DTYPES:
                raise ValueError("%s: unsupported dtype enum %d" % (name, e.dtype))
            dt = _DTYPES[e.dtype]
            count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
            if count * dt.itemsize != e.size:
                raise ValueError("%s: size %d does not match shape %s" % (name, e.size, e.shape))
            if e.shard_id not in files:
                files[e.shard_id] = open(shard_path(prefix, e.shard_id, num_shards), "rb")
            f = files[e.shard_id]
            f.seek(e.offset)
            buf = f.read(e.size)
            if len(buf) != e.size:
                raise ValueError("%s: data shard truncated" % name)
            if verify and unmask_crc(e.crc) != _crc32c_large(buf):
                raise ValueError("%s: tensor checksum mismatch" % name)
            out[name] = np.frombuffer(buf, dtype=dt).reshape(e.shape).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def is_checkpoint(path: str) -> bool:
    return os.path.exists(path + ".index")


def checkpoint_to_weights(prefix: str, kmer_len: int = 17, signal_len: int = 360, class_num: int = 2,
                          verify: bool = True, **variant) -> Dict[str, np.ndarray]:
    """The inference parameters of a reference-trained model as the engine's weight dict.

    Variable names are the reference graph's (SURVEY.md Appendix B.7 == spec.tensor_table); shapes are checked;
    optimizer slots and counters in the checkpoint are ignored. Conv kernels are stored HWIO [1,K,Cin,Cout] by TF,
    which is also the container's layout."""
    table = spec.tensor_table(kmer_len, signal_len, class_num, **variant)
    _, entries = read_index(prefix, verify)
    missing = [n for n, _ in table if n not in entries]
    if missing:
        raise KeyError("checkpoint %s lacks %d model variable(s), e.g. %s (is_cnn/is_rnn/is_base or kmer/signal "
                       "length differ from the trained model?)" % (prefix, len(missing), missing[0]))
    got = load_checkpoint(prefix, [n for n, _ in table], verify)
    out: Dict[str, np.ndarray] = {}
    for name, shape in table:
        a = got[name]
        if tuple(a.shape) != tuple(shape):
            raise ValueError("checkpoint tensor %s has shape %s, the model expects %s" % (name, a.shape, tuple(shape)))
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def convert(prefix: str, out_path: str, **kw) -> None:
    """TF checkpoint -> DSAMDW01 weight file (`python -m deepsignal_amd.tf_checkpoint in.ckpt out.dsw`)."""
    from . import weights
    weights.save_weights(out_path, checkpoint_to_weights(prefix, **kw))


# ------------------------------------------------------------------------------------------------
# writer (same format; used to export weights for the reference and by the round-trip tests)
# ------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval: int = 16):
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes) -> None:
        shared = 0
        if self.counter < self.interval:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def finish(self) -> bytes:
        out = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        return out

    def empty(self) -> bool:
        return not self.buf

    def size(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4


def _emit_block(f, contents: bytes) -> Tuple[int, int]:
    off = f.tell()
    f.write(contents)
    f.write(bytes([NO_COMPRESSION]))
    f.write(struct.pack("<I", mask_crc(crc32c(contents + bytes([NO_COMPRESSION])))))
    return off, len(contents)


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = 4096, restart_interval: int = 16) -> None:
    items = sorted(items, key=lambda kv: kv[0])
    with open(path, "wb") as f:
        index = _BlockBuilder(1)
        blk = _BlockBuilder(restart_interval)
        last = b""

        def flush():
            nonlocal blk
            if blk.empty():
                return
            off, size = _emit_block(f, blk.finish())
            index.add(last, _put_varint(off) + _put_varint(size))        # separator key >= every key of the block
            blk = _BlockBuilder(restart_interval)

        for k, v in items:
            blk.add(k, v)
            last = k
            if blk.size() >= block_size:
                flush()
        flush()
        meta_off, meta_size = _emit_block(f, _BlockBuilder(1).finish())   # empty metaindex block
        idx_off, idx_size = _emit_block(f, index.finish())
        footer = _put_varint(meta_off) + _put_varint(meta_size) + _put_varint(idx_off) + _put_varint(idx_size)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
        f.write(footer)


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    """Write `tensors` as a single-shard checkpoint-V2 (`prefix`.index + `prefix`.data-00000-of-00001)."""
    items: List[Tuple[bytes, bytes]] = []
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"          # num_shards=1, (endianness LITTLE = 0 omitted), version{producer=1}
    items.append((b"", header))
    offset = 0
    with open(shard_path(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])                    # (ascontiguousarray would turn scalars into shape (1,))
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _DTYPE_IDS:
                raise ValueError("%s: dtype %s not supported" % (name, a.dtype))
            buf = a.astype(dt, order="C", copy=False).tobytes()
            f.write(buf)
            e = BundleEntry(_DTYPE_IDS[np.dtype(dt)], a.shape, 0, offset, len(buf), mask_crc(_crc32c_large(buf)))
            items.append((name.encode("utf-8"), e.serialize()))
            offset += len(buf)
    write_table(prefix + ".index", items, block_size)


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 3:
        sys.exit("usage: python -m deepsignal_amd.tf_checkpoint <checkpoint prefix> <out.dsw>")
    convert(sys.argv[1], sys.argv[2])
    print("wrote %s" % sys.argv[2])
