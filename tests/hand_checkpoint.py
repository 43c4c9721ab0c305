"""A TensorFlow checkpoint-V2 pair (<prefix>.index + <prefix>.data-00000-of-00001) assembled BY HAND: LevelDB block and
footer layout, BundleHeaderProto / BundleEntryProto field bytes and masked CRC-32C are written out here from the
format descriptions, independent of deepsignal_amd.tf_checkpoint's own writer -- so the importer is tested against
bytes it did not produce (tests/test_tf_checkpoint.py on CPU, tests/test_gpu_configs.py through the CLI on the GPU).
No TensorFlow-written file exists in the build environment; this is the strongest pin available."""
import struct

import numpy as np


def _varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def _crc_tables():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return np.array(t, dtype=np.uint32)


def _crc32c(buf):
    """CRC-32C (Castagnoli, RFC 3720), the test's own table-driven statement; numpy-vectorised over 4 KiB strides would
    be overkill here -- the 145 MB dense kernel goes through the library's ds_crc32c, which test_tf_checkpoint.py pins
    to the RFC vectors; small buffers go through this one."""
    tab = _crc_tables()
    c = 0xFFFFFFFF
    for b in buf:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _block(entries):
    """LevelDB block, no prefix sharing, one restart point, no compression; returns (bytes incl. trailer, content size)."""
    body = b"".join(_varint(0) + _varint(len(k)) + _varint(len(v)) + k + v for k, v in entries)
    body += struct.pack("<II", 0, 1)
    trailer = b"\x00"
    return body + trailer + struct.pack("<I", _mask(_crc32c(body + trailer))), len(body)


def hand_checkpoint(prefix, tensors, native_crc=True):
    """<prefix>.data-00000-of-00001 = the float32 tensors in key order; <prefix>.index = one data block holding the
    header entry (key "", BundleHeaderProto{num_shards: 1, version{producer: 1}}) and one BundleEntryProto per tensor
    {dtype: DT_FLOAT(1), shape{dim{size}...}, offset, size, crc32c (masked CRC-32C of the tensor bytes)}, an empty
    meta-index block, an index block and the 48-byte footer -- the file layout TensorFlow's BundleWriter produces."""
    lib = None
    if native_crc:
        import ctypes
        from deepsignal_amd.engine import load_library
        lib = load_library()
        lib.ds_crc32c.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        lib.ds_crc32c.restype = ctypes.c_uint32
    entries = [(b"", b"\x08\x01" + b"\x1a\x02\x08\x01")]
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors):
            a = np.ascontiguousarray(tensors[name], dtype="<f4")
            raw = a.tobytes()
            crc = _crc32c(raw) if (lib is None or len(raw) <= 4096) else int(lib.ds_crc32c(a.ctypes.data, len(raw), 0))
            shape = b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(int(s)) for s in a.shape))
            proto = b"\x08\x01" + b"\x12" + _varint(len(shape)) + shape
            if off:
                proto += b"\x20" + _varint(off)
            proto += b"\x28" + _varint(len(raw)) + b"\x35" + struct.pack("<I", _mask(crc))
            entries.append((name.encode(), proto))
            f.write(raw)
            off += len(raw)
    blk, blk_size = _block(entries)
    meta, meta_size = _block([])
    index, index_size = _block([(b"\xff", _varint(0) + _varint(blk_size))])
    footer = _varint(len(blk)) + _varint(meta_size) + _varint(len(blk) + len(meta)) + _varint(index_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(prefix + ".index", "wb") as f:
        f.write(blk + meta + index + footer)
