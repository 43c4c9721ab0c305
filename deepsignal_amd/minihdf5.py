"""Read-only HDF5 access for tombo-resquiggled single-read fast5 files, in plain Python + numpy.

The reference opens its inputs with h5py (deepsignal/extract_features.py:35-72,75-140,193-208: `Raw/Reads/<read>/Signal`
and its `read_id` attribute, the `UniqueGlobalKey/channel_id` attributes, the compound `Events` dataset of the
corrected group and the `Alignment` attributes -- SURVEY.md Appendix C.4). h5py is not part of the MI355X image, so the
host feature extraction could not open a real fast5 file there. This module reads exactly the subset of the HDF5 file
format such files use, following the published "HDF5 File Format Specification Version 2.0/3.0":

  * superblock versions 0, 1 (HDF5 1.8 / h5py default `libver="earliest"`) and 2, 3;
  * object headers version 1 and 2, continuation blocks;
  * groups as symbol tables (B-tree v1 + local heap + SNOD nodes) and as compact link messages;
  * datasets: contiguous, compact and chunked (B-tree v1 chunk index) layouts, filters deflate / shuffle / fletcher32;
  * datatypes: fixed-point, IEEE float, fixed strings, variable-length strings (global heap), compound, enum, array;
  * attributes (message versions 1 - 3) stored in the object header.

Anything else (dense groups / attributes in fractal heaps, layout version 4 indexes, the VBZ signal filter of recent ONT
files, ...) raises `Unsupported` with the feature's name: nothing is silently skipped. The API mirrors the h5py calls the
extractor makes: `File(path)[name]`, `group.values()`, `name in file`, `dataset[()]`, `dataset[field]`, `.attrs[name]`.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Unsupported(NotImplementedError):
    """The file uses an HDF5 feature this reader does not implement (named in the message)."""


class _Buf:
    def __init__(self, data: bytes, so: int = 8, sl: int = 8):
        self.d, self.so, self.sl = data, so, sl

    def u(self, pos: int, size: int) -> int:
        return int.from_bytes(self.d[pos:pos + size], "little")

    def off(self, pos: int) -> int:
        return self.u(pos, self.so)

    def length(self, pos: int) -> int:
        return self.u(pos, self.sl)

    def undefined(self, v: int) -> bool:
        return v == (1 << (8 * self.so)) - 1


# ------------------------------------------------------------------------------------------------ datatypes
class _Type:
    """numpy dtype of an HDF5 datatype message, plus what the fixed layout cannot express (variable-length strings)."""

    def __init__(self, dtype, vlen_str: bool = False, size: int = 0):
        self.dtype, self.vlen_str, self.size = dtype, vlen_str, size


def _cstr(d: bytes, pos: int) -> Tuple[str, int]:
    end = d.index(b"\x00", pos)
    return d[pos:end].decode("utf-8"), end + 1


def _parse_type(d: bytes, pos: int) -> Tuple[_Type, int]:
    """Datatype message at d[pos:] -> (_Type, position behind the message)."""
    cv = d[pos]
    cls, ver = cv & 0x0F, cv >> 4
    bits = d[pos + 1] | (d[pos + 2] << 8) | (d[pos + 3] << 16)
    size = int.from_bytes(d[pos + 4:pos + 8], "little")
    p = pos + 8
    if cls == 0:                                   # fixed point
        order = ">" if bits & 1 else "<"
        kind = "i" if bits & 0x08 else "u"
        if size not in (1, 2, 4, 8):
            raise Unsupported("fixed-point type of %d bytes" % size)
        return _Type(np.dtype(order + kind + str(size)), size=size), p + 4
    if cls == 1:                                   # floating point
        order = ">" if bits & 1 else "<"
        if size not in (2, 4, 8):
            raise Unsupported("floating-point type of %d bytes" % size)
        return _Type(np.dtype(order + "f" + str(size)), size=size), p + 12
    if cls == 2:                                   # time
        return _Type(np.dtype("V%d" % size), size=size), p + 2
    if cls == 3:                                   # fixed-length string
        return _Type(np.dtype("S%d" % size), size=size), p
    if cls == 4:                                   # bit field
        return _Type(np.dtype("<u%d" % size) if size in (1, 2, 4, 8) else np.dtype("V%d" % size), size=size), p + 4
    if cls == 5:                                   # opaque: tag padded to a multiple of 8
        return _Type(np.dtype("V%d" % size), size=size), p + ((bits & 0xFF) + 7) // 8 * 8
    if cls == 6:                                   # compound
        nmemb = bits & 0xFFFF
        names, formats, offsets = [], [], []
        for _ in range(nmemb):
            name, q = _cstr(d, p)
            if ver < 3:
                q = p + (q - p + 7) // 8 * 8       # name padded to a multiple of 8 bytes
                moff = int.from_bytes(d[q:q + 4], "little")
                q += 4
                if ver == 1:
                    q += 1 + 3 + 4 + 4 + 16        # dimensionality, reserved, permutation, reserved, 4 dimension sizes
            else:
                nb = 1
                while (1 << (8 * nb)) <= size and nb < 8:
                    nb += 1
                moff = int.from_bytes(d[q:q + nb], "little")
                q += nb
            mt, p = _parse_type(d, q)
            if mt.vlen_str:
                raise Unsupported("variable-length string inside a compound type")
            names.append(name)
            formats.append(mt.dtype)
            offsets.append(moff)
        return _Type(np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": size}), size=size), p
    if cls == 7:                                   # reference
        return _Type(np.dtype("V%d" % size), size=size), p
    if cls == 8:                                   # enumeration: base type, names, values -> read as the base type
        nmemb = bits & 0xFFFF
        base, p = _parse_type(d, p)
        for _ in range(nmemb):
            _, q = _cstr(d, p)
            p = q if ver >= 3 else p + (q - p + 7) // 8 * 8
        return _Type(base.dtype, size=size), p + nmemb * base.size
    if cls == 9:                                   # variable length
        base, p = _parse_type(d, p)
        if (bits & 0x0F) == 1:
            return _Type(np.dtype("O"), vlen_str=True, size=size), p
        raise Unsupported("variable-length sequence type")
    if cls == 10:                                  # array
        rank = d[p]
        p += 4 if ver < 3 else 1
        dims = [int.from_bytes(d[p + 4 * i:p + 4 * i + 4], "little") for i in range(rank)]
        p += 4 * rank
        if ver < 3:
            p += 4 * rank                          # permutation indices
        base, p = _parse_type(d, p)
        return _Type(np.dtype((base.dtype, tuple(dims))), size=size), p
    raise Unsupported("datatype class %d" % cls)


def _parse_space(d: bytes, pos: int, sl: int) -> Tuple[int, ...]:
    ver, rank, flags = d[pos], d[pos + 1], d[pos + 2]
    if ver == 1:
        p = pos + 8
    elif ver == 2:
        if d[pos + 3] == 2:                        # null dataspace
            return (0,)
        p = pos + 4
    else:
        raise Unsupported("dataspace message version %d" % ver)
    return tuple(int.from_bytes(d[p + i * sl:p + (i + 1) * sl], "little") for i in range(rank))


# ------------------------------------------------------------------------------------------------ objects
class _Messages:
    """(type, data bytes, file offset of the data) of every header message of one object."""

    def __init__(self, f: "File", addr: int):
        b = f._b
        self.items: List[Tuple[int, bytes]] = []
        self.shared = set()       # message types whose body is a shared-message reference (header message flag bit 1)
        d = b.d
        if d[addr:addr + 4] == b"OHDR":
            self._v2(f, addr)
            return
        if d[addr] != 1:
            raise Unsupported("object header version %d" % d[addr])
        nmsg = b.u(addr + 2, 2)
        size = b.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(self.items) < nmsg + 64:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(self.items) < nmsg:
                mtype, msize = b.u(p, 2), b.u(p + 2, 2)
                body = d[p + 8:p + 8 + msize]
                if mtype == 0x10:
                    blocks.append((b.off(p + 8), b.length(p + 8 + b.so)))
                if d[p + 4] & 0x02:
                    self.shared.add(mtype)                         # body = a reference to a shared / committed message
                self.items.append((mtype, body))
                p += 8 + msize

    def _v2(self, f: "File", addr: int):
        b, d = f._b, f._b.d
        flags = d[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        csize_bytes = 1 << (flags & 3)
        csize = b.u(p, csize_bytes)
        p += csize_bytes
        blocks = [(p, csize)]
        track = bool(flags & 0x04)
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 4 <= end:
                mtype, msize = d[p], b.u(p + 1, 2)
                if d[p + 3] & 0x02:
                    self.shared.add(mtype)
                p += 4 + (2 if track else 0)
                if p + msize > end:
                    break
                body = d[p:p + msize]
                if mtype == 0x10:
                    caddr, clen = b.off(p), b.length(p + b.so)
                    if d[caddr:caddr + 4] != b"OCHK":
                        raise ValueError("bad object header continuation block")
                    blocks.append((caddr + 4, clen - 8))          # signature in front, checksum behind
                self.items.append((mtype, body))
                p += msize

    def _not_shared(self, mtype: int) -> None:
        if mtype in self.shared:
            raise Unsupported("shared header message (type 0x%02x: a committed datatype or a shared-message-table entry)" % mtype)

    def first(self, mtype: int) -> Optional[bytes]:
        self._not_shared(mtype)
        for t, body in self.items:
            if t == mtype:
                return body
        return None

    def all(self, mtype: int) -> Iterator[bytes]:
        self._not_shared(mtype)
        return (body for t, body in self.items if t == mtype)


class _Attrs:
    def __init__(self, f: "File", msgs: _Messages):
        self._f, self._raw = f, {}
        if msgs.first(0x15) is not None:
            info = msgs.first(0x15)
            flags = info[1]
            p = 2 + (2 if flags & 1 else 0)
            if not f._b.undefined(int.from_bytes(info[p:p + f._b.so], "little")):
                raise Unsupported("attributes stored densely (fractal heap)")
        for body in msgs.all(0x0C):
            ver = body[0]
            nlen, tlen, slen = (int.from_bytes(body[2 + 2 * i:4 + 2 * i], "little") for i in range(3))
            p = 8 if ver < 3 else 9
            pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
            name = body[p:p + nlen].split(b"\x00")[0].decode("utf-8")
            p += pad(nlen)
            tpos = p
            p += pad(tlen)
            spos = p
            p += pad(slen)
            self._raw[name] = (body, tpos, spos, p)

    def __contains__(self, name: str) -> bool:
        return name in self._raw

    def keys(self):
        return self._raw.keys()

    def __getitem__(self, name: str):
        body, tpos, spos, dpos = self._raw[name]
        typ, _ = _parse_type(body, tpos)
        shape = _parse_space(body, spos, self._f._b.sl)
        n = int(np.prod(shape)) if shape else 1
        if typ.vlen_str:
            vals = [self._f._vlen_string(body, dpos + i * (8 + self._f._b.so)) for i in range(n)]
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        arr = np.frombuffer(body, dtype=typ.dtype, count=n, offset=dpos)
        return arr[0] if not shape else arr.reshape(shape).copy()


class _Object:
    def __init__(self, f: "File", addr: int, name: str):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = _Messages(f, addr)
        self.attrs = _Attrs(f, self._msgs)


class Dataset(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        m = self._msgs
        self._type, _ = _parse_type(m.first(0x03), 0)
        self.shape = _parse_space(m.first(0x01), 0, f._b.sl)
        self.dtype = self._type.dtype

    def _filters(self) -> List[Tuple[int, List[int]]]:
        body = self._msgs.first(0x0B)
        if body is None:
            return []
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(body[p:p + 2], "little")
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(body[p:p + 2], "little")
                p += 2
            p += 2                                     # flags
            nvals = int.from_bytes(body[p:p + 2], "little")
            p += 2
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            vals = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(nvals)]
            p += 4 * nvals
            if ver == 1 and nvals % 2:
                p += 4
            out.append((fid, vals))
        return out

    def _unfilter(self, chunk: bytes, mask: int, filters) -> bytes:
        for idx in range(len(filters) - 1, -1, -1):    # the pipeline is undone in reverse order
            if mask & (1 << idx):
                continue
            fid, vals = filters[idx]
            if fid == 1:
                chunk = zlib.decompress(chunk)
            elif fid == 2:
                esize = vals[0] if vals else self._type.size
                a = np.frombuffer(chunk, dtype=np.uint8)
                n = len(a) // esize
                chunk = a[:n * esize].reshape(esize, n).T.tobytes() + a[n * esize:].tobytes()
            elif fid == 3:
                chunk = chunk[:-4]                     # fletcher32 checksum behind the data
            elif fid == 32020:
                raise Unsupported("the VBZ signal compression filter (id 32020) of recent ONT fast5 files")
            else:
                raise Unsupported("dataset filter id %d" % fid)
        return chunk

    def _filled(self) -> np.ndarray:
        """An array of the dataset's shape holding its fill value: what HDF5 returns for storage that was never written
        (an undefined data address, a chunk absent from the index). Fill value message 0x05, versions 1 - 3; the old fill
        message 0x04 is raised as unsupported rather than guessed at."""
        out = np.zeros(self.shape, self.dtype)
        body = self._msgs.first(0x05)
        if body is None:
            if self._msgs.first(0x04) is not None:
                raise Unsupported("old-style fill value message (0x04) on a dataset with unwritten storage")
            return out
        ver = body[0]
        if ver in (1, 2):
            defined = ver == 1 or body[3]
            if not defined or len(body) < 8:
                return out
            size, p = int.from_bytes(body[4:8], "little"), 8
        elif ver == 3:
            if not body[1] & 0x20:
                return out
            size, p = int.from_bytes(body[2:6], "little"), 6
        else:
            raise Unsupported("fill value message version %d" % ver)
        if size == 0:
            return out
        if size != self.dtype.itemsize:
            raise Unsupported("fill value of %d bytes for a %d-byte element type" % (size, self.dtype.itemsize))
        out[...] = np.frombuffer(body, dtype=self.dtype, count=1, offset=p)[0]
        return out

    def _read_all(self) -> np.ndarray:
        f, b, d = self._f, self._f._b, self._f._b.d
        if self._type.vlen_str:
            raise Unsupported("dataset of variable-length strings")
        body = self._msgs.first(0x08)
        n = int(np.prod(self.shape)) if self.shape else 1
        isz = self.dtype.itemsize
        ver = body[0]
        if ver in (1, 2):
            rank, cls = body[1], body[2]
            p = 8
            addr = None
            if cls != 0:
                addr = int.from_bytes(body[p:p + b.so], "little")
                p += b.so
            dims = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(rank)]
            p += 4 * rank
            if cls == 0:
                size = int.from_bytes(body[p:p + 4], "little")
                return np.frombuffer(body, dtype=self.dtype, count=n, offset=p + 4).reshape(self.shape).copy()
            if cls == 1:
                if b.undefined(addr):
                    return self._filled()                      # never written
                return np.frombuffer(d, dtype=self.dtype, count=n, offset=addr).reshape(self.shape).copy()
            chunk_dims, btree = dims, addr                     # rank counts the element-size dimension, dims hold the chunk shape
            p_es = p
            esize = int.from_bytes(body[p_es:p_es + 4], "little")
            chunk_dims = dims + [esize] if len(dims) == len(self.shape) else dims
        elif ver in (3, 4):                            # version 4 (HDF5 1.10) keeps the compact / contiguous forms of version 3
            cls = body[1]
            if cls == 0:
                size = int.from_bytes(body[2:4], "little")
                return np.frombuffer(body, dtype=self.dtype, count=n, offset=4).reshape(self.shape).copy()
            if cls == 1:
                addr = int.from_bytes(body[2:2 + b.so], "little")
                if b.undefined(addr):
                    return self._filled()                      # never written
                return np.frombuffer(d, dtype=self.dtype, count=n, offset=addr).reshape(self.shape).copy()
            if cls != 2:
                raise Unsupported("data layout class %d" % cls)
            if ver == 4:
                raise Unsupported("chunked dataset with a version-4 chunk index (HDF5 1.10 `libver=latest`)")
            rank1 = body[2]
            btree = int.from_bytes(body[3:3 + b.so], "little")
            p = 3 + b.so
            chunk_dims = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], "little") for i in range(rank1)]
        else:
            raise Unsupported("data layout message version %d" % ver)
        # ---- chunked: walk the version-1 B-tree of chunks
        cshape = tuple(chunk_dims[:-1])
        out = self._filled()                           # chunks absent from the index keep the fill value
        if b.undefined(btree):
            return out
        filters = self._filters()
        rank = len(cshape)
        for offs, csize, mask, caddr in f._chunks(btree, rank):
            raw = self._unfilter(d[caddr:caddr + csize], mask, filters)
            chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cshape))).reshape(cshape)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, self.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = chunk[sel_in]
        return out

    def __getitem__(self, key):
        arr = self._read_all()
        if isinstance(key, str):
            return arr[key]
        if key == ():
            return arr if self.shape else arr.reshape(())[()]
        return arr[key]

    def __len__(self):
        return self.shape[0]


class Group(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        self._links: Optional[Dict[str, int]] = None

    def _load(self) -> Dict[str, int]:
        if self._links is not None:
            return self._links
        f, b, m = self._f, self._f._b, self._msgs
        links: Dict[str, int] = {}
        st = m.first(0x11)
        if st is not None:                             # symbol table: B-tree v1 of SNOD nodes, names in a local heap
            btree, heap = int.from_bytes(st[:b.so], "little"), int.from_bytes(st[b.so:2 * b.so], "little")
            links.update(f._symbol_table(btree, heap))
        info = m.first(0x02)
        if info is not None:
            flags = info[1]
            p = 2 + (8 if flags & 1 else 0)
            if not b.undefined(int.from_bytes(info[p:p + b.so], "little")):
                raise Unsupported("group with densely stored links (fractal heap)")
        for body in m.all(0x06):                       # compact new-style group: one link message per member
            flags = body[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = body[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            nb = 1 << (flags & 3)
            nlen = int.from_bytes(body[p:p + nb], "little")
            p += nb
            name = body[p:p + nlen].decode("utf-8")
            p += nlen
            if ltype == 0:
                links[name] = int.from_bytes(body[p:p + b.so], "little")
        self._links = links
        return links

    def keys(self):
        return list(self._load().keys())

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, path: str) -> bool:
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._load()
            if part not in links:
                raise KeyError("%s: no member %r (members: %s)" % (node.name or "/", part, sorted(links)))
            node = node._f._open(links[part], (node.name.rstrip("/") + "/" + part))
        return node


class File(Group):
    """`File(path)` -- read-only; usable as a context manager like h5py.File."""

    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise ValueError("minihdf5 is read-only")
        with open(path, "rb") as fh:
            data = fh.read()
        base = 0
        while data[base:base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2
            if base + 8 > len(data):
                raise ValueError("%s: not an HDF5 file" % path)
        ver = data[base + 8]
        if ver in (0, 1):
            so, sl = data[base + 13], data[base + 14]
            self._b = _Buf(data, so, sl)
            p = base + 24 + (4 if ver == 1 else 0)
            self._base = self._b.off(p)
            root_entry = p + 4 * so
            root = self._b.off(root_entry + so)
        elif ver in (2, 3):
            so, sl = data[base + 9], data[base + 10]
            self._b = _Buf(data, so, sl)
            self._base = self._b.off(base + 12)
            root = self._b.off(base + 12 + 3 * so)
        else:
            raise Unsupported("superblock version %d" % ver)
        if self._base != 0 or base != 0:
            raise Unsupported("a user block / non-zero base address")
        self._cache: Dict[int, _Object] = {}
        self.filename = path
        Group.__init__(self, self, root, "/")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    # ---- helpers over the raw file image
    def _open(self, addr: int, name: str) -> _Object:
        if addr not in self._cache:
            msgs = _Messages(self, addr)
            is_dataset = msgs.first(0x08) is not None and msgs.first(0x03) is not None
            self._cache[addr] = (Dataset if is_dataset else Group)(self, addr, name)
        return self._cache[addr]

    def _symbol_table(self, btree: int, heap: int) -> Dict[str, int]:
        b, d = self._b, self._b.d
        if d[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        heap_data = b.off(heap + 8 + 2 * b.sl)
        out: Dict[str, int] = {}

        def walk(addr: int):
            sig = d[addr:addr + 4]
            if sig == b"TREE":
                level, used = d[addr + 5], b.u(addr + 6, 2)
                p = addr + 8 + 2 * b.so
                for i in range(used):
                    p += b.sl                           # key i
                    walk(b.off(p))
                    p += b.so
            elif sig == b"SNOD":
                nsym = b.u(addr + 6, 2)
                p = addr + 8
                for _ in range(nsym):
                    noff, oaddr = b.off(p), b.off(p + b.so)
                    name, _ = _cstr(d, heap_data + noff)
                    out[name] = oaddr
                    p += 2 * b.so + 24
            else:
                raise ValueError("bad group B-tree node signature %r" % sig)

        walk(btree)
        return out

    def _chunks(self, btree: int, rank: int):
        """(element offsets, stored size, filter mask, address) of every chunk under a version-1 chunk B-tree."""
        b, d = self._b, self._b.d
        keysize = 8 + 8 * (rank + 1)

        def walk(addr: int):
            if d[addr:addr + 4] != b"TREE" or d[addr + 4] != 1:
                raise ValueError("bad chunk B-tree node")
            level, used = d[addr + 5], b.u(addr + 6, 2)
            p = addr + 8 + 2 * b.so
            for _ in range(used):
                csize, mask = b.u(p, 4), b.u(p + 4, 4)
                offs = tuple(b.u(p + 8 + 8 * i, 8) for i in range(rank))
                child = b.off(p + keysize)
                if level == 0:
                    yield offs, csize, mask, child
                else:
                    yield from walk(child)
                p += keysize + b.so

        yield from walk(btree)

    def _vlen_string(self, buf: bytes, pos: int) -> str:
        b, d = self._b, self._b.d
        length = int.from_bytes(buf[pos:pos + 4], "little")
        gaddr = int.from_bytes(buf[pos + 4:pos + 4 + b.so], "little")
        index = int.from_bytes(buf[pos + 4 + b.so:pos + 8 + b.so], "little")
        if length == 0 or gaddr == 0:
            return ""
        if d[gaddr:gaddr + 4] != b"GCOL":
            raise ValueError("bad global heap collection signature")
        csize = b.length(gaddr + 8)
        p = gaddr + 8 + b.sl
        end = gaddr + csize
        while p + 8 + b.sl <= end:
            idx = b.u(p, 2)
            osize = b.length(p + 8)
            if idx == 0:
                break
            if idx == index:
                return d[p + 8 + b.sl:p + 8 + b.sl + length].decode("utf-8")
            p += 8 + b.sl + (osize + 7) // 8 * 8
        raise KeyError("global heap object %d not found" % index)
