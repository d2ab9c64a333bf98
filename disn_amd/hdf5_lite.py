"""Read-only HDF5 for the files the reference's loaders open (data/data_sdf_h5_queue.py:121-186: per-object SDF samples
``ori_sample.h5`` and per-view image files ``%02d.h5``, written by preprocessing/create_point_sdf_grid.py and
preprocessing/create_img_h5.py with ``h5py.File(...).create_dataset(name, data=..., compression='gzip')``).

h5py is not part of this environment, and the loader needs very little of HDF5: the root group's datasets, N-d arrays of
fixed-size numbers, contiguous / compact / chunked layouts, the deflate (gzip) and shuffle filters.  This module restates
exactly that subset from the HDF5 File Format Specification (version 1.x structures, which is what libhdf5 writes for
``libver='earliest'`` -- h5py's default):

    superblock v0 / v1          -> root group symbol table entry (object header address, B-tree + local heap addresses)
    group: v1 B-tree "TREE" (node type 0) over symbol table nodes "SNOD", names in the local heap "HEAP"
    object header v1            -> messages: dataspace 0x0001, datatype 0x0003, data layout 0x0008 (v3),
                                   filter pipeline 0x000B, continuation 0x0010, symbol table 0x0011 (sub-groups)
    chunked data: v1 B-tree (node type 1), keys = (chunk size, filter mask, offsets), deflate / shuffle per chunk

Not supported (raises NotImplementedError with the structure's name): superblock v2 / v3 and new-style groups
(``libver='latest'``), variable-length / compound / string types, other filters (szip, lzf), external / virtual storage.
STATUS: pinned by a file assembled BY HAND from the specification in tests/test_data.py (contiguous, chunked + deflate,
chunked + shuffle + deflate, edge chunks, a sub-group, a continuation block); not validated against a file written by
libhdf5 itself (none here).  If h5py is importable the loader prefers it.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5File:
    """f = Hdf5File(path); f.keys(); f[name] -> numpy array (datasets of the root group; 'group/name' for sub-groups)"""

    def __init__(self, path: str):
        self.buf = open(path, "rb").read()
        b = self.buf
        base = b.find(SIGNATURE)
        if base != 0:
            raise ValueError("%s: not an HDF5 file (signature at offset %d)" % (path, base))
        if len(b) < 96:
            raise ValueError("%s: truncated HDF5 file (%d bytes)" % (path, len(b)))
        ver = b[8]
        if ver > 1:
            raise NotImplementedError("HDF5 superblock version %d (written with libver='latest'); only v0 / v1" % ver)
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("HDF5 with %d-byte offsets / %d-byte lengths" % (b[13], b[14]))
        pos = 24 + (4 if ver == 1 else 0)       # after group K values and consistency flags (+ v1: indexed-storage K)
        self.base_address = struct.unpack_from("<Q", b, pos)[0]
        root_entry = pos + 32                    # base, free-space, end-of-file, driver-info addresses
        _, hdr_addr, cache_type = struct.unpack_from("<QQI", b, root_entry)
        self._links: Dict[str, int] = {}
        if cache_type == 1:                      # scratch pad: B-tree and heap of the root group
            btree, heap = struct.unpack_from("<QQ", b, root_entry + 24)
            self._walk_group(btree, heap, "")
        else:
            self._group_from_header(hdr_addr, "")

    # ------------------------------------------------------------------ groups
    def _group_from_header(self, addr: int, prefix: str) -> None:
        for mtype, data in self._messages(addr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", data, 0)
                self._walk_group(btree, heap, prefix)
                return
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style group (link info / link messages): written with libver='latest'")

    def _heap_data(self, heap: int) -> int:
        b = self.buf
        if b[heap:heap + 4] != b"HEAP":
            raise ValueError("local heap signature missing at %d" % heap)
        return struct.unpack_from("<Q", b, heap + 24)[0]

    def _walk_group(self, node: int, heap: int, prefix: str) -> None:
        b = self.buf
        if b[node:node + 4] != b"TREE":
            raise ValueError("B-tree signature missing at %d" % node)
        ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
        if ntype != 0:
            raise ValueError("group B-tree node of type %d" % ntype)
        pos = node + 24                          # after the two sibling addresses
        for i in range(used):
            child = struct.unpack_from("<Q", b, pos + 8)[0]     # key i (8 bytes), then child i
            pos += 16
            if level > 0:
                self._walk_group(child, heap, prefix)
            else:
                self._symbol_node(child, heap, prefix)

    def _symbol_node(self, addr: int, heap: int, prefix: str) -> None:
        b = self.buf
        if b[addr:addr + 4] != b"SNOD":
            raise ValueError("symbol table node signature missing at %d" % addr)
        nsym = struct.unpack_from("<H", b, addr + 6)[0]
        data = self._heap_data(heap)
        for i in range(nsym):
            e = addr + 8 + 40 * i
            name_off, hdr, cache_type = struct.unpack_from("<QQI", b, e)
            end = b.index(b"\x00", data + name_off)
            name = prefix + b[data + name_off:end].decode("utf-8")
            if cache_type == 1:                  # a sub-group whose B-tree / heap are cached in the entry
                bt, hp = struct.unpack_from("<QQ", b, e + 24)
                self._walk_group(bt, hp, name + "/")
                continue
            sub = [d for t, d in self._messages(hdr) if t == 0x0011]
            if sub:                              # a sub-group known only by its header's symbol table message
                bt, hp = struct.unpack_from("<QQ", sub[0], 0)
                self._walk_group(bt, hp, name + "/")
            else:
                self._links[name] = hdr

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr: int) -> List[Tuple[int, bytes]]:
        b = self.buf
        if b[addr] != 1:
            raise NotImplementedError("object header version %d (only v1; v2 'OHDR' comes with libver='latest')" % b[addr])
        nmsg, _refs, size = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, size)]             # 12-byte prefix padded to 16
        out: List[Tuple[int, bytes]] = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                data = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:              # continuation: more messages elsewhere
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((off, ln))
                out.append((mtype, data))
        return out

    # ------------------------------------------------------------------ datasets
    def keys(self) -> List[str]:
        return sorted(self._links)

    def __contains__(self, name: str) -> bool:
        return name in self._links

    def __getitem__(self, name: str) -> np.ndarray:
        if name not in self._links:
            raise KeyError(name)
        shape = dtype = layout = None
        filters: List[Tuple[int, List[int]]] = []
        for mtype, d in self._messages(self._links[name]):
            if mtype == 0x0001:
                shape = _dataspace(d)
            elif mtype == 0x0003:
                dtype = _datatype(d)
            elif mtype == 0x0008:
                layout = d
            elif mtype == 0x000B:
                filters = _filters(d)
            elif mtype == 0x0011:
                raise KeyError("%s is a group" % name)
        if shape is None or dtype is None or layout is None:
            raise ValueError("%s: not a dataset (dataspace / datatype / layout message missing)" % name)
        return self._read(shape, dtype, layout, filters, name)

    def _read(self, shape, dtype, layout: bytes, filters, name: str) -> np.ndarray:
        b = self.buf
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] != 3:
            raise NotImplementedError("%s: data layout message version %d (only v3)" % (name, layout[0]))
        cls = layout[1]
        if cls == 0:                             # compact: the data sits in the message
            size = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + size]
            return np.frombuffer(raw, dtype, count).reshape(shape).copy()
        if cls == 1:                             # contiguous
            addr, size = struct.unpack_from("<QQ", layout, 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype)
            return np.frombuffer(b, dtype, count, addr).reshape(shape).copy()
        if cls != 2:
            raise NotImplementedError("%s: layout class %d" % (name, cls))
        ndim = layout[2]                         # rank + 1 (the last 'dimension' is the element size)
        btree = struct.unpack_from("<Q", layout, 3)[0]
        cdims = struct.unpack_from("<%dI" % ndim, layout, 11)
        chunk = tuple(cdims[:-1])
        if len(chunk) != len(shape) or cdims[-1] != dtype.itemsize:
            raise ValueError("%s: chunk dimensionality does not match the dataspace" % name)
        out = np.zeros(shape, dtype)
        if btree != UNDEF:
            for offs, fmask, addr, size in self._chunks(btree, ndim):
                raw = b[addr:addr + size]
                for k in range(len(filters) - 1, -1, -1):      # undo the pipeline, last filter first
                    fid, cd = filters[k]
                    if fmask & (1 << k):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        raw = _unshuffle(raw, cd[0] if cd else dtype.itemsize)
                    elif fid == 3:
                        raw = raw[:-4]           # fletcher32 checksum appended
                    else:
                        raise NotImplementedError("%s: HDF5 filter id %d" % (name, fid))
                blk = np.frombuffer(raw, dtype, int(np.prod(chunk, dtype=np.int64))).reshape(chunk)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks(self, node: int, ndim: int):
        b = self.buf
        if b[node:node + 4] != b"TREE":
            raise ValueError("chunk B-tree signature missing at %d" % node)
        ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
        if ntype != 1:
            raise ValueError("chunk B-tree node of type %d" % ntype)
        ksize = 8 + 8 * ndim
        pos = node + 24
        for i in range(used):
            size, fmask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from("<%dQ" % ndim, b, pos + 8)
            child = struct.unpack_from("<Q", b, pos + ksize)[0]
            pos += ksize + 8
            if level > 0:
                yield from self._chunks(child, ndim)
            else:
                yield offs[:-1], fmask, child, size


def _dataspace(d: bytes) -> Tuple[int, ...]:
    ver, rank = d[0], d[1]
    if ver == 1:
        pos = 8
    elif ver == 2:
        pos = 4
    else:
        raise NotImplementedError("dataspace message version %d" % ver)
    return tuple(struct.unpack_from("<%dQ" % rank, d, pos)) if rank else ()


def _datatype(d: bytes) -> np.dtype:
    cls, bits0 = d[0] & 0x0F, d[1]
    size = struct.unpack_from("<I", d, 4)[0]
    order = ">" if bits0 & 1 else "<"
    if cls == 1:                                 # floating point
        if size not in (2, 4, 8):
            raise NotImplementedError("%d-byte float" % size)
        return np.dtype("%sf%d" % (order, size))
    if cls == 0:                                 # fixed point; bit 3: signed
        if size not in (1, 2, 4, 8):
            raise NotImplementedError("%d-byte integer" % size)
        return np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", size))
    raise NotImplementedError("HDF5 datatype class %d (only fixed- and floating-point numbers)" % cls)


def _filters(d: bytes) -> List[Tuple[int, List[int]]]:
    ver, n = d[0], d[1]
    pos = 8 if ver == 1 else 2
    out = []
    for _ in range(n):
        fid = struct.unpack_from("<H", d, pos)[0]
        pos += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = struct.unpack_from("<H", d, pos)[0]
            pos += 2
        _flags, ncd = struct.unpack_from("<HH", d, pos)
        pos += 4
        if nlen:
            pos += (nlen + 7) // 8 * 8 if ver == 1 else nlen
        cd = list(struct.unpack_from("<%dI" % ncd, d, pos))
        pos += 4 * ncd
        if ver == 1 and ncd % 2:
            pos += 4
        out.append((fid, cd))
    return out


def _unshuffle(raw: bytes, elem: int) -> bytes:
    n = len(raw) // elem
    if elem <= 1 or n == 0:
        return raw
    a = np.frombuffer(raw, np.uint8, n * elem).reshape(elem, n)
    return a.T.tobytes() + raw[n * elem:]
