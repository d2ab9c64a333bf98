"""TensorFlow-free reader / writer for ``tf.train.Saver`` V2 checkpoints (tensor bundles).

The reference saves and restores through ``tf.train.Saver`` (train/train_sdf.py:285-286,322-328;
test/create_sdf.py:180-192) and ships its weights as V2 bundles (``checkpoint/SDF_DISN``,
``vgg_16.ckpt``; README.md:27-39).  "Keep the checkpoint layout" therefore means: the files

    <prefix>.index                  a LevelDB-style sorted string table (TF lib/io/table):
                                    key ""   -> BundleHeaderProto {num_shards, endianness, version}
                                    key name -> BundleEntryProto  {dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001    the raw little-endian tensor bytes, back to back
    checkpoint                      text proto: model_checkpoint_path / all_model_checkpoint_paths

This module restates that format from the TensorFlow source (tensorflow/core/util/tensor_bundle,
tensorflow/core/lib/io/{table_builder,block_builder,format}.cc, protobuf wire format) in plain
Python.  STATUS: round-trip tested, and pinned by a bundle assembled BY HAND from the format description,
independently of this module (tests/test_tf_checkpoint.py: the reader reads it, the writer reproduces it byte for
byte, shortened index keys included); NOT validated against a file written by TensorFlow itself (none is available
in this environment -- the reference's checkpoints are Dropbox downloads).
Only what DISN checkpoints contain is supported: uncompressed blocks, one shard, float32/int32/
int64/float64 tensors, no tensor slices.

V1 checkpoints (``load_checkpoint_v1``; ``load_checkpoint`` / ``list_variables`` detect the format).  The training
recipe starts from TF-slim's ``vgg_16.ckpt`` (README.md:128; train/train_sdf.py:190-219 lists and restores it through
``checkpoint_utils`` / ``tf.train.Saver``, which read both formats).  That 2016 file is a SINGLE-FILE V1 checkpoint
(``tf.train.Saver`` wrote V1 until TF 0.12), i.e. one TF table -- the same table format as the V2 ``.index`` -- written
by tensorflow/core/util/tensor_slice_writer.cc (``table::Options::compression = kNoCompression``):

    key ""                                 -> SavedTensorSlices { meta = 1: SavedTensorSliceMeta {
                                                 repeated SavedSliceMeta tensor = 1 { name = 1, shape = 2, type = 3,
                                                 repeated TensorSliceProto slice = 4 }, versions = 2 } }
    key OrderedCode(0, name, slice extents) -> SavedTensorSlices { data = 2: SavedSlice { name = 1,
                                                 TensorSliceProto slice = 2, TensorProto data = 3 } }

with the values in the TensorProto's typed repeated fields (float_val = 5, double_val = 6, int_val = 7,
int64_val = 10; packed) or tensor_content = 4.  The reader takes name and slice from the VALUE (the ordered-code key is
only checked for its leading 0), assembles slices of partitioned variables into the full tensor, and is pinned by a file
assembled by hand from this description (tests/test_tf_checkpoint.py).  STATUS: like the V2 code, not validated against
a file written by TensorFlow (none here); if slim's file turns out to be compressed (kSnappyCompression) the reader
says so instead of guessing.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57          # tensorflow/core/lib/io/format.h kTableMagicNumber
FOOTER_LEN = 48                            # 2 x BlockHandle::kMaxEncodedLength (20) + 8
BLOCK_TRAILER = 5                          # 1 byte compression type + 4 byte masked crc32c
DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}      # types.proto DataType
DT_INV = {np.dtype(v): k for k, v in DT.items()}

# ---------------------------------------------------------------- crc32c (Castagnoli), masked
_CRC_TABLE: List[int] = []


def _crc_table() -> List[int]:
    if not _CRC_TABLE:
        poly = 0x82F63B78
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def _crc32c_py(data: bytes, crc: int = 0) -> int:
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C (Castagnoli).  Large buffers go through the library's host helper disn_crc32c
    (SSE4.2 instruction); the pure-Python loop is the definition and is used for small blocks."""
    if len(data) >= 4096:
        try:
            from ._lib import lib
            import ctypes as C
            # no copy of the buffer (a 411 MB block of slim's vgg_16.ckpt): numpy views bytes / memoryview / ndarray in place
            a = (np.ascontiguousarray(data) if isinstance(data, np.ndarray) else np.frombuffer(data, np.uint8)).view(np.uint8)
            return int(lib().disn_crc32c(C.c_void_p(a.ctypes.data), a.size, crc))
        except Exception:  # library not built: fall back to the definition
            pass
    return _crc32c_py(bytes(data), crc)


def mask_crc(c: int) -> int:
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right 15 and add a constant."""
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf wire format
def _put_varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = n = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


_PB_VIEW_MIN = 1 << 16


def _pb_fields(buf) -> Iterable[Tuple[int, int, object]]:
    """yield (field_number, wire_type, value) -- value is int for varint/fixed, bytes (a memoryview from 64 KiB on) for
    len-delimited"""
    pos = 0
    mv = memoryview(buf)
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            # small values (names, shapes, nested headers) as bytes; large ones (tensor payloads and the messages around
            # them: fc6 of slim's VGG-16 is 411 MB, nested three deep) as views of the caller's buffer -- no copies
            v = mv[pos:pos + ln] if ln >= _PB_VIEW_MIN else bytes(mv[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _pb_varint(fn: int, v: int) -> bytes:
    return _put_varint(fn << 3) + _put_varint(v)


def _pb_bytes(fn: int, v: bytes) -> bytes:
    return _put_varint((fn << 3) | 2) + _put_varint(len(v)) + v


def _encode_shape(shape: Tuple[int, ...]) -> bytes:          # TensorShapeProto: repeated Dim dim = 2
    return b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)


def _decode_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for fn, wt, v in _pb_fields(buf):
        if fn == 2 and wt == 2:
            size = 0
            for f2, w2, v2 in _pb_fields(v):
                if f2 == 1 and w2 == 0:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _encode_entry(dtype: int, shape, shard: int, offset: int, size: int, crc: int) -> bytes:
    """BundleEntryProto (tensor_bundle.proto): dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)"""
    out = _pb_varint(1, dtype) + _pb_bytes(2, _encode_shape(shape))
    if shard:
        out += _pb_varint(3, shard)
    if offset:
        out += _pb_varint(4, offset)
    out += _pb_varint(5, size) + _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def _decode_entry(buf: bytes) -> Dict[str, object]:
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            e["shape"] = _decode_shape(v)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
        elif fn == 7:
            e["slices"] += 1
    return e


# ---------------------------------------------------------------- table blocks
def _read_block(buf, offset: int, size: int, verify: bool):
    mv = memoryview(buf)
    body = mv[offset:offset + size]                    # a view: the block is never copied
    trailer = bytes(mv[offset + size:offset + size + BLOCK_TRAILER])
    if len(trailer) != BLOCK_TRAILER:
        raise ValueError("truncated table block")
    if trailer[0] != 0:
        raise NotImplementedError("compressed table block (type %d); TF bundles are written uncompressed" % trailer[0])
    if verify:
        want = struct.unpack("<I", trailer[1:])[0]
        got = mask_crc(crc32c(trailer[:1], crc32c(body)))    # the crc of body + type byte, chained
        if want != got:
            raise ValueError("table block crc mismatch at %d" % offset)
    return body


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    """block_builder.cc: entries (shared varint, non_shared varint, value_len varint, key delta,
    value) ... restarts uint32[n], n uint32"""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    out, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        val = block[pos:pos + vlen]                    # (a view when the block is one)
        out.append((key, val if vlen >= _PB_VIEW_MIN else bytes(val)))
        pos += vlen
    return out


def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval: int) -> bytes:
    out = bytearray()
    restarts, last = [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _shortest_separator(start: bytes, limit: bytes) -> bytes:
    """BytewiseComparator::FindShortestSeparator (lib/io/table_builder.cc calls it for the index key between two
    data blocks): a short key k with start <= k < limit"""
    m = min(len(start), len(limit))
    d = 0
    while d < m and start[d] == limit[d]:
        d += 1
    if d < m and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _short_successor(key: bytes) -> bytes:
    """BytewiseComparator::FindShortSuccessor (index key of the LAST data block): first byte that is not 0xff
    incremented, the rest dropped"""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _emit_block(f, block: bytes) -> Tuple[int, int]:
    off = f.tell()
    f.write(block)
    f.write(b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
    return off, len(block)


# ---------------------------------------------------------------- V1 (single-file, tensor_slice_writer.cc)
def _table_entries(buf: bytes, verify: bool) -> Iterable[Tuple[bytes, bytes]]:
    """every (key, value) of a TF table held in ``buf``, in key order"""
    if len(buf) < FOOTER_LEN or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("not a TensorFlow table (bad magic)")
    footer = buf[-FOOTER_LEN:]
    pos = 0
    _mi_off, pos = _get_varint(footer, pos)
    _mi_size, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        for kv in _block_entries(_read_block(buf, off, size, verify)):
            yield kv


def _decode_slice(buf: bytes) -> List[Tuple[int, int]]:
    """TensorSliceProto: repeated Extent extent = 1 { int64 start = 1; oneof { int64 length = 2 } } -> [(start, length)],
    length -1 = the whole dimension (an Extent without a length)"""
    out = []
    for fn, wt, v in _pb_fields(buf):
        if fn == 1 and wt == 2:
            start, length = 0, -1
            for f2, w2, v2 in _pb_fields(v):
                if f2 == 1:
                    start = v2
                elif f2 == 2:
                    length = v2
            out.append((start, length))
    return out


def _packed_varints(buf: bytes, signed_bits: int) -> List[int]:
    out, pos = [], 0
    while pos < len(buf):
        v, pos = _get_varint(buf, pos)
        if v >= 1 << 63:
            v -= 1 << 64
        if signed_bits == 32:
            v = ((v + (1 << 31)) % (1 << 32)) - (1 << 31)
        out.append(v)
    return out


def _decode_tensor_proto(buf: bytes) -> Tuple[int, Tuple[int, ...], np.ndarray]:
    """TensorProto (tensor.proto) -> (dtype, shape, flat values): tensor_content, or the typed repeated field"""
    dtype, shape, content = 0, (), None
    typed: Dict[int, List] = {5: [], 6: [], 7: [], 10: []}
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            dtype = v
        elif fn == 2:
            shape = _decode_shape(v)
        elif fn == 4:
            content = v
        elif fn == 5:        # float_val: packed (wire type 2) or one fixed32 per element
            typed[5].append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(struct.pack("<I", v), "<f4"))
        elif fn == 6:
            typed[6].append(np.frombuffer(v, "<f8") if wt == 2 else np.frombuffer(struct.pack("<Q", v), "<f8"))
        elif fn == 7:
            typed[7].append(np.asarray(_packed_varints(v, 32) if wt == 2 else [((v + (1 << 31)) % (1 << 32)) - (1 << 31)], np.int32))
        elif fn == 10:
            typed[10].append(np.asarray(_packed_varints(v, 64) if wt == 2 else [v - (1 << 64) if v >= 1 << 63 else v], np.int64))
    if dtype not in DT:
        raise NotImplementedError("DataType %d" % dtype)
    dt = np.dtype(DT[dtype])
    if content is not None:
        vals = np.frombuffer(content, dtype=dt.newbyteorder("<"))
        if vals.dtype != dt:                             # a big-endian host
            vals = vals.astype(dt)
    else:
        field = {1: 5, 2: 6, 3: 7, 9: 10}[dtype]
        vals = np.concatenate(typed[field]).astype(dt) if typed[field] else np.zeros(0, dt)
    return dtype, shape, vals


def is_v1_checkpoint(path: str) -> bool:
    """a single FILE that is a TF table (V2 is <prefix>.index + <prefix>.data-*)"""
    if not os.path.isfile(path):
        return False
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        if f.tell() < FOOTER_LEN:
            return False
        f.seek(-8, os.SEEK_END)
        return struct.unpack("<Q", f.read(8))[0] == TABLE_MAGIC


def list_variables_v1(path: str, verify: bool = True, _buf: Optional[bytes] = None) -> Dict[str, Dict[str, object]]:
    """name -> {dtype, shape, slices} from the SavedTensorSliceMeta under key '' (`_buf`: the file's bytes when the
    caller has read them already -- slim's vgg_16.ckpt is 550 MB: it is read ONCE)"""
    buf = open(path, "rb").read() if _buf is None else _buf
    out: Dict[str, Dict[str, object]] = {}
    for k, v in _table_entries(buf, verify):
        if k != b"":
            break                               # keys are sorted: '' comes first
        for fn, wt, meta in _pb_fields(v):      # SavedTensorSlices.meta = 1
            if fn != 1 or wt != 2:
                continue
            for f2, w2, t in _pb_fields(meta):  # SavedTensorSliceMeta.tensor = 1
                if f2 != 1 or w2 != 2:
                    continue
                e = {"dtype": 0, "shape": (), "slices": 0}
                name = ""
                for f3, w3, x in _pb_fields(t):
                    if f3 == 1:
                        name = x.decode("utf-8")
                    elif f3 == 2:
                        e["shape"] = _decode_shape(x)
                    elif f3 == 3:
                        e["dtype"] = x
                    elif f3 == 4:
                        e["slices"] += 1
                out[name] = e
    if not out:
        raise ValueError("%s: no SavedTensorSliceMeta under key '' (not a V1 checkpoint?)" % path)
    return out


def load_checkpoint_v1(path: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """Read tensors of a single-file V1 checkpoint by variable name (all, or the ``names`` given)."""
    buf = open(path, "rb").read()
    want = set(names) if names is not None else None
    meta = list_variables_v1(path, verify, _buf=buf)
    out: Dict[str, np.ndarray] = {}
    filled: Dict[str, int] = {}
    for k, v in _table_entries(buf, verify):
        if k == b"":
            continue
        if k[:1] != b"\x00":                    # OrderedCode::WriteNumIncreasing(0) = one zero byte
            raise ValueError("V1 checkpoint key does not start with the ordered code of 0")
        for fn, wt, ss in _pb_fields(v):        # SavedTensorSlices.data = 2: SavedSlice
            if fn != 2 or wt != 2:
                continue
            name, extents, tp = "", [], None
            for f2, w2, x in _pb_fields(ss):
                if f2 == 1:
                    name = x.decode("utf-8")
                elif f2 == 2:
                    extents = _decode_slice(x)
                elif f2 == 3:
                    tp = x
            if want is not None and name not in want:
                continue
            if name not in meta or tp is None:
                raise ValueError("V1 checkpoint: slice of unknown tensor %r" % name)
            dtype, _, vals = _decode_tensor_proto(tp)
            full = tuple(int(d) for d in meta[name]["shape"])
            if dtype != meta[name]["dtype"]:
                raise ValueError("%s: slice dtype %d, meta dtype %d" % (name, dtype, meta[name]["dtype"]))
            if name not in out:
                out[name] = np.zeros(full, DT[dtype])
                filled[name] = 0
            idx = tuple(slice(None) if ln < 0 else slice(st, st + ln)
                        for (st, ln) in (extents + [(0, -1)] * (len(full) - len(extents))))
            target = out[name][idx] if full else out[name]
            if vals.size != target.size:
                raise ValueError("%s: slice holds %d values, its extent %d" % (name, vals.size, target.size))
            if full:
                out[name][idx] = vals.reshape(target.shape)
            else:
                out[name] = vals.reshape(()).astype(DT[dtype])
            filled[name] += int(vals.size)
    for name, n in filled.items():
        size = int(np.prod(meta[name]["shape"], dtype=np.int64)) if meta[name]["shape"] else 1
        if n != size:
            raise ValueError("%s: slices cover %d of %d elements" % (name, n, size))
    if want is not None and want - set(out):
        pass                                     # absent names are simply not returned (as load_checkpoint)
    return out


def _ordered_string(b: bytes) -> bytes:
    """OrderedCode::WriteString's escaping: 0x00 -> 00 ff, 0xff -> ff 00 (the terminator 00 01 is added by the caller)"""
    return b"".join(b"\x00\xff" if c == 0 else (b"\xff\x00" if c == 0xFF else bytes([c])) for c in b)


def save_checkpoint_v1(path: str, tensors: Dict[str, np.ndarray], block_bytes: int = 4096) -> None:
    """Write a single-file V1 checkpoint as tensor_slice_writer.cc does for unpartitioned variables: one full slice per
    tensor, values in the TensorProto's typed repeated field.  (The reference never writes V1 -- its Saver is V2; this
    exists so that the V1 reader can be exercised on real weights and a V1 file can be produced for a TF-slim tool.)"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    field = {1: 5, 2: 6, 3: 7, 9: 10}
    meta = b""
    items: List[Tuple[bytes, bytes]] = []
    for name in sorted(tensors, key=lambda n: n.encode("utf-8")):
        a = np.asarray(tensors[name])
        if a.dtype not in DT_INV:
            raise NotImplementedError("%s: dtype %s" % (name, a.dtype))
        dt = DT_INV[a.dtype]
        nm = name.encode("utf-8")
        shape = _encode_shape(a.shape)
        slc = b"".join(_pb_bytes(1, b"") for _ in a.shape)                 # one empty Extent per dimension
        meta += _pb_bytes(1, _pb_bytes(1, nm) + _pb_bytes(2, shape) + _pb_varint(3, dt) + _pb_bytes(4, slc))
        if dt in (1, 2):
            payload = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes()
        else:
            payload = b"".join(_put_varint(int(x)) for x in np.ascontiguousarray(a).reshape(-1))
        tp = _pb_varint(1, dt) + _pb_bytes(2, shape) + _pb_bytes(field[dt], payload)
        value = _pb_bytes(2, _pb_bytes(1, nm) + _pb_bytes(2, slc) + _pb_bytes(3, tp))
        nd = len(a.shape)
        key = b"\x00" + _ordered_string(nm) + b"\x00\x01" + (b"\x00" if nd == 0 else bytes([1, nd])) + b"\x80\x7f" * nd
        items.append((key, value))
    items.sort(key=lambda kv: kv[0])
    items.insert(0, (b"", _pb_bytes(1, meta + _pb_bytes(2, _pb_varint(1, 1)))))     # meta + versions { producer: 1 }
    with open(path, "wb") as f:
        index_entries: List[Tuple[bytes, bytes]] = []
        cur: List[Tuple[bytes, bytes]] = []
        cur_bytes = 0
        pending: Optional[Tuple[bytes, bytes]] = None

        def flush():
            nonlocal cur, cur_bytes, pending
            if cur:
                off, size = _emit_block(f, _build_block(cur, 16))
                pending = (cur[-1][0], _put_varint(off) + _put_varint(size))
                cur, cur_bytes = [], 0

        for k, v in items:
            if pending is not None:
                index_entries.append((_shortest_separator(pending[0], k), pending[1]))
                pending = None
            cur.append((k, v))
            cur_bytes += len(k) + len(v) + 3
            if cur_bytes >= block_bytes:
                flush()
        flush()
        if pending is not None:
            index_entries.append((_short_successor(pending[0]), pending[1]))
        mi_off, mi_size = _emit_block(f, _build_block([], 1))
        ix_off, ix_size = _emit_block(f, _build_block(index_entries, 1))
        footer = _put_varint(mi_off) + _put_varint(mi_size) + _put_varint(ix_off) + _put_varint(ix_size)
        f.write(footer + b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ---------------------------------------------------------------- public API
def data_path(prefix: str, shard: int = 0, num_shards: int = 1) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_variables(prefix: str, verify: bool = True) -> Dict[str, Dict[str, object]]:
    """name -> {dtype, shape, shard_id, offset, size, crc32c}; the header is returned under ''.
    A single-file V1 checkpoint (no <prefix>.index, <prefix> itself a table): list_variables_v1 + {'': {'format': 'v1'}}."""
    if not os.path.exists(prefix + ".index") and is_v1_checkpoint(prefix):
        d = list_variables_v1(prefix, verify)
        d[""] = {"format": "v1", "num_shards": 1, "endianness": 0}
        return d
    buf = open(prefix + ".index", "rb").read()
    if len(buf) < FOOTER_LEN or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s.index is not a TensorFlow table (bad magic)" % prefix)
    footer = buf[-FOOTER_LEN:]
    pos = 0
    _mi_off, pos = _get_varint(footer, pos)
    _mi_size, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    out: Dict[str, Dict[str, object]] = {}
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        for k, v in _block_entries(_read_block(buf, off, size, verify)):
            if k == b"":
                hdr = {"num_shards": 1, "endianness": 0}
                for fn, wt, val in _pb_fields(v):
                    if fn == 1:
                        hdr["num_shards"] = val
                    elif fn == 2:
                        hdr["endianness"] = val
                out[""] = hdr
            else:
                out[k.decode("utf-8")] = _decode_entry(v)
    if "" not in out:
        raise ValueError("bundle header missing in %s.index" % prefix)
    if out[""]["endianness"] != 0:
        raise NotImplementedError("big-endian bundle")
    return out


def load_checkpoint(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True
                    ) -> Dict[str, np.ndarray]:
    """Read tensors of a checkpoint by variable name (all, or the ``names`` given): a V2 bundle
    (<prefix>.index + .data-*) or, when <prefix> itself is a table file, a single-file V1 checkpoint."""
    if not os.path.exists(prefix + ".index") and is_v1_checkpoint(prefix):
        return load_checkpoint_v1(prefix, names, verify)
    entries = list_variables(prefix, verify)
    hdr = entries.pop("")
    want = set(names) if names is not None else None
    files: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    for name, e in entries.items():
        if want is not None and name not in want:
            continue
        if e["slices"]:
            raise NotImplementedError("%s is stored as tensor slices (partitioned variable)" % name)
        if e["dtype"] not in DT:
            raise NotImplementedError("%s: DataType %d" % (name, e["dtype"]))
        sh = int(e["shard_id"])
        if sh not in files:
            files[sh] = np.memmap(data_path(prefix, sh, int(hdr["num_shards"])), dtype=np.uint8, mode="r")
        raw = files[sh][int(e["offset"]):int(e["offset"]) + int(e["size"])]
        dt = np.dtype(DT[e["dtype"]])
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != int(e["size"]):
            raise ValueError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        if verify and mask_crc(crc32c(raw)) != e["crc32c"]:          # over the mapped bytes in place
            raise ValueError("%s: tensor crc mismatch" % name)
        out[name] = np.array(np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e["shape"]), dtype=dt)   # the one copy
    return out


def save_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_bytes: int = 4096) -> None:
    """Write a one-shard V2 bundle (+ nothing else; see write_checkpoint_state)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: List[Tuple[bytes, bytes]] = []
    offset = 0
    with open(data_path(prefix), "wb") as fd:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])            # (ascontiguousarray would turn a scalar into shape (1,))
            if not a.flags.c_contiguous:
                a = a.copy(order="C")
            if a.dtype not in DT_INV:
                raise NotImplementedError("%s: dtype %s" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            fd.write(raw)
            items.append((name.encode("utf-8"),
                          _encode_entry(DT_INV[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    # BundleHeaderProto: num_shards=1 (field 1), endianness LITTLE=0 (default, omitted), version {producer=1}
    header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))
    items.insert(0, (b"", header))
    with open(prefix + ".index", "wb") as f:
        index_entries: List[Tuple[bytes, bytes]] = []
        cur: List[Tuple[bytes, bytes]] = []
        cur_bytes = 0
        pending: Optional[Tuple[bytes, bytes]] = None      # (last key, handle) of a flushed block without its index key

        def flush():
            nonlocal cur, cur_bytes, pending
            if cur:
                off, size = _emit_block(f, _build_block(cur, 16))
                pending = (cur[-1][0], _put_varint(off) + _put_varint(size))
                cur, cur_bytes = [], 0

        for k, v in items:
            if pending is not None:       # TableBuilder::Add: the index key is chosen when the NEXT key is known
                index_entries.append((_shortest_separator(pending[0], k), pending[1]))
                pending = None
            cur.append((k, v))
            cur_bytes += len(k) + len(v) + 3
            if cur_bytes >= block_bytes:
                flush()
        flush()
        if pending is not None:           # TableBuilder::Finish
            index_entries.append((_short_successor(pending[0]), pending[1]))
        mi_off, mi_size = _emit_block(f, _build_block([], 1))             # empty metaindex block
        ix_off, ix_size = _emit_block(f, _build_block(index_entries, 1))
        footer = _put_varint(mi_off) + _put_varint(mi_size) + _put_varint(ix_off) + _put_varint(ix_size)
        f.write(footer + b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ---------------------------------------------------------------- the 'checkpoint' state file
def write_checkpoint_state(directory: str, model_checkpoint_path: str, all_paths: Optional[List[str]] = None) -> None:
    all_paths = all_paths or [model_checkpoint_path]
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % model_checkpoint_path)
        for p in all_paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)


def all_checkpoint_paths(directory: str) -> List[str]:
    """the all_model_checkpoint_paths entries of the directory's state file, oldest first ([] if absent)"""
    p = os.path.join(directory, "checkpoint")
    if not os.path.exists(p):
        return []
    return re.findall(r'^all_model_checkpoint_paths:\s*"([^"]*)"', open(p).read(), flags=re.M)


def get_checkpoint_state(directory: str) -> Optional[str]:
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (test/create_sdf.py:182-185):
    the prefix of the latest checkpoint, resolved relative to ``directory``; None if absent."""
    p = os.path.join(directory, "checkpoint")
    if not os.path.exists(p):
        return None
    m = re.search(r'^model_checkpoint_path:\s*"([^"]*)"', open(p).read(), flags=re.M)
    if not m:
        return None
    path = m.group(1)
    return path if os.path.isabs(path) else os.path.join(directory, path)
