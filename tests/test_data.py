"""Loader mirror (disn_amd/data_sdf.py vs data/data_sdf_h5_queue.py): batch schema, sampling rules,
epoch order, the producer thread.  CPU only; a synthetic dataset is written as .npz files."""
import types

import os

import numpy as np
import pytest

from disn_amd import data_sdf as D


def _flags(**kw):
    base = dict(num_points=64, num_sample_points=32, batch_size=2, img_h=137, img_w=137, rot=False,
                max_epoch=2, cat_limit=100, backcolorwhite=False, alpha=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


@pytest.fixture()
def dataset(tmp_path):
    rng = np.random.default_rng(0)
    info = {"rendered_dir": str(tmp_path / "img"), "sdf_dir": str(tmp_path / "sdf")}
    listinfo = []
    for cat, n_obj, n_pts in (("03001627", 3, 100), ("02691156", 2, 20)):
        for o in range(n_obj):
            obj = "obj%d" % o
            pts = rng.uniform(-1, 1, (n_pts, 3)).astype(np.float32)
            sdf = np.concatenate([pts, pts[:, :1] * 0.1 + o], axis=1)         # value identifies (point, obj)
            D.save_sample(info["sdf_dir"], cat, obj, rng.uniform(-1, 1, (50, 4)), sdf, [0, 0, 0, 1],
                          [-1, -1, -1, 1, 1, 1])
            for view in range(2):
                img = rng.integers(0, 256, (137, 137, 4), dtype=np.uint8)
                img[:10, :, 3] = 0
                rot = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32)
                D.save_view(info["rendered_dir"], cat, obj, view, img, rng.standard_normal((4, 3)), rot,
                            rng.standard_normal((4, 3)))
                listinfo.append([cat, obj, view])
    return info, listinfo


def test_batch_schema_and_sampling(dataset):
    info, listinfo = dataset
    ds = D.Pt_sdf_img(_flags(rot=True), listinfo=listinfo, info=info, shuffle=False, seed=1)
    assert len(ds) == len(listinfo) == 10 and ds.num_batches == 5
    b = ds.get_batch(0)
    assert set(b) == {"pc", "sdf_pt", "sdf_pt_rot", "sdf_val", "norm_params", "sdf_params", "img", "trans_mat",
                      "cat_id", "obj_nm", "view_id"}
    assert b["pc"].shape == (2, 64, 3) and b["sdf_pt"].shape == (2, 32, 3) and b["sdf_val"].shape == (2, 32, 1)
    assert b["img"].shape == (2, 137, 137, 3) and b["img"].dtype == np.float32 and 0 <= b["img"].min() and b["img"].max() <= 1
    assert b["trans_mat"].shape == (2, 4, 3) and b["sdf_params"].shape == (2, 6) and b["norm_params"].shape == (2, 4)
    assert b["cat_id"] == ["03001627", "03001627"] and b["view_id"] == [0, 1]
    # 100 candidates >= 32 requested: sampling WITHOUT replacement; value column stays attached to its point
    for i in range(2):
        assert len({tuple(p) for p in b["sdf_pt"][i]}) == 32
        assert np.allclose(b["sdf_val"][i, :, 0], b["sdf_pt"][i, :, 0] * 0.1 + 0)
        assert np.allclose(b["sdf_pt_rot"][i], b["sdf_pt"][i] @ np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32))
    # 20 candidates < 32 requested: WITH replacement
    b2 = D.Pt_sdf_img(_flags(), listinfo=[l for l in listinfo if l[0] == "02691156"], info=info, shuffle=False,
                      seed=2).get_batch(0)
    assert len({tuple(p) for p in b2["sdf_pt"][0]}) <= 20 and np.array_equal(b2["sdf_pt_rot"], b2["sdf_pt"])


def test_backcolorwhite_and_truncated_h5(dataset, tmp_path):
    info, listinfo = dataset
    ds = D.Pt_sdf_img(_flags(backcolorwhite=True), listinfo=listinfo, info=info, shuffle=False, seed=1)
    assert np.all(ds.get_batch(0)["img"][:, :10] == 1.0)
    (tmp_path / "x.h5").write_bytes(b"\x89HDF\r\n\x1a\n")          # a signature and nothing else
    with pytest.raises(ValueError, match="truncated"):
        D._load(str(tmp_path / "x.h5"), ("a",))


def test_epoch_order_respects_category_quota(dataset):
    info, listinfo = dataset
    ds = D.Pt_sdf_img(_flags(cat_limit=3), listinfo=listinfo, info=info, shuffle=True, seed=3)
    assert ds.cats_limit == {"03001627": 3, "02691156": 3} and len(ds) == 6
    for _ in range(5):
        order = ds.refill_data_order()
        assert len(order) == 6 and len(set(order)) == 6
        cats = [listinfo[i][0] for i in order]
        assert cats.count("03001627") == 3 and cats.count("02691156") == 3


def test_producer_thread_fetch_and_shutdown(dataset):
    info, listinfo = dataset
    ds = D.Pt_sdf_img(_flags(max_epoch=1), listinfo=listinfo, info=info, qsize=2, shuffle=True, seed=4)
    ds.start()
    seen = []
    for _ in range(ds.num_batches):
        b = ds.fetch(timeout=60)
        seen += list(zip(b["cat_id"], b["obj_nm"], b["view_id"]))
    assert len(seen) == 10 and len(set(seen)) == 10           # one epoch = every sample once
    ds.shutdown()
    assert ds.fetch() is None
    ds.join(timeout=10)
    assert not ds.is_alive()


def test_feed_from_batch_shards_and_offsets(dataset):
    torch = pytest.importorskip("torch")
    from disn_amd.train_sdf import feed_from_batch
    info, listinfo = dataset
    b = D.Pt_sdf_img(_flags(batch_size=4), listinfo=listinfo, info=info, shuffle=False, seed=5).get_batch(0)
    f = feed_from_batch(b, torch.device("cpu"), rank=1, world=2)
    assert f["imgs"].shape == (2, 137, 137, 3) and f["sample_pc"].shape == (2, 32, 3)
    assert torch.equal(f["sdf"], torch.from_numpy(b["sdf_val"][2:4] - np.float32(0.003)))   # train_sdf.py:375
    assert torch.equal(f["trans_mat"], torch.from_numpy(b["trans_mat"][2:4]))


# ---------------------------------------------------------------------------------------------------------------
# An HDF5 file assembled BY HAND from the HDF5 File Format Specification (version 1.x structures: superblock v0, v1
# B-trees, symbol table nodes, local heap, version-1 object headers, layout message v3, filter pipeline v1) -- nothing
# below uses disn_amd.hdf5_lite to build the bytes.  Contents (what the reference's loaders meet, data/data_sdf_h5_queue.py
# :121-186, written by h5py's create_dataset(..., compression='gzip')):
#   /c      float32 [2,3]   contiguous
#   /z      float32 [5,4]   chunked [2,4], deflate; the last chunk hangs over the dataset's edge
#   /s      int16   [3,5]   chunked [2,2], shuffle + deflate, edge chunks in both dimensions; its object header continues
#                           in a second block (continuation message)
#   /g/d    int32   [2]     compact, inside a sub-group whose symbol table entry has no cached B-tree / heap addresses
# ---------------------------------------------------------------------------------------------------------------
import struct
import zlib


class _Img:
    """a growing byte image with 8-byte aligned allocation"""

    def __init__(self):
        self.b = bytearray()

    def alloc(self, data: bytes) -> int:
        while len(self.b) % 8:
            self.b += b"\x00"
        a = len(self.b)
        self.b += data
        return a

    def reserve(self, n: int) -> int:
        return self.alloc(b"\x00" * n)

    def put(self, addr: int, data: bytes):
        self.b[addr:addr + len(data)] = data


def _msg(mtype, data):
    data = data + b"\x00" * (-len(data) % 8)
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _ohdr(msgs):                      # version-1 object header: prefix (12 bytes, padded to 16) + messages
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


def _dataspace(shape):                # version 1: version, rank, flags, reserved (1 + 4), dimension sizes
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _dtype_f32():                     # class 1 (float) version 1; little endian, IEEE: sign bit 31; size 4; properties
    return struct.pack("<B3BI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dtype_int(size, signed):         # class 0 (fixed point) version 1; bit 3 of the first flag byte: signed
    return struct.pack("<B3BI", 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack("<HH", 0, 8 * size)


def _filters(ids_cd):                 # filter pipeline version 1: version, count, 6 reserved; per filter id, name length
    out = struct.pack("<BB6x", 1, len(ids_cd))          # (0), flags, #client values, client values (+ pad if odd)
    for fid, cd in ids_cd:
        out += struct.pack("<HHHH", fid, 0, 0, len(cd)) + b"".join(struct.pack("<I", v) for v in cd)
        if len(cd) % 2:
            out += b"\x00" * 4
    return out


def _chunk_btree(img, ndim1, entries):   # a leaf of the chunk index: "TREE", type 1, level 0, N entries, no siblings
    node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(entries), 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
    for size, offs, addr in entries:
        node += struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", addr)
    node += struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", 0) for _ in range(ndim1))   # the final key
    return img.alloc(node)


def _group(img, names_to_entries):
    """local heap + one symbol table node + one B-tree leaf -> (btree address, heap address)"""
    heap_data = bytearray(b"\x00" * 8)              # offset 0: the empty name
    offs = {}
    for name in sorted(names_to_entries):
        offs[name] = len(heap_data)
        heap_data += name.encode() + b"\x00"
        heap_data += b"\x00" * (-len(heap_data) % 8)
    data_addr = img.alloc(bytes(heap_data))
    heap = img.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 0xFFFFFFFFFFFFFFFF, data_addr))
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names_to_entries))
    for name in sorted(names_to_entries):
        hdr, cache, scratch = names_to_entries[name]
        snod += struct.pack("<QQI4x", offs[name], hdr, cache) + scratch
    snod_addr = img.alloc(snod)
    last = offs[sorted(names_to_entries)[-1]]
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
    tree += struct.pack("<QQQ", 0, snod_addr, last)       # key 0 (empty name), child 0, key 1 (largest name)
    return img.alloc(tree), heap


def _hand_assembled_hdf5():
    img = _Img()
    img.reserve(96)                                           # superblock, written last
    c = np.arange(6, dtype="<f4").reshape(2, 3) * 1.5
    z = (np.arange(20, dtype="<f4").reshape(5, 4) - 7.25)
    s = (np.arange(15, dtype="<i2").reshape(3, 5) * 37 - 200)
    d = np.array([-5, 123456], "<i4")
    # /c contiguous
    c_addr = img.alloc(c.tobytes())
    c_hdr = img.alloc(_ohdr([_msg(1, _dataspace(c.shape)), _msg(3, _dtype_f32()),
                             _msg(8, struct.pack("<BBQQ", 3, 1, c_addr, c.nbytes))]))
    # /z chunked [2,4] + deflate
    ents = []
    for r0 in (0, 2, 4):
        blk = np.zeros((2, 4), "<f4")
        blk[:min(2, 5 - r0)] = z[r0:r0 + 2]
        raw = zlib.compress(blk.tobytes(), 4)
        ents.append((len(raw), (r0, 0, 0), img.alloc(raw)))
    zt = _chunk_btree(img, 3, ents)
    z_hdr = img.alloc(_ohdr([_msg(1, _dataspace(z.shape)), _msg(3, _dtype_f32()), _msg(0xB, _filters([(1, [4])])),
                             _msg(8, struct.pack("<BBBQ3I", 3, 2, 3, zt, 2, 4, 4))]))
    # /s chunked [2,2], shuffle (element size 2) + deflate; edge chunks in both dimensions; header with a continuation
    ents = []
    for r0 in (0, 2):
        for c0 in (0, 2, 4):
            blk = np.zeros((2, 2), "<i2")
            sub = s[r0:r0 + 2, c0:c0 + 2]
            blk[:sub.shape[0], :sub.shape[1]] = sub
            by = np.frombuffer(blk.tobytes(), np.uint8).reshape(4, 2).T.tobytes()     # shuffle: byte 0 of all, byte 1 of all
            raw = zlib.compress(by, 4)
            ents.append((len(raw), (r0, c0, 0), img.alloc(raw)))
    st = _chunk_btree(img, 3, ents)
    cont = _msg(0xB, _filters([(2, [2]), (1, [4])])) + _msg(8, struct.pack("<BBBQ3I", 3, 2, 3, st, 2, 2, 2))
    cont_addr = img.alloc(cont)
    first = [_msg(1, _dataspace(s.shape)), _msg(3, _dtype_int(2, True)), _msg(0x10, struct.pack("<QQ", cont_addr, len(cont)))]
    body = b"".join(first)
    s_hdr = img.alloc(struct.pack("<BxHII4x", 1, 5, 1, len(body)) + body)            # 5 messages in two blocks
    # /g/d compact
    d_hdr = img.alloc(_ohdr([_msg(1, _dataspace(d.shape)), _msg(3, _dtype_int(4, True)),
                             _msg(8, struct.pack("<BBH", 3, 0, d.nbytes) + d.tobytes())]))
    gt, gh = _group(img, {"d": (d_hdr, 0, b"\x00" * 16)})
    g_hdr = img.alloc(_ohdr([_msg(0x11, struct.pack("<QQ", gt, gh))]))
    # root group
    rt, rh = _group(img, {"c": (c_hdr, 0, b"\x00" * 16), "z": (z_hdr, 0, b"\x00" * 16), "s": (s_hdr, 0, b"\x00" * 16),
                          "g": (g_hdr, 0, b"\x00" * 16)})     # the sub-group WITHOUT cached addresses
    root_hdr = img.alloc(_ohdr([_msg(0x11, struct.pack("<QQ", rt, rh))]))
    eof = len(img.b)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, 0xFFFFFFFFFFFFFFFF, eof, 0xFFFFFFFFFFFFFFFF)
    sb += struct.pack("<QQI4x", 0, root_hdr, 1) + struct.pack("<QQ", rt, rh)      # root entry: cached B-tree / heap
    assert len(sb) == 96
    img.put(0, sb)
    return bytes(img.b), {"c": c, "z": z, "s": s, "g/d": d}


def test_hdf5_reader_reads_a_hand_assembled_file(tmp_path):
    from disn_amd.hdf5_lite import Hdf5File
    data, want = _hand_assembled_hdf5()
    path = str(tmp_path / "kat.h5")
    open(path, "wb").write(data)
    f = Hdf5File(path)
    assert f.keys() == sorted(want)
    for k, v in want.items():
        got = f[k]
        assert got.dtype == v.dtype.newbyteorder("=") or got.dtype == v.dtype
        assert got.shape == v.shape and np.array_equal(got, v), k
    with pytest.raises(KeyError):
        f["nope"]
    bad = bytearray(data)
    bad[8] = 2                                  # a superblock written with libver='latest'
    open(path, "wb").write(bytes(bad))
    with pytest.raises(NotImplementedError):
        Hdf5File(path)


def test_loader_reads_reference_style_h5_files(tmp_path, monkeypatch):
    """data_sdf._load on an .h5 file (no .npz beside it, no h5py): the plain-Python HDF5 subset reader"""
    data, want = _hand_assembled_hdf5()
    path = str(tmp_path / "ori_sample.h5")
    open(path, "wb").write(data)
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py is not installed")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_h5py)
    got = D._load(path, ["c", "z", "missing"])
    assert sorted(got) == ["c", "z"] and np.array_equal(got["z"], want["z"])


# ---------------------------------------------------------------------------------------------------------------
# round 5: files WRITTEN BY libhdf5 (tests/golden/h5/, generated by tests/golden/make_h5_fixtures.py with the h5py 3.3.0 /
# HDF5 1.10.6 of this image's conda tree: the reference's own create_dataset calls, preprocessing/create_point_sdf_grid.py
# :154-158 and preprocessing/create_img_h5.py:188-200, plus the other layouts h5py produces) -- the reader against files
# it did not shape itself
# ---------------------------------------------------------------------------------------------------------------
H5_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
H5_SHA256 = {   # the committed fixtures (sha256 of the bytes libhdf5 wrote)
    "ori_sample.h5": None, "img_00.h5": None, "layouts.h5": None}


def test_hdf5_reader_reads_libhdf5_written_files():
    import hashlib
    from disn_amd.hdf5_lite import Hdf5File
    exp = np.load(os.path.join(H5_DIR, "expected.npz"))
    seen = set()
    for key in exp.files:
        fn, name = key.split(":")
        f = Hdf5File(os.path.join(H5_DIR, fn))
        got = np.asarray(f[name])
        want = exp[key]
        assert got.dtype == want.dtype and got.shape == want.shape, (key, got.dtype, got.shape)
        assert np.array_equal(got, want), key
        seen.add(fn)
    assert seen == set(H5_SHA256)
    sums = {fn: hashlib.sha256(open(os.path.join(H5_DIR, fn), "rb").read()).hexdigest() for fn in H5_SHA256}
    recorded = dict(reversed(l.split()) for l in open(os.path.join(H5_DIR, "SHA256SUMS")).read().splitlines() if l.strip())
    assert sums == {fn: recorded[fn] for fn in H5_SHA256}, "fixture bytes changed: re-run make_h5_fixtures.py and update SHA256SUMS"
    # top-level listing as h5py's keys()
    assert sorted(Hdf5File(os.path.join(H5_DIR, "ori_sample.h5")).keys()) == ["norm_params", "pc_sdf_original", "pc_sdf_sample", "sdf_params"]


def test_loader_reads_the_libhdf5_written_reference_files(monkeypatch):
    """data_sdf._load (data/data_sdf_h5_queue.py:121-186's h5py reads) on the libhdf5-written files, h5py absent"""
    import builtins
    import disn_amd.data_sdf as D
    real = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py is not installed")
        return real(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_h5py)
    exp = np.load(os.path.join(H5_DIR, "expected.npz"))
    got = D._load(os.path.join(H5_DIR, "ori_sample.h5"), ("pc_sdf_original", "pc_sdf_sample", "norm_params", "sdf_params"))
    for k in ("pc_sdf_original", "pc_sdf_sample", "norm_params", "sdf_params"):
        assert np.array_equal(np.asarray(got[k]), exp["ori_sample.h5:" + k]), k
    img = D._load(os.path.join(H5_DIR, "img_00.h5"), ("img_arr", "trans_mat", "K", "RT"))
    assert np.array_equal(np.asarray(img["img_arr"]), exp["img_00.h5:img_arr"]) and img["img_arr"].dtype == np.uint8
    assert np.array_equal(np.asarray(img["trans_mat"]), exp["img_00.h5:trans_mat"])
