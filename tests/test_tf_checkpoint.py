"""CPU tests of the TensorFlow-free Saver-V2 bundle reader/writer.  (-m "not gpu")"""
import os
import struct

import numpy as np
import pytest

from disn_amd import tf_checkpoint as tfc
from disn_amd.weights import WeightStore


def test_crc32c_known_answers():
    # RFC 3720 / iSCSI check value, and the masked form TF stores
    assert tfc._crc32c_py(b"123456789") == 0xE3069283
    assert tfc.crc32c(b"123456789") == 0xE3069283
    big = bytes(range(256)) * 64                       # 16 KiB -> goes through disn_crc32c
    assert tfc.crc32c(big) == tfc._crc32c_py(big)
    assert tfc.crc32c(big[:5000]) == tfc._crc32c_py(big[:5000])   # ragged tail (n % 8 != 0)
    assert tfc._crc32c_py(b"") == 0
    assert tfc.mask_crc(0) == 0xA282EAD8               # rotate of zero + the mask constant


def test_varint_and_proto_roundtrip():
    for n in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 40 + 7):
        b = tfc._put_varint(n)
        assert tfc._get_varint(b, 0) == (n, len(b))
    e = tfc._encode_entry(1, (3, 3, 64, 128), 0, 123456789012, 294912, 0xDEADBEEF)
    d = tfc._decode_entry(e)
    assert d["dtype"] == 1 and d["shape"] == (3, 3, 64, 128) and d["offset"] == 123456789012
    assert d["size"] == 294912 and d["crc32c"] == 0xDEADBEEF and d["shard_id"] == 0
    assert tfc._decode_entry(tfc._encode_entry(1, (), 0, 0, 4, 1))["shape"] == ()   # scalar (beta1_power)


def test_bundle_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"vgg_16/conv1/conv1_1/weights": rng.standard_normal((3, 3, 3, 64)).astype(np.float32),
               "vgg_16/conv1/conv1_1/biases": rng.standard_normal(64).astype(np.float32),
               "sdfprediction/fold2/conv5/weights": rng.standard_normal((1, 1, 256, 1)).astype(np.float32),
               "beta1_power": np.float32(0.5) * np.ones((), np.float32),
               "global_step": np.array(1234, np.int64)}
    for i in range(300):                                # enough names for several index blocks
        tensors["pad/var_%03d/Adam" % i] = rng.standard_normal(5).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt")
    tfc.save_checkpoint(prefix, tensors)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57          # table magic
    ents = tfc.list_variables(prefix)
    assert ents[""]["num_shards"] == 1 and ents[""]["endianness"] == 0
    names = [k for k in ents if k]
    assert names == sorted(names, key=lambda s: s.encode())                # table keys are sorted
    assert set(names) == set(tensors)
    # data file = tensors back to back in key order
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in tensors.values())
    back = tfc.load_checkpoint(prefix)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    some = tfc.load_checkpoint(prefix, ["global_step", "vgg_16/conv1/conv1_1/biases"])
    assert set(some) == {"global_step", "vgg_16/conv1/conv1_1/biases"} and some["global_step"] == 1234


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    tfc.save_checkpoint(prefix, {"a/weights": np.arange(4096, dtype=np.float32)})
    d = prefix + ".data-00000-of-00001"
    b = bytearray(open(d, "rb").read()); b[100] ^= 0xFF
    open(d, "wb").write(bytes(b))
    with pytest.raises(ValueError, match="crc"):
        tfc.load_checkpoint(prefix)
    assert tfc.load_checkpoint(prefix, verify=False)["a/weights"].shape == (4096,)
    idx = prefix + ".index"
    b = bytearray(open(idx, "rb").read()); b[3] ^= 0x55
    open(idx, "wb").write(bytes(b))
    with pytest.raises(ValueError):
        tfc.list_variables(prefix)
    open(idx, "wb").write(b"not a table")
    with pytest.raises(ValueError, match="magic"):
        tfc.list_variables(prefix)


def test_weight_store_tf_restore_semantics(tmp_path):
    """decoder + a few VGG variables + Adam slots, restored the way the reference does
    (name + exact shape; train/train_sdf.py:196-205, 276-278, 285-286)"""
    full = WeightStore.random_init(0, mode="he")
    sub = {k: v for k, v in full.items() if k.startswith("sdfprediction") or "conv1_1" in k}
    sub["sdfprediction/fold1/conv1/weights/Adam"] = np.zeros((1, 1, 3, 64), np.float32)   # optimizer slot
    sub["beta1_power"] = np.array(0.5, np.float32)
    sub["vgg_16/fc8/weights"] = np.zeros((1, 1, 4096, 1000), np.float32)                  # ImageNet head: shape mismatch
    d = tmp_path / "ckpt"; d.mkdir()
    prefix = str(d / "model.ckpt")
    tfc.save_checkpoint(prefix, sub)
    tfc.write_checkpoint_state(str(d), "model.ckpt")
    assert tfc.get_checkpoint_state(str(d)) == prefix
    assert tfc.get_checkpoint_state(str(tmp_path)) is None
    ws = WeightStore.restore_latest(str(d))
    assert ws is not None and not ws.complete()
    for k in sub:
        if k in ws.shapes and sub[k].shape == ws.shapes[k]:
            assert np.array_equal(ws[k], sub[k]), k
    assert "vgg_16/fc8/weights" not in ws and "beta1_power" not in ws
    only_vgg = WeightStore.load_tf(prefix, name_prefix="vgg_16")
    assert set(only_vgg.keys()) == {"vgg_16/conv1/conv1_1/weights", "vgg_16/conv1/conv1_1/biases"}
    with pytest.raises((KeyError, ValueError)):
        WeightStore.load_tf(prefix, strict=True)
    assert WeightStore.restore_latest(str(tmp_path)) is None


@pytest.mark.timeout(600)
def test_full_model_bundle_roundtrip(tmp_path):
    """all 140.8 M parameters (563 MB) through save_tf / load_tf, byte-identical"""
    ws = WeightStore.random_init(1)
    prefix = str(tmp_path / "full" / "model.ckpt")
    ws.save_tf(prefix)
    back = WeightStore.load_tf(prefix, strict=True)
    assert back.complete()
    for k in ws.keys():
        assert np.array_equal(back[k], ws[k]), k
    assert tfc.get_checkpoint_state(os.path.dirname(prefix)) == prefix


def test_step_recovery_from_a_bundle(tmp_path):
    """the reference's Saver leaves `batch` out of its bundles (train/train_sdf.py:285-286), so a restore
    restarts the learning-rate schedule at 0 and Adam's timestep comes back from beta2_power; `batch` in a bundle is
    this implementation's opt-in extension; an underflowed power gives the caller's explicit step"""
    import pytest
    pytest.importorskip("torch")
    from disn_amd.train_sdf import adam_step_from_checkpoint, schedule_step_from_checkpoint
    b2 = 0.999
    for t in (0, 1, 999, 50000):
        assert adam_step_from_checkpoint({"beta2_power": np.float32(b2 ** (t + 1))}, b2) == t
    assert adam_step_from_checkpoint({"beta2_power": np.float32(0.0)}, b2, current=0) >= 200000
    assert adam_step_from_checkpoint({"beta2_power": np.float32(0.0)}, b2, current=0, underflow_step=123456) == 123456
    assert adam_step_from_checkpoint({}, b2, current=7) == 7
    assert np.float32(b2 ** 104000) == 0.0 and np.float32(b2 ** 99000) > 0   # where float32 gives up
    assert schedule_step_from_checkpoint({"beta2_power": np.float32(0.5)}) == 0          # a reference bundle
    assert schedule_step_from_checkpoint({"batch": np.asarray(123456, np.int32)}) == 123456
    # the state file keeps a history (tf.train.Saver's all_model_checkpoint_paths)
    d = str(tmp_path)
    tfc.write_checkpoint_state(d, "m-2", ["m-1", "m-2"])
    assert tfc.all_checkpoint_paths(d) == ["m-1", "m-2"] and tfc.get_checkpoint_state(d) == os.path.join(d, "m-2")


# ---------------------------------------------------------------------------------------------------------------
# A bundle assembled BY HAND from the TensorFlow format description (tensorflow/core/lib/io/{format,block_builder,
# table_builder}.cc, tensor_bundle.proto, RFC 3720 CRC-32C) -- nothing below uses disn_amd.tf_checkpoint to build
# the bytes; the reader has to read it, and the writer has to produce exactly these bytes.
# ---------------------------------------------------------------------------------------------------------------
def _kat_crc32c(data: bytes) -> int:
    """bit-by-bit CRC-32C (reflected polynomial 0x82F63B78), independent of the module's table version"""
    c = 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def _kat_mask(c: int) -> int:           # crc32c.h: ((crc >> 15) | (crc << 17)) + 0xa282ead8
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _kat_block(body: bytes) -> bytes:   # block + trailer: compression type 0, masked crc over body + type
    return body + b"\x00" + struct.pack("<I", _kat_mask(_kat_crc32c(body + b"\x00")))


def _hand_assembled_bundle():
    assert _kat_crc32c(b"123456789") == 0xE3069283 and _kat_crc32c(b"\x00" * 32) == 0x8A9136AA   # RFC 3720 B.4
    t_ab = struct.pack("<2f", 1.0, -2.0)                    # 'a/b' float32 [2]
    t_ac = struct.pack("<i", 7)                             # 'a/c' int32 scalar
    data = t_ab + t_ac
    header = bytes.fromhex("0801" "1a020801")                # num_shards = 1; version { producer: 1 }
    e_ab = (bytes.fromhex("0801" "1204" "1202" "0802" "2808" "35")   # dtype DT_FLOAT; shape {dim {size: 2}}; size 8; crc32c:
            + struct.pack("<I", _kat_mask(_kat_crc32c(t_ab))))
    e_ac = (bytes.fromhex("0803" "1200" "2008" "2804" "35")          # DT_INT32; shape {}; offset 8; size 4
            + struct.pack("<I", _kat_mask(_kat_crc32c(t_ac))))
    body = b""
    body += bytes([0, 0, len(header)]) + header                      # key "":   shared 0, non-shared 0
    body += bytes([0, 3, len(e_ab)]) + b"a/b" + e_ab                 # key a/b:  shared 0, non-shared 3
    body += bytes([2, 1, len(e_ac)]) + b"c" + e_ac                   # key a/c:  shared 2 ("a/"), non-shared 1
    body += struct.pack("<II", 0, 1)                                 # restart offsets [0], count 1
    data_block = _kat_block(body)
    meta_block = _kat_block(struct.pack("<II", 0, 1))                # empty metaindex block
    handle0 = bytes([0, len(body)])                                  # BlockHandle varints: offset 0, size
    # index entry: key = FindShortSuccessor("a/c") = "b" (first byte incremented, rest dropped)
    index_body = bytes([0, 1, len(handle0)]) + b"b" + handle0 + struct.pack("<II", 0, 1)
    index_block = _kat_block(index_body)
    meta_off = len(data_block)
    index_off = meta_off + len(meta_block)
    footer = bytes([meta_off, 8]) + bytes([index_off, len(index_body)])
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    assert max(len(body), len(index_body), index_off) < 128           # every varint above is one byte
    return data, data_block + meta_block + index_block + footer


def test_reader_reads_a_hand_assembled_bundle(tmp_path):
    data, index = _hand_assembled_bundle()
    prefix = str(tmp_path / "kat.ckpt")
    open(prefix + ".index", "wb").write(index)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = tfc.load_checkpoint(prefix)
    assert sorted(got) == ["a/b", "a/c"]
    assert got["a/b"].dtype == np.float32 and got["a/b"].tolist() == [1.0, -2.0]
    assert got["a/c"].dtype == np.int32 and got["a/c"].shape == () and int(got["a/c"]) == 7
    lv = tfc.list_variables(prefix)
    assert lv[""]["num_shards"] == 1 and lv["a/c"]["offset"] == 8 and lv["a/b"]["shape"] == (2,)
    bad = bytearray(index)
    bad[10] ^= 1                                                      # inside the data block
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tfc.load_checkpoint(prefix)


def test_writer_produces_the_hand_assembled_bytes(tmp_path):
    data, index = _hand_assembled_bundle()
    prefix = str(tmp_path / "w.ckpt")
    tfc.save_checkpoint(prefix, {"a/c": np.asarray(7, np.int32), "a/b": np.asarray([1.0, -2.0], np.float32)})
    assert open(prefix + ".data-00000-of-00001", "rb").read() == data
    assert open(prefix + ".index", "rb").read() == index


# ---------------------------------------------------------------------------------------------------------------
# A single-file V1 checkpoint assembled BY HAND (tensorflow/core/util/tensor_slice_writer.cc, saved_tensor_slice.proto,
# tensor_slice.proto, tensor.proto, lib/strings/ordered_code.cc): what TF-slim's vgg_16.ckpt is (README.md:128).
#   key ""                     -> SavedTensorSlices { meta { tensor { name, shape, type, slice } ... } }
#   key 00 <name> 00 01 <ndims> (<start> <length>)*  (ordered code; small signed numbers are one byte 0x80 + x, the
#                                 full extent's length -1 is 0x7f)
#                              -> SavedTensorSlices { data { name, slice, data: TensorProto with float_val / int_val } }
# ---------------------------------------------------------------------------------------------------------------
def _pb(fn, payload):              # a length-delimited field (payload < 128 bytes)
    assert len(payload) < 128
    return bytes([(fn << 3) | 2, len(payload)]) + payload


def _hand_assembled_v1():
    shape2 = _pb(2, bytes([0x08, 2]))                       # TensorShapeProto { dim { size: 2 } }
    shape4 = _pb(2, bytes([0x08, 4]))
    full1 = _pb(1, b"")                                     # TensorSliceProto { extent {} }: the whole dimension
    meta = b"".join(_pb(1, m) for m in (                    # SavedTensorSliceMeta.tensor = 1
        _pb(1, b"a/b") + _pb(2, shape2) + bytes([0x18, 1]) + _pb(4, full1),          # float32 [2], one full slice
        _pb(1, b"a/c") + _pb(2, b"") + bytes([0x18, 3]) + _pb(4, b""),               # int32 scalar
        _pb(1, b"p") + _pb(2, shape4) + bytes([0x18, 1])                             # float32 [4] in two slices
        + _pb(4, _pb(1, bytes([0x10, 2]))) + _pb(4, _pb(1, bytes([0x08, 2, 0x10, 2])))))
    v_meta = _pb(1, meta)                                   # SavedTensorSlices.meta = 1
    tp_ab = bytes([0x08, 1]) + _pb(2, shape2) + _pb(5, struct.pack("<2f", 1.0, -2.0))          # dtype, shape, packed float_val
    tp_ac = bytes([0x08, 3]) + _pb(2, b"") + _pb(7, bytes([7]))                                # int_val packed: varint 7
    tp_p0 = bytes([0x08, 1]) + _pb(2, shape2) + _pb(5, struct.pack("<2f", 10.0, 11.0))
    tp_p1 = bytes([0x08, 1]) + _pb(2, shape2) + _pb(4, struct.pack("<2f", 12.0, 13.0))         # tensor_content instead
    v_ab = _pb(2, _pb(1, b"a/b") + _pb(2, full1) + _pb(3, tp_ab))                              # SavedTensorSlices.data = 2
    v_ac = _pb(2, _pb(1, b"a/c") + _pb(2, b"") + _pb(3, tp_ac))
    v_p0 = _pb(2, _pb(1, b"p") + _pb(2, _pb(1, bytes([0x10, 2]))) + _pb(3, tp_p0))             # extent { length: 2 } (start 0)
    v_p1 = _pb(2, _pb(1, b"p") + _pb(2, _pb(1, bytes([0x08, 2, 0x10, 2]))) + _pb(3, tp_p1))    # extent { start: 2 length: 2 }
    k_ab = b"\x00" + b"a/b" + b"\x00\x01" + b"\x01\x01" + b"\x80\x7f"       # 0, "a/b", 1 dim, start 0, length -1
    k_ac = b"\x00" + b"a/c" + b"\x00\x01" + b"\x00"                         # 0 dims
    k_p0 = b"\x00" + b"p" + b"\x00\x01" + b"\x01\x01" + b"\x80\x82"         # start 0, length 2
    k_p1 = b"\x00" + b"p" + b"\x00\x01" + b"\x01\x01" + b"\x82\x82"         # start 2, length 2
    entries = [(b"", v_meta), (k_ab, v_ab), (k_ac, v_ac), (k_p0, v_p0), (k_p1, v_p1)]
    assert [k for k, _ in entries] == sorted(k for k, _ in entries)
    body = b""
    for k, v in entries:                                    # no key sharing (shared = 0 is always legal)
        assert len(k) < 128 and len(v) < 128
        body += bytes([0, len(k), len(v)]) + k + v
    body += struct.pack("<II", 0, 1)
    assert len(body) < 16384
    data_block = _kat_block(body)
    meta_block = _kat_block(struct.pack("<II", 0, 1))
    handle0 = bytes([0]) + bytes([len(body) & 0x7F | 0x80, len(body) >> 7])               # varint of a size >= 128
    index_body = bytes([0, 1, len(handle0)]) + b"\x01" + handle0 + struct.pack("<II", 0, 1)   # any key >= the last data key
    index_block = _kat_block(index_body)
    meta_off = len(data_block)
    index_off = meta_off + len(meta_block)

    def varint(n):
        out = b""
        while n >= 128:
            out += bytes([n & 0x7F | 0x80])
            n >>= 7
        return out + bytes([n])
    footer = varint(meta_off) + varint(8) + varint(index_off) + varint(len(index_body))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    return data_block + meta_block + index_block + footer


def test_reader_reads_a_hand_assembled_v1_checkpoint(tmp_path):
    """the single-file format of TF-slim's vgg_16.ckpt: typed value fields and tensor_content, a scalar, a partitioned
    variable in two slices; auto-detected by load_checkpoint / list_variables and by the weight store"""
    path = str(tmp_path / "vgg_like.ckpt")
    open(path, "wb").write(_hand_assembled_v1())
    assert tfc.is_v1_checkpoint(path) and not os.path.exists(path + ".index")
    lv = tfc.list_variables(path)
    assert lv[""]["format"] == "v1" and lv["a/b"]["shape"] == (2,) and lv["p"]["slices"] == 2 and lv["a/c"]["dtype"] == 3
    got = tfc.load_checkpoint(path)
    assert sorted(got) == ["a/b", "a/c", "p"]
    assert got["a/b"].dtype == np.float32 and got["a/b"].tolist() == [1.0, -2.0]
    assert got["a/c"].dtype == np.int32 and got["a/c"].shape == () and int(got["a/c"]) == 7
    assert got["p"].tolist() == [10.0, 11.0, 12.0, 13.0]
    assert sorted(tfc.load_checkpoint(path, names=["p"])) == ["p"]
    bad = bytearray(open(path, "rb").read())
    bad[20] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tfc.load_checkpoint(path)


def test_weight_store_restores_from_a_v1_file(tmp_path):
    """WeightStore.load_tf on a V1 file written with the module's own table code (same data-block builder as the V2
    index): the conv1_1 pair of a slim-style vgg_16.ckpt, prefix restore as train/train_sdf.py:196-205"""
    from disn_amd.weights import WeightStore
    ws = WeightStore.random_init(5)
    names = ["vgg_16/conv1/conv1_1/weights", "vgg_16/conv1/conv1_1/biases"]
    path = str(tmp_path / "vgg_16.ckpt")
    tfc.save_checkpoint_v1(path, {n: ws[n] for n in names})
    back = WeightStore.load_tf(path, name_prefix="vgg_16")
    for n in names:
        assert np.array_equal(back[n], ws[n])
