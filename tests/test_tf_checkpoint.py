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
