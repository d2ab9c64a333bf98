"""Loader mirror (disn_amd/data_sdf.py vs data/data_sdf_h5_queue.py): batch schema, sampling rules,
epoch order, the producer thread.  CPU only; a synthetic dataset is written as .npz files."""
import types

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


def test_backcolorwhite_and_missing_h5py_message(dataset, tmp_path):
    info, listinfo = dataset
    ds = D.Pt_sdf_img(_flags(backcolorwhite=True), listinfo=listinfo, info=info, shuffle=False, seed=1)
    assert np.all(ds.get_batch(0)["img"][:, :10] == 1.0)
    (tmp_path / "x.h5").write_bytes(b"\x89HDF\r\n\x1a\n")
    with pytest.raises(RuntimeError, match="h5py"):
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
