"""CPU tests of the oracle: against the reference-executed pins, the survey KATs, the committed
oracle KATs, and independent torch ops.  (-m "not gpu")"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import disn_oracle as O


# ---- pinned by the reference's own code (tests/golden/make_golden.py:reference_pins) ---------
def test_to_binary_matches_reference_bytes(pins):
    blob = O.to_binary(int(pins["dist_res"]), pins["dist_pos"], pins["dist_vals"])
    assert np.array_equal(np.frombuffer(blob, np.uint8), pins["dist_bytes"])


def test_split_plan_matches_reference(pins):
    for r, total, split, nsp in pins["split_plans"]:
        t, s, n, pad = O.split_plan(int(r))
        assert (t, s, n) == (total, split, nsp)
        assert pad == s * n - t and 0 <= pad < s
    # SURVEY §8c chunking KAT
    assert O.split_plan(64) == (274625, 2, 137313, 1)
    assert O.split_plan(256) == (16974593, 80, 212183, 47)


def test_grid_points_match_reference(pins):
    for tag in ("a", "b"):
        pts = O.grid_points(pins["grid_%s_params" % tag], int(pins["grid_%s_res" % tag]))
        assert pts.dtype == np.float32
        assert np.array_equal(pts, pins["grid_%s_pts" % tag])


def test_blender_proj_matches_reference(pins):
    for (az, el, d), K, RT in zip(pins["cam_params"], pins["cam_K"], pins["cam_RT"]):
        k, rt = O.blender_proj(az, el, d)
        assert np.allclose(k, K, rtol=0, atol=1e-12)
        assert np.allclose(rt, RT, rtol=0, atol=1e-12)
    # intrinsics identity (preprocessing/create_img_h5.py:30-34)
    assert pins["cam_K"][0][0, 0] == 35.0 * 137 / 32 and pins["cam_K"][0][0, 2] == 68.5


def test_synth_trans_mat_projects_in_front_of_camera():
    tm = O.synth_trans_mat(30.0, 25.0, 0.8)
    assert tm.shape == (4, 3) and tm.dtype == np.float32
    pts = np.random.default_rng(0).uniform(-0.5, 0.5, (1, 256, 3)).astype(np.float32)
    homo = np.concatenate([pts[0], np.ones((256, 1), np.float32)], 1) @ tm
    assert (homo[:, 2] > 0).all()          # positive depth for an object-sized cloud
    xy = O.get_img_points(pts, tm[None])
    assert ((xy >= 0) & (xy <= 136)).all()


# ---- survey / committed KATs -------------------------------------------------------------------
def test_projection_kat(kat):
    xy = O.get_img_points(kat["proj_pts"], O.DEMO_TRANS_MAT)
    assert np.array_equal(xy, kat["proj_xy"])
    assert np.allclose(xy[0], kat["proj_xy_survey"], rtol=0, atol=2e-5)   # values quoted in SURVEY §8c
    grid = O.grid_points([-1, -1, -1, 1, 1, 1], 64)
    gxy = O.get_img_points(grid[None], O.DEMO_TRANS_MAT)[0]
    clamp = ((gxy == 0) | (gxy == 136)).any(1).mean()
    assert abs(clamp - 0.0539) < 5e-4                                     # "5.39 % of points hit the clamp"


def test_projection_nan_policy():
    tm = np.zeros((1, 4, 3), np.float32)      # p = 0 -> 0/0
    xy = O.get_img_points(np.ones((1, 2, 3), np.float32), tm)
    assert np.isnan(xy).all()
    out = O.resampler(np.ones((1, 137, 137, 4), np.float32), xy)
    assert (out == 0).all()                  # documented choice: NaN -> outside -> zero features


def test_resize_kat_and_index_facts(kat):
    for hin, hout in ((14, 137), (224, 137), (137, 224), (28, 137)):
        out = O.resize_bilinear_legacy(kat["resize_%d_%d_in" % (hin, hout)], hout, hout)
        assert np.array_equal(out, kat["resize_%d_%d_out" % (hin, hout)])
    # Appendix A.3: rows with hi == lo (clamped at the last source row)
    for size, n_clamped in ((224, 0), (112, 1), (56, 2), (28, 4), (14, 9)):
        lo, hi, _ = O.resize_index_table(size, 137)
        assert int((lo == hi).sum()) == n_clamped
    lo, hi, _ = O.resize_index_table(224, 137)
    assert lo.max() == 222 and hi.max() == 223


def test_resize_against_independent_bilinear():
    rng = np.random.default_rng(3)
    for hin, hout in ((14, 137), (224, 137), (137, 224), (56, 137)):
        a = rng.random((2, hin, hin, 3), dtype=np.float32)
        o = O.resize_bilinear_legacy(a, hout, hout)
        src = np.minimum(np.arange(hout, dtype=np.float32) * (np.float32(hin) / np.float32(hout)), hin - 1)
        gy, gx = np.meshgrid(src, src, indexing="ij")
        grid = np.stack([2 * gx / (hin - 1) - 1, 2 * gy / (hin - 1) - 1], -1)[None].repeat(2, 0).astype(np.float32)
        t = F.grid_sample(torch.from_numpy(a).permute(0, 3, 1, 2), torch.from_numpy(grid), mode="bilinear",
                          padding_mode="border", align_corners=True).permute(0, 2, 3, 1).numpy()
        assert np.abs(o - t).max() < 2e-5


def test_resampler_kat_and_grid_sample(kat):
    out = O.resampler(kat["resampler_data"], kat["resampler_warp"])
    assert np.array_equal(out, kat["resampler_out"])
    d, w = kat["resampler_data"], kat["resampler_warp"]
    # edge cases: (0,0) exact pixel; (136,136) cx=137 out of range with weight 0; outside -> 0
    assert np.array_equal(out[0, 0], d[0, 0, 0]) and np.array_equal(out[0, 1], d[0, 136, 136])
    assert np.array_equal(out[0, 2], d[0, 0, 136]) and (out[0, 4] != 0).any()   # x=136.5 < W: half weight
    gx = 2 * w / (137 - 1) - 1
    t = F.grid_sample(torch.from_numpy(d).permute(0, 3, 1, 2), torch.from_numpy(gx)[:, None], mode="bilinear",
                      padding_mode="zeros", align_corners=True)[0, :, 0].T.numpy()
    assert np.abs(out[0] - t).max() < 2e-5


def test_conv_and_pool_against_numpy():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 9, 7, 5)).astype(np.float32)
    w = rng.standard_normal((3, 3, 5, 6)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    assert np.abs(O.conv2d(x, w, b) - O.conv2d_numpy(x, w, b, "SAME", True)).max() < 2e-5
    w7 = rng.standard_normal((7, 7, 5, 4)).astype(np.float32)
    x7 = rng.standard_normal((1, 7, 7, 5)).astype(np.float32)
    assert np.abs(O.conv2d(x7, w7, b[:4], "VALID", False) - O.conv2d_numpy(x7, w7, b[:4], "VALID", False)).max() < 5e-5
    p = O.max_pool_2x2(x[:, :8, :6])
    assert p.shape == (2, 4, 3, 5) and p[0, 0, 0, 0] == x[0, :2, :2, 0].max()


def test_variable_namespace():
    shp = O.variable_shapes()
    assert shp["vgg_16/conv1/conv1_1/weights"] == (3, 3, 3, 64)
    assert shp["vgg_16/fc6/weights"] == (7, 7, 512, 4096)
    assert shp["sdfprediction/fold2/conv1/weights"] == (1, 1, 1536, 512)
    assert shp["sdfprediction_imgfeat/fold2/conv1/weights"] == (1, 1, 1984, 512)
    n = sum(int(np.prod(s)) for s in shp.values())
    assert n == 138455872 + 2363394 - 0 or n == 140819266     # VGG 138 455 872 + decoder 2 363 394


def test_mlp_small_known_answer():
    # hand-checkable: zero weights except biases -> pred = b6 for both streams
    W = {k: np.zeros(s, np.float32) for k, s in O.variable_shapes().items()}
    W["sdfprediction/fold2/conv5/biases"][:] = 0.25
    W["sdfprediction_imgfeat/fold2/conv5/biases"][:] = -1.5
    pts = np.zeros((1, 3, 3), np.float32)
    g = O.get_sdf_basic2(pts, np.zeros((1, 1024), np.float32), W)
    l = O.get_sdf_basic2_imgfeat_twostream(pts, np.zeros((1, 3, 1, 1472), np.float32), W)
    assert g.shape == (1, 3, 1) and (g == 0.25).all() and (l == -1.5).all()


@pytest.mark.timeout(600)
def test_full_model_kat(kat):
    """cfg2 (seed 0, 2048 pts) reproduces the committed fp32 prediction and stays within the
    fp32 noise floor of the fp64 shadow."""
    W = O.init_weights(0, "he")
    ep = O.get_model(O.synth_inputs(0, 1, 2048), W)
    assert np.array_equal(ep["sample_img_points"], kat["cfg2_he_xy"])
    assert np.abs(ep["pred_sdf"] - kat["cfg2_he_pred"]).max() < 5e-5      # BLAS thread-count dependent order
    assert np.abs(ep["pred_sdf"] - kat["cfg2_he_pred64"]).max() < 5e-5
    assert np.abs(kat["cfg2_he_pred"]).mean() > 0.1                       # He weights keep activations O(1)


def test_loss_formula():
    pred = np.array([[[1.0], [-2.0], [0.5]]], np.float32)
    gt = np.array([[[0.2], [-0.1], [0.005]]], np.float32)
    L = O.get_loss(pred, gt)
    assert L["accuracy"] == 1.0
    # |10*gt - pred| * mask, mask = 4 where gt <= 0.01
    expect = np.mean([abs(2.0 - 1.0) * 1, abs(-1.0 + 2.0) * 4, abs(0.05 - 0.5) * 4]) * 1000
    assert abs(L["sdf_loss"] - expect) < 1e-3


def test_fold_identities_behind_the_folded_local_stream():
    """The two linear-algebra facts disn_fold_local / project_gather_taps rest on, in oracle terms:
    (1) resampling commutes with a right-multiplication of the map: resampler(map) @ W == resampler(map @ W)
        (models/model_normalization.py:172-190 gather, models/sdfnet.py:180-182 first local fold2 layer);
    (2) resampling the legacy-resized tap == resampling a map that holds the resized tap only at the touched
        pixels (trivially), and the legacy resize is itself linear in the tap: resize(tap @ W) == resize(tap) @ W."""
    rng = np.random.default_rng(11)
    fm = rng.standard_normal((1, 137, 137, 24)).astype(np.float32)
    W = rng.standard_normal((24, 7)).astype(np.float32)
    xy = np.concatenate([rng.uniform(-2, 139, (1, 300, 2)),
                         np.array([[[0, 0], [136, 136], [136.0, 0.5], [-0.5, 10], [12.25, 136.9]]])], 1).astype(np.float32)
    a = O.resampler(fm, xy).astype(np.float64) @ W.astype(np.float64)
    pm = (fm.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
    b = O.resampler(pm, xy).astype(np.float64)
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
    tap = rng.standard_normal((1, 14, 14, 24)).astype(np.float32)
    r1 = O.resize_bilinear_legacy(tap, 137, 137).astype(np.float64) @ W.astype(np.float64)
    r2 = O.resize_bilinear_legacy((tap.astype(np.float64) @ W.astype(np.float64)).astype(np.float32), 137, 137).astype(np.float64)
    assert np.abs(r1 - r2).max() <= 2e-5 * max(1.0, np.abs(r1).max())


def test_multithreaded_resize_and_resampler_are_bit_identical():
    """bench.py's cpu_baseline leg times torch-CPU (threaded) forms of the two numpy-bound oracle stages;
    they must be the same functions: same expressions, same rounding order"""
    rng = np.random.default_rng(5)
    for h, c, o in ((224, 8, 137), (14, 16, 137), (137, 3, 224), (56, 4, 137)):
        x = rng.standard_normal((2, h, h, c)).astype(np.float32)
        assert np.array_equal(O.resize_bilinear_legacy(x, o, o), O.resize_bilinear_legacy_mt(x, o, o))
        for workers in (1, 3):     # the two-pass row-block form the cpu_baseline leg threads over
            assert np.array_equal(O.resize_bilinear_legacy(x, o, o), O.resize_bilinear_legacy_blocks(x, o, o, workers))
    m = rng.standard_normal((2, 137, 137, 24)).astype(np.float32)
    w = (rng.random((2, 500, 2)) * 141 - 2).astype(np.float32)
    w[0, :4] = [[0, 0], [136, 136], [136, 0], [np.nan, 1]]
    assert np.array_equal(O.resampler(m, w), O.resampler_mt(m, w))


def _external_kat():
    import json
    import os
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "external_kat.json")))


def test_external_kat_tensorflow_resize_vectors():
    """TensorFlow's own unit-test vectors for the legacy bilinear resize (image_ops_test.py,
    ResizeImagesTest): the one pin of the TF-executed arithmetic that does not come from this repo"""
    k = _external_kat()
    names = ["resize_up", "resize_down"] + sorted(n for n in k if n.startswith("resize_exact"))
    assert len(names) >= 7          # + the vectors derived with exact rational arithmetic (make_external_resize.py):
    for nm in names:                # the 137 -> 224, 14 -> 137 and 224 -> 137 rows of the path and two small grids
        x = np.asarray(k[nm]["in"], np.float32).reshape(k[nm]["in_shape"])
        oh, ow = k[nm]["out_hw"]
        assert np.array_equal(O.resize_bilinear_legacy(x, oh, ow).ravel(), np.asarray(k[nm]["out"], np.float32)), nm


def test_external_kat_tensorflow_conv2d_vectors():
    """TensorFlow's own conv2d unit-test vectors (conv_ops_test.py: testConv2D1x1Filter, testConv2D2x2Filter): the
    orientation of the convolution rows B of the path are built on -- cross-correlation, NHWC x HWIO, channel-minor"""
    k = _external_kat()
    for nm in ("conv2d_tf_1x1", "conv2d_tf_2x2"):
        c = k[nm]
        x = np.arange(1, 1 + int(np.prod(c["in_shape"])), dtype=np.float32).reshape(c["in_shape"])
        w = np.arange(1, 1 + int(np.prod(c["filter_shape"])), dtype=np.float32).reshape(c["filter_shape"])
        b = np.zeros(c["filter_shape"][-1], np.float32)
        want = np.asarray(c["out"], np.float32)
        for dt in (np.float32, np.float64):
            got = O.conv2d_numpy(x, w, b, c["padding"], False, dtype=dt)
            assert np.array_equal(np.asarray(got, np.float64).ravel(), want.astype(np.float64)), (nm, dt)
        assert np.array_equal(np.asarray(O.conv2d(x, w, b, c["padding"], False)).ravel(), want), nm
    mp = k["maxpool_tf_valid"]       # pooling_ops_test.py: the 2 x 2 / stride-2 pool of rows B (VALID drops the odd row / column)
    x = np.arange(1, 1 + int(np.prod(mp["in_shape"])), dtype=np.float32).reshape(mp["in_shape"])
    assert np.array_equal(O.max_pool_2x2(x).ravel(), np.asarray(mp["out"], np.float32))


def test_external_kat_hand_derived_resampler_cases():
    k = _external_kat()
    m = np.asarray(k["resampler_2x2"]["map"], np.float32)[None, :, :, None]
    got = O.resampler(m, np.asarray(k["resampler_2x2"]["xy"], np.float32)[None])[0, :, 0]
    assert np.array_equal(got, np.asarray(k["resampler_2x2"]["out"], np.float32))
    yy, xx = np.meshgrid(np.arange(137), np.arange(137), indexing="ij")
    m = (1 + xx + 1000 * yy).astype(np.float32)[None, :, :, None]
    got = O.resampler(m, np.asarray(k["resampler_137"]["xy"], np.float32)[None])[0, :, 0]
    assert np.array_equal(got, np.asarray(k["resampler_137"]["out"], np.float32))


def test_grid_float32_params_caveat():
    """with FLOAT32 sdf_params numpy 1.x rounds the linspace step to float32.  The product / oracle grid
    (float64 step; what numpy 1.x computes for the ints of demo/demo.py:278 and for float64 params) is identical
    to that for dyadic boxes (the +-1 box at a power-of-two resolution) and otherwise drifts by the accumulated
    step rounding: |difference| <= R * ulp32(step) / 2 + one float32 rounding -- about 1e-7 of the box size."""
    for box, R in (([-1, -1, -1, 1, 1, 1], 64), ([-1, -1, -1, 1, 1, 1], 256), ([-0.5, -1, -2, 0.5, 1, 2], 32)):
        assert np.array_equal(O.grid_points(box, R), O.grid_points_numpy1_float32(box, R))
    box = np.asarray([-0.83, -0.91, -0.77, 0.79, 0.95, 0.81], np.float32)
    R = 100
    a, b = O.grid_points(box, R), O.grid_points_numpy1_float32(box, R)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=0)
    step = (box[3:].astype(np.float64) - box[:3]) / R
    bound = R * np.spacing(step.astype(np.float32)).astype(np.float64) / 2 + np.spacing(np.float32(1.0))
    print("non-dyadic float32 box, R = 100: max |difference| per axis %s (bound %s)" % (d, bound))
    assert (d <= bound).all() and d.max() > 0 and d.max() < 2e-7 * 2.0


def test_cfg3_cfg4_goldens_reproduce_from_the_oracle():
    """tests/golden/cfg3_strided.npz / cfg4_sampled.npz (make_golden_cfg34.py) are what the float64 oracle gives:
    a subset is recomputed here (one float64 encode: cfg4's image 1 is cfg3's demo image) and must match exactly"""
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_golden_cfg34 as G
    g3 = np.load(os.path.join(GOLDEN, "cfg3_strided.npz"))
    g4 = np.load(os.path.join(GOLDEN, "cfg4_sampled.npz"))
    assert g3["pred64"].shape == (len(range(0, G.TOTAL, G.CFG3_STRIDE)) + 1,) and g4["pred64"].shape == (8, 8288)
    assert int(g3["stride"]) == G.CFG3_STRIDE and int(g4["stride"]) == G.CFG4_STRIDE and int(g3["weight_seed"]) == G.WEIGHT_SEED
    W = O.init_weights(G.WEIGHT_SEED, "he")
    c3, c4 = G.cfg3_inputs(), G.cfg4_inputs()
    assert np.array_equal(c4["imgs"][1], c3["img"][0]) and len({im.tobytes() for im in c4["imgs"]}) == 8
    _, emb, maps, _ = O.encode(c3["img"], W, np.float64)
    sel = np.arange(0, c3["idx"].size, 4099)
    pts = np.concatenate([G.grid_points_at(c3["sdf_params"], G.RES, c3["idx"][sel]), np.zeros((1, 3), np.float32)])
    got = G.oracle_pred64(W, emb, maps, c3["trans_mat"], pts)
    assert np.allclose(got[:-1], g3["pred64"][sel], rtol=0, atol=1e-12) and abs(got[-1] - g3["pred64"][-1]) <= 1e-12
    sel4 = np.arange(0, 8288, 257)
    pts4 = G.grid_points_at(c4["sdf_params"][1], G.RES, c4["idx"][1][sel4])
    got4 = G.oracle_pred64(W, emb, maps, c4["trans_mat"][1:2], pts4)
    assert np.allclose(got4, g4["pred64"][1][sel4], rtol=0, atol=1e-12)
    # the sampled grid points are the reference grid's points (test/create_sdf.py:246-256), bit for bit
    small = O.grid_points(c4["sdf_params"][3], 16)
    assert np.array_equal(G.grid_points_at(c4["sdf_params"][3], 16, np.arange(17 ** 3)), small)
