"""GPU parity tests of the whole path through the drop-in surface
(disn_amd.model_normalization + graph.Session) against the committed golden vectors and the
oracle: BASELINE configs 1-3 and the size-independent properties at full size."""
import numpy as np
import pytest
import torch

from conftest import internal_arrays, report_close
from oracle import disn_oracle as O

pytestmark = pytest.mark.gpu

# The bar of the GEMM-shaped part of the path (north_star: "within 1e-5 of the reference TF CPU path"):
#   * what the path RETURNS (pred_sdf, the two stream sums, the embedding):  |gpu - f64| <= 1e-5 ABSOLUTE
#     against the float64 shadow of the oracle on the He-weight configs (|pred| ~ 1.7; measured on MI355X
#     3-5e-6, so a 2x regression fails), and against the float32 CPU path
#     |gpu - oracle32| <= |oracle32 - f64| + 1e-5 (two float32 evaluations that sum up to 25088-term dot
#     products in different orders differ from each other by their own distances to the truth);
#   * intermediate tensors (taps, feature map: values up to O(10)): 1e-5 + 1e-5 |ref|.
PRED_ATOL = 1e-5
ATOL, RTOL = 1e-5, 1e-5


def _session(mode):
    from disn_amd.graph import Session
    from disn_amd.weights import WeightStore
    return Session(WeightStore.random_init(0, mode=mode))


def _graph(B, N):
    import disn_amd.model_normalization as model
    pls = model.placeholder_inputs(B, 1, (137, 137), num_sample_pc=N)
    ep = model.get_model(pls, 1, None, bn=False)
    loss, ep = model.get_loss(ep, num_sample_points=N, batch_size=B)
    return pls, ep


def _feed(pls, d):
    return {pls[k]: d[k] for k in ("imgs", "sample_pc", "sample_pc_rot", "trans_mat") if k in d}


@pytest.mark.parametrize("mode", ["he", "xavier"])
def test_cfg2_against_golden(kat, mode):
    """BASELINE config 2: VGG-16 encode + 2048 query points, seed-0 inputs, both weight sets."""
    sess = _session(mode)
    pls, ep = _graph(1, 2048)
    feed = O.synth_inputs(0, 1, 2048)
    pred, xy, emb = sess.run([ep["pred_sdf"], ep["sample_img_points"], ep["img_embedding"]], _feed(pls, feed))
    assert pred.shape == (1, 2048, 1) and xy.shape == (1, 2048, 2) and emb.shape == (1, 1024)
    assert np.array_equal(xy, kat["cfg2_%s_xy" % mode])                                   # row D bit-exact
    e_emb = report_close("embedding(%s)" % mode, emb, kat["cfg2_%s_emb64" % mode], PRED_ATOL)
    e_pred = report_close("pred_sdf(%s)" % mode, pred, kat["cfg2_%s_pred64" % mode], PRED_ATOL)
    e32 = float(np.abs(pred - kat["cfg2_%s_pred" % mode]).max())
    o32 = float(np.abs(kat["cfg2_%s_pred" % mode].astype(np.float64) - kat["cfg2_%s_pred64" % mode]).max())
    print("\n[parity cfg2 %s] max|gpu-f64| pred %.3g emb %.3g ; max|gpu-oracle32| %.3g ; max|oracle32-f64| %.3g ; "
          "|pred| mean %.3g" % (mode, e_pred, e_emb, e32, o32, float(np.abs(pred).mean())))
    assert e32 <= o32 + 1e-5, "GPU is further from the float32 CPU path than that path's own error allows"
    # The number itself, asserted (VERDICT r2 #5c).  north_star's bar reads "within 1e-5 of the reference TF CPU path";
    # the float32 CPU oracle stands for that path and is itself 8.2e-6 away from the float64 truth on this case (o32),
    # because it sums K <= 25088-term dot products in fp32 in its own order.  The GPU result is CLOSER to the truth
    # (5e-6) and the two fp32 results differ by up to the sum of their errors: measured 1.1e-5 on He weights.  What is
    # asserted: |gpu - f64| <= 1e-5 (above), |gpu - oracle32| <= |gpu - f64| + |oracle32 - f64| (triangle, the line
    # above is its weaker form) and the measured distance itself <= 1.5e-5 so that a regression shows.
    assert e32 <= e_pred + o32 + 1e-7
    assert e32 <= 1.5e-5


def test_cfg2_every_end_point_vs_oracle():
    sess = _session("he")
    pls, ep = _graph(1, 512)
    feed = O.synth_inputs(3, 1, 512)
    feed["trans_mat"] = O.synth_trans_mat(201.5, 30.0, 0.65)[None]
    W = sess.weights.arrays
    ref = O.get_model(feed, W)
    ref64 = O.get_model(feed, W, dtype=np.float64)
    keys = ["pred_sdf", "ref_img", "sample_img_points", "resized_ref_img", "img_embedding",
            "pred_sdf_value_global", "pred_sdf_value_local", "point_img_feat", "ref_feats_embedding_cnn"]
    vals = dict(zip(keys, sess.run([ep[k] for k in keys], _feed(pls, feed))))
    assert np.array_equal(vals["ref_img"], feed["imgs"])                     # un-resized (Appendix C #4)
    assert np.array_equal(vals["resized_ref_img"], ref["resized_ref_img"])   # row A bit-exact
    assert np.array_equal(vals["sample_img_points"], ref["sample_img_points"])
    report_close("img_embedding", vals["img_embedding"], ref64["img_embedding"], PRED_ATOL)
    # point_img_feat: the gather is bit-exact given the taps; the taps carry conv rounding
    report_close("point_img_feat", vals["point_img_feat"], ref64["point_img_feat"], ATOL, RTOL)
    report_close("global", vals["pred_sdf_value_global"], ref64["pred_sdf_value_global"], PRED_ATOL)
    report_close("local", vals["pred_sdf_value_local"], ref64["pred_sdf_value_local"], PRED_ATOL)
    e = report_close("pred", vals["pred_sdf"], ref64["pred_sdf"], PRED_ATOL)
    e32 = float(np.abs(vals["pred_sdf"] - ref["pred_sdf"]).max())
    o32 = float(np.abs(ref["pred_sdf"].astype(np.float64) - ref64["pred_sdf"]).max())
    print("\n[parity every end point] max|gpu-f64| %.3g ; max|gpu-oracle32| %.3g ; max|oracle32-f64| %.3g" % (e, e32, o32))
    assert e32 <= o32 + 1e-5


def test_vgg_taps_vs_oracle():
    """rows B/C layer by layer: every tap and the embedding, B=2 (exercises the batch index math)"""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(0, mode="he")
    eng = SdfEngine(store)
    imgs = np.random.default_rng(9).random((2, 137, 137, 3), dtype=np.float32)
    enc = eng.encode(imgs)
    resized, emb64, maps, eps = O.encode(imgs, store.arrays, dtype=np.float64)
    assert np.array_equal(enc.resized.cpu().numpy(), resized)
    # (the engine computes in equalised units -- DeviceWeights(equalise=True): true_taps / true_features convert)
    assert eng.weights.status["equalised"]
    for t, nm in zip(eng.true_taps(enc), O.TAP_NAMES):
        report_close(nm, t.cpu().numpy(), eps["vgg_16/%s/%s" % (nm[:5], nm)], ATOL, RTOL)
    report_close("embedding", enc.embedding.cpu().numpy(), emb64, ATOL, RTOL)
    fm = eng.true_features(enc.featmap).cpu().numpy()
    ref_fm = np.concatenate(maps, axis=3)
    report_close("featmap", fm, ref_fm, ATOL, RTOL)


def test_cfg1_demo_slice_against_golden(kat):
    """BASELINE config 1 fixture (demo PNG, GT trans_mat, bbox [-1,1]^3, res 64): a 4096-point
    slice of the 65^3 grid through the dense-grid driver, compared with the golden oracle run."""
    from disn_amd import create_sdf as cs
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    eng = SdfEngine(WeightStore.random_init(0, mode="he"))
    img = kat["demo_img"].astype(np.float32) / np.float32(255.0)
    enc = eng.encode(img)
    k0 = int(kat["demo_k0"])
    out = cs.dense_grid_sdf(eng, enc, 0, O.DEMO_TRANS_MAT, [-1, -1, -1, 1, 1, 1], 64, k_range=(k0, k0 + 4096))
    got = out.cpu().numpy()
    report_close("demo slice (pred/10)", got, kat["demo_pred64"] / 10.0, ATOL / 10, RTOL)
    # un-divided value through the same entry
    out1 = cs.dense_grid_sdf(eng, enc, 0, O.DEMO_TRANS_MAT, [-1, -1, -1, 1, 1, 1], 64, sdf_weight=1.0,
                             k_range=(k0, k0 + 4096)).cpu().numpy()
    assert np.array_equal(got, (out1 / np.float32(10.0)).astype(np.float32))   # IEEE float32 division


def test_session_loop_like_create_sdf():
    """the reference's per-split sess.run loop (test/create_sdf.py:262-285) runs unchanged and
    equals the device-side dense-grid driver"""
    from disn_amd import create_sdf as cs
    sess = _session("he")
    R = 16
    total, split, nsp, pad = cs.split_plan(R)
    assert split == 1
    pls, ep = _graph(1, nsp)
    rng = np.random.default_rng(2)
    img = rng.random((1, 137, 137, 3), dtype=np.float32)
    sdf_params = np.array([-0.9, -0.8, -0.7, 0.9, 0.8, 0.7], np.float32)
    pts = np.concatenate([cs.grid_points_host(sdf_params, R), np.zeros((pad, 3), np.float32)], 0).reshape(split, 1, nsp, 3)
    acc = np.zeros((split, 1, nsp, 1))
    for sp in range(split):
        feed = {pls["sample_pc"]: pts[sp], pls["sample_pc_rot"]: pts[sp], pls["imgs"]: img,
                pls["trans_mat"]: O.DEMO_TRANS_MAT}
        pred, ref_img, xy = sess.run([ep["pred_sdf"], ep["ref_img"], ep["sample_img_points"]], feed_dict=feed)
        acc[sp] = pred
    result = (acc.reshape(1, -1, 1)[:, :total, :] / 10.0)[0, :, 0]
    dev_res = cs.create_sdf(sess.engine, img, O.DEMO_TRANS_MAT, sdf_params[None], R)[0].cpu().numpy()
    report_close("session loop vs device driver", dev_res, result, 2e-6, 1e-6)


def test_cfg1_full_demo_grid_against_golden(kat):
    """BASELINE config 1 IN FULL (demo/demo.py:263-339: the chair image, the ground-truth camera, sdf_res 64): all
    274 625 grid points in .dist order plus the pad point of the demo's padded splits, against the float64 oracle run
    stored in tests/golden/cfg1_full65.npz (tests/golden/make_golden_cfg1.py) -- through (a) the reference's own
    per-split sess.run loop with its zero padding (demo/demo.py:292-328), (b) the device-side dense-grid driver."""
    import os
    from disn_amd import create_sdf as cs
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_full65.npz"))["pred64"]
    R = 64
    total, split, nsp, pad = cs.split_plan(R)
    assert total == 65 ** 3 == gold.size - 1 and pad >= 1
    img = (kat["demo_img"].astype(np.float32) / np.float32(255.0))
    sess = _session("he")
    pls, ep = _graph(1, nsp)
    sdf_params = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    pts = np.concatenate([cs.grid_points_host(sdf_params, R), np.zeros((pad, 3), np.float32)], 0).reshape(split, 1, nsp, 3)
    acc = np.zeros((split, 1, nsp, 1), np.float32)
    for sp in range(split):                                   # demo/demo.py:311-326
        feed = {pls["sample_pc"]: pts[sp], pls["sample_pc_rot"]: pts[sp], pls["imgs"]: img,
                pls["trans_mat"]: O.DEMO_TRANS_MAT}
        acc[sp] = sess.run(ep["pred_sdf"], feed_dict=feed)
    flat = acc.reshape(-1)
    err = float(np.abs(flat[:total] - gold[:total]).max())
    err_pad = float(np.abs(flat[total:] - gold[total]).max())   # every pad point is (0, 0, 0)
    print("cfg1 full grid, session loop: max |gpu - f64| %.3g over %d points (|pred| max %.3g); pad point %.3g" % (
        err, total, float(np.abs(gold).max()), err_pad))
    assert err <= PRED_ATOL and err_pad <= PRED_ATOL
    dev = cs.create_sdf(sess.engine, img, O.DEMO_TRANS_MAT, sdf_params[None], R)[0].cpu().numpy()
    err_dev = float(np.abs(dev.astype(np.float64) * 10.0 - gold[:total]).max())
    print("cfg1 full grid, device driver (fused point MLP, pred / 10): max |10 gpu - f64| %.3g" % err_dev)
    assert err_dev <= PRED_ATOL


def test_full_size_grid_properties():
    """BASELINE config 3 size (257^3 = 16 974 593 points): properties that need no oracle run.
    (a) any slice of the full-grid result equals the same slice evaluated alone;
    (b) a strided sample of points agrees with the oracle's float64 MLP on the GPU's features;
    (c) order: flat (iz,iy,ix) -- the first / last points are the bbox corners."""
    from disn_amd import create_sdf as cs
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(0, mode="he")
    eng = SdfEngine(store)
    img = O.synth_inputs(0, 1, 8)["imgs"]
    enc = eng.encode(img)
    R = 256
    total = (R + 1) ** 3
    sp = [-1, -1, -1, 1, 1, 1]
    full = cs.dense_grid_sdf(eng, enc, 0, O.DEMO_TRANS_MAT, sp, R)
    torch.cuda.synchronize()
    assert full.shape == (total,) and bool(torch.isfinite(full).all())
    # (a) slices -- including one that straddles internal chunk boundaries and the ragged tail
    for k0, k1 in ((0, 1000), (65536 - 10, 65536 + 5000), (total - 77777, total)):
        part = cs.dense_grid_sdf(eng, enc, 0, O.DEMO_TRANS_MAT, sp, R, k_range=(k0, k1))
        report_close("slice [%d,%d)" % (k0, k1), part.cpu().numpy(), full[k0:k1].cpu().numpy(), 2e-6, 1e-6)
    # (b) strided sample vs oracle MLP (float64) on the GPU's own feature map / embedding
    idx = np.arange(0, total, 40009)
    from disn_amd import ops
    pts = torch.cat([ops.grid_points(sp, R, int(k), int(k) + 1, "cuda") for k in idx])[None]
    xy = O.get_img_points(pts.cpu().numpy(), O.DEMO_TRANS_MAT)
    fm = eng.true_features(enc.featmap).cpu().numpy()
    feat = O.resampler(fm, xy)[:, :, None, :]
    emb = enc.embedding.cpu().numpy()
    ref = (O.get_sdf_basic2(pts.cpu().numpy(), emb, store.arrays, dtype=np.float64)
           + O.get_sdf_basic2_imgfeat_twostream(pts.cpu().numpy(), feat, store.arrays, dtype=np.float64))[0, :, 0] / 10.0
    report_close("strided sample", full[torch.from_numpy(idx).cuda()].cpu().numpy(), ref, ATOL / 10, RTOL)
    # (c) corners
    c = ops.grid_points(sp, R, 0, total, "cuda")
    assert c[0].tolist() == [-1.0, -1.0, -1.0] and c[-1].tolist() == [1.0, 1.0, 1.0]
    assert c[1].tolist()[1:] == [-1.0, -1.0] and c[1, 0] > -1.0        # x fastest


def test_get_decoder_and_standalone_streams():
    import disn_amd.model_normalization as model
    sess = _session("he")
    N = 300
    pls = model.placeholder_inputs(1, 1, (137, 137), num_sample_pc=N)
    fpl = model.placeholder_features(1, num_sample_pc=N)
    out = model.get_decoder(N, pls, fpl)
    rng = np.random.default_rng(4)
    pc = rng.uniform(-1, 1, (1, N, 3)).astype(np.float32)
    emb = rng.standard_normal((1, 1, 1, 1024)).astype(np.float32)
    feat = np.maximum(rng.standard_normal((1, N, 1, 1472)), 0).astype(np.float32)
    got = sess.run(out, {pls["sample_pc_rot"]: pc, fpl["ref_feats_embedding_cnn"]: emb, fpl["point_img_feat"]: feat})
    W = sess.weights.arrays
    ref = (O.get_sdf_basic2(pc, emb.reshape(1, -1), W, dtype=np.float64)
           + O.get_sdf_basic2_imgfeat_twostream(pc, feat, W, dtype=np.float64))
    report_close("get_decoder", got, ref, PRED_ATOL)


def test_losses_vs_oracle():
    sess = _session("he")
    pls, ep = _graph(1, 256)
    feed = O.synth_inputs(5, 1, 256)
    gt = (np.random.default_rng(1).standard_normal((1, 256, 1)) * 0.05).astype(np.float32)
    fd = _feed(pls, feed); fd[pls["sdf"]] = gt
    names = ["accuracy", "sdf_loss_realvalue", "sdf_loss", "regularization", "overall_loss"]
    vals = sess.run([ep["losses"][n] for n in names] + [ep["pred_sdf"]], fd)
    pred = vals[-1]
    ref = O.get_loss(pred, gt, sess.weights.arrays)
    for n, v in zip(names, vals[:-1]):
        assert abs(float(v) - ref[n]) <= 1e-4 * max(1.0, abs(ref[n])), (n, float(v), ref[n])


def test_encoder_cache_and_rerun_flag():
    from disn_amd.graph import Session
    from disn_amd.weights import WeightStore
    pls, ep = _graph(1, 64)
    feed = O.synth_inputs(0, 1, 64)
    a = Session(WeightStore.random_init(0, mode="he"), cache_encoder=True)
    p1 = a.run(ep["pred_sdf"], _feed(pls, feed))
    e1 = a._enc_val
    p2 = a.run(ep["pred_sdf"], _feed(pls, feed))
    assert a._enc_val is e1 and np.array_equal(p1, p2)
    feed2 = dict(feed); feed2["imgs"] = feed["imgs"][:, ::-1].copy()
    a.run(ep["pred_sdf"], _feed(pls, feed2))
    assert a._enc_val is not e1                                  # different image bytes -> re-encode
    b = Session(WeightStore.random_init(0, mode="he"), cache_encoder=False)
    assert np.array_equal(b.run(ep["pred_sdf"], _feed(pls, feed)), p1) and b._enc_val is None


@pytest.mark.parametrize("B,N", [(1, 2048), (2, 700), (3, 64)])
def test_encode_query_equals_encode_plus_query(B, N):
    """disn_encode_query (two streams, fork/join with events) == disn_encode + disn_query,
    bit for bit, on every repetition: a missing dependency between the streams would show up as
    a mismatch on some run."""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    eng = SdfEngine(WeightStore.random_init(0, mode="he"))
    rng = np.random.default_rng(B * 100 + N)
    tms = np.stack([O.DEMO_TRANS_MAT[0], O.synth_trans_mat(30, 25, 0.8), O.synth_trans_mat(201.5, 30, 0.65)])[:B]
    for rep in range(6):
        imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
        pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
        enc_a, sdf_a = eng.encode_query(imgs, pts, tms)
        enc_b = eng.encode(imgs)
        sdf_b = eng.query(enc_b, pts, tms)
        torch.cuda.synchronize()
        assert torch.equal(enc_a.embedding, enc_b.embedding), "embedding differs (rep %d)" % rep
        assert enc_a.featmap is None          # the default never writes the 110 MB/image map ...
        sdf_k = eng.encode_query(imgs, pts, tms, keep_featmap=True)
        assert torch.equal(sdf_k[1], sdf_a), "gather from taps != gather from the map (rep %d)" % rep
        assert torch.equal(sdf_k[0].featmap, enc_b.featmap), "featmap differs (rep %d)" % rep
        assert torch.equal(eng.featmap_of(enc_a), enc_b.featmap)   # ... and builds it on demand
        for ta, tb in zip(enc_a.taps, enc_b.taps):
            assert torch.equal(ta, tb)
        if B == 1:   # same row count -> same stream-K plan -> same summation order
            assert torch.equal(sdf_a, sdf_b), "pred_sdf differs (rep %d): max %g" % (rep, float((sdf_a - sdf_b).abs().max()))
        else:        # M = B*N vs N rows: another plan, fp32 summation-order noise only
            report_close("encode_query vs encode+query (rep %d)" % rep, sdf_a.cpu().numpy(), sdf_b.cpu().numpy(), ATOL, RTOL)
    ref = O.get_model({"imgs": imgs, "sample_pc": pts, "sample_pc_rot": pts, "trans_mat": tms},
                      eng.weights and WeightStore.random_init(0, mode="he").arrays, dtype=np.float64)
    report_close("encode_query vs oracle", sdf_a.cpu().numpy(), ref["pred_sdf"][..., 0], PRED_ATOL)


@pytest.mark.parametrize("conv_h2", [True, False])
def test_vgg_stack_equals_standalone_layer_chain(conv_h2):
    """The conv stack inside disn_encode == the same layers one by one through the unit entry points, bit for
    bit at every tap.  conv_h2 (engine default): disn_conv1_1 + disn_conv3x3_h2 with its fused pool (inside the
    encoder each layer's scale comes from the producer's atomic maximum, standalone from a pass over the
    input: the same number).  Otherwise: disn_conv3x3_x3 (split-K reduce that also emits the 2x2 max pool for
    conv2_2 .. conv5_3) + disn_maxpool2x2."""
    from disn_amd import ops
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore, VGG_CONV_NAMES
    store = WeightStore.random_init(3, mode="he")
    eng = SdfEngine(store, conv_h2=conv_h2)
    rng = np.random.default_rng(5)
    for B in (1, 2):
        imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
        enc = eng.encode(imgs)
        x = enc.resized
        tap = 0
        pool_after = {1, 3, 6, 9, 12}
        W_int = internal_arrays(store)          # the variables as the engine uploaded them (equalised copy)
        for i, nm in enumerate(VGG_CONV_NAMES):
            w = W_int[nm + "/weights"]
            kh, kw, ci, co = w.shape
            wd = torch.from_numpy(np.ascontiguousarray(w.reshape(kh * kw * ci, co))).cuda()
            bias = torch.from_numpy(W_int[nm + "/biases"]).cuda()
            pooled = None
            if conv_h2:
                if ci == 3:
                    x = ops.conv1_1(x, wd, bias, True)
                else:
                    x, pooled, _ = ops.conv3x3_h2(x, ops.pack_conv_h2(wd), bias, co, True, pool=i in pool_after,
                                                  want_amax=True)
            elif ci == 3:
                x = ops.conv3x3(x, ops.pack_kn(wd), bias, co, True)
            else:
                x = ops.conv3x3_x3(x, ops.pack_kn_x3(wd), bias, co, True)
            if i in pool_after:
                assert torch.equal(x, enc.taps[tap]), "tap %d (%s) differs, B=%d" % (tap, nm, B)
                tap += 1
                x = pooled if pooled is not None else ops.maxpool2x2(x)
        assert tap == 5


def test_pipelined_grid_equals_sequential():
    """disn_query_grid_ctx (gather of chunk i+1 on the aux stream under the MLP of chunk i, double
    buffer + events) == disn_query_grid, bit for bit, on a range with several chunks and a ragged
    tail, repeatedly (a buffer-reuse race would differ on some run)."""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    eng = SdfEngine(WeightStore.random_init(0, mode="he"))
    enc = eng.encode(O.synth_inputs(1, 1, 8)["imgs"])
    R, sp = 128, [-1, -0.9, -0.8, 1, 0.9, 0.8]
    k0, k1 = 1000, 1000 + 5 * 65536 + 4321
    ref = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, k0, k1, pipelined=False, fold=False).clone()
    for rep in range(4):
        got = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, k0, k1, pipelined=True)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), "rep %d: max diff %g" % (rep, float((got - ref).abs().max()))


def test_batched_images_equal_single_images():
    """BASELINE config 4's per-GPU shape: a batch of images through one encode / create_sdf equals the
    images processed one by one (different GEMM plans for M = B*H*W rows: fp32 noise only), and the
    sharded driver with a single rank equals the plain one exactly."""
    from disn_amd import create_sdf as cs, parallel as par
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    eng = SdfEngine(WeightStore.random_init(0, mode="he"))
    rng = np.random.default_rng(21)
    B, R = 4, 24
    imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
    tms = np.stack([O.DEMO_TRANS_MAT[0], O.synth_trans_mat(30, 25, 0.8), O.synth_trans_mat(201.5, 30, 0.65),
                    O.synth_trans_mat(310, 12.5, 0.9)])
    sps = np.array([[-1, -1, -1, 1, 1, 1], [-0.9, -0.8, -0.7, 0.9, 0.8, 0.7], [-1, -1, -1, 1, 1, 1],
                    [-0.5, -0.6, -0.7, 0.8, 0.9, 1.0]], np.float64)
    enc = eng.encode(imgs)
    for b in range(B):
        e1 = eng.encode(imgs[b:b + 1])
        report_close("embedding[%d]" % b, enc.embedding[b].cpu().numpy(), e1.embedding[0].cpu().numpy(), ATOL, RTOL)
        # (in the reference's units: the tolerance is an absolute one)
        report_close("featmap[%d]" % b, eng.true_features(enc.featmap[b]).cpu().numpy(),
                     eng.true_features(e1.featmap[0]).cpu().numpy(), ATOL, RTOL)
    batch = cs.create_sdf(eng, imgs, tms, sps, R)
    assert batch.shape == (B, (R + 1) ** 3)
    for b in range(B):
        one = cs.create_sdf(eng, imgs[b:b + 1], tms[b:b + 1], sps[b:b + 1], R)[0]
        report_close("grid[%d]" % b, batch[b].cpu().numpy(), one.cpu().numpy(), ATOL / 10, RTOL)
    sharded = par.sharded_create_sdf(eng, imgs, tms, sps, R)          # world size 1: no process group
    assert torch.equal(sharded, batch)


def test_bench_contract_on_gpu():
    """bench.py --no-extras prints ONE JSON line with the contract's keys, finite value"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 1e5 and d["dtype"].startswith("f32") and d["vs_baseline"] is None
    assert "workload" in d["config"]


def test_one_rank_rccl_group_runs_the_collective_path(tmp_path):
    """A one-rank 'nccl' (= RCCL) process group on this GPU: bench.py launched by
    torch.distributed.run (init, barrier, MAX all-reduce) and the sharded grid's
    all_gather_into_tensor -- the call sequence of the 8-rank job."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 1e5
    script = tmp_path / "shard.py"
    script.write_text('''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from disn_amd import create_sdf as cs, parallel as par
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
eng = SdfEngine(WeightStore.random_init(0))
imgs = np.random.default_rng(0).random((2, 137, 137, 3), dtype=np.float32)
tm = np.array([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * 2, np.float32)
sp = np.array([[-1, -1, -1, 1, 1, 1]] * 2, np.float64)
a = par.sharded_create_sdf(eng, imgs, tm, sp, 20)
b = cs.create_sdf(eng, imgs, tm, sp, 20)
assert torch.equal(a, b), float((a - b).abs().max())
dist.barrier(); dist.destroy_process_group(); print("SHARD_OK")
''' % root)
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env2)
    assert r.returncode == 0 and "SHARD_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_step_pipeline_equals_single_stream_steps():
    """disn_amd.engine.StepPipeline: three steps in flight (own contexts, streams, host threads, shared weights)
    return, for every job, exactly what one engine returns one step at a time"""
    from disn_amd.engine import SdfEngine, StepPipeline
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(2, mode="he")
    pipe = StepPipeline(store, in_flight=3)
    jobs = []
    for k in range(7):
        d = O.synth_inputs(20 + k, 1, 256 + 64 * k)
        jobs.append((torch.from_numpy(d["imgs"]).cuda(), torch.from_numpy(d["sample_pc"]).cuda(),
                     torch.from_numpy(d["trans_mat"]).cuda()))
    got = pipe.run(jobs)
    torch.cuda.synchronize()
    one = SdfEngine(None, weights=pipe.engines[0].weights)
    for k, job in enumerate(jobs):
        ref = one.encode_query(*job)[1]
        assert torch.equal(got[k], ref), "job %d differs" % k


def test_chained_pipeline_equals_single_stream_steps():
    """StepPipeline.run(chained=True): one host thread, convolution stacks of consecutive steps chained by events
    (disn_ctx_pipeline) -- same results as one engine, one step at a time"""
    from disn_amd.engine import SdfEngine, StepPipeline
    from disn_amd.weights import WeightStore
    pipe = StepPipeline(WeightStore.random_init(4, mode="he"), in_flight=2)
    jobs = []
    for k in range(5):
        d = O.synth_inputs(40 + k, 1, 512)
        jobs.append((torch.from_numpy(d["imgs"]).cuda(), torch.from_numpy(d["sample_pc"]).cuda(),
                     torch.from_numpy(d["trans_mat"]).cuda()))
    got = pipe.run(jobs, chained=True)
    got2 = pipe.run(jobs, chained=True)
    torch.cuda.synchronize()
    one = SdfEngine(None, weights=pipe.engines[0].weights)
    for k, job in enumerate(jobs):
        ref = one.encode_query(*job)[1]
        assert torch.equal(got[k], ref) and torch.equal(got2[k], ref), "job %d differs" % k


@pytest.mark.parametrize("batch,npts,njobs", [(3, 1024, 8), (2, 1000, 3)])
def test_batched_pipeline_equals_single_image_steps(batch, npts, njobs):
    """StepPipeline(batch=B), B < 4: B independent (image, point set) steps per disn_encode_query call -- every image
    keeps its own activation scales (conv stack and point MLPs) and calls of fewer than four images run the
    single-image form of every kernel, so each result is bit for bit the single-image one.  1000 points per image are
    not a multiple of 64: the point-MLP layers then run image by image."""
    from disn_amd.engine import SdfEngine, StepPipeline
    from disn_amd.weights import WeightStore
    pipe = StepPipeline(WeightStore.random_init(6, mode="he"), in_flight=2, batch=batch)
    jobs = _pipeline_jobs(njobs, npts)
    got = pipe.run(jobs)
    torch.cuda.synchronize()
    one = SdfEngine(None, weights=pipe.engines[0].weights)
    for k, job in enumerate(jobs):
        ref = one.encode_query(*job)[1]
        assert torch.equal(got[k], ref), "job %d differs from its single-image run" % k


def test_chained_batched_pipeline_on_separately_allocated_jobs():
    """run(jobs, chained=True) with batch > 1 assembles each call's inputs with torch.cat / an H2D copy;
    those copies must run on the stream of the context that reads them.  Jobs are separately allocated device tensors
    (a real cat) and, for a second run, numpy feeds (a real H2D copy); both runs are repeated so that the allocator
    gets the chance to hand a temporary out again.  Every job must equal the non-chained batched run bit for bit."""
    from disn_amd.engine import StepPipeline
    from disn_amd.weights import WeightStore
    pipe = StepPipeline(WeightStore.random_init(6, mode="he"), in_flight=2, batch=3)
    jobs = _pipeline_jobs(8, 1024)
    ref = pipe.run(jobs)
    torch.cuda.synchronize()
    for rep in range(3):
        got = pipe.run(jobs, chained=True)
        torch.cuda.synchronize()
        for k in range(len(jobs)):
            assert torch.equal(got[k], ref[k]), "device jobs, repeat %d, job %d" % (rep, k)
    host_jobs = [tuple(t.cpu().numpy() for t in job) for job in jobs]
    for rep in range(2):
        got = pipe.run(host_jobs, chained=True)
        torch.cuda.synchronize()
        for k in range(len(jobs)):
            assert torch.equal(got[k], ref[k]), "numpy jobs, repeat %d, job %d" % (rep, k)
    # one shape per call is checked on the chained path too, and a raising call leaves no stale pipeline events behind
    bad = list(jobs)
    bad[1] = (bad[1][0], bad[1][1][:, :512].contiguous(), bad[1][2])
    with pytest.raises(ValueError):
        pipe.run(bad, chained=True)
    got = pipe.run(jobs)
    torch.cuda.synchronize()
    assert all(torch.equal(got[k], ref[k]) for k in range(len(jobs)))


@pytest.mark.parametrize("B,N", [(5, 256), (4, 1000), (16, 2048), (4, 8192)])
def test_strict_mode_runs_the_single_image_forms(B, N):
    """disn_vgg_weights_t.strict_forms = 1 (SdfEngine(strict=True), include/disn_amd.h): a call of >= 4 requests through the
    single-image forms of the convolutions (conv_h2.hip), the fc head (one-row split count / row kernels) and the point-MLP
    layers (dense_h2.hip's four-k-wave tiles).  (1) every request's taps, embedding AND pred_sdf are BIT FOR BIT those of the
    request alone; (2) the flag changes something: the default engine's batched forms give other bits; (3) within 1e-5 of
    the float64 oracle.  N = 1000 is not a multiple of 64: the layers then run image by image; N = 8192 per request takes the
    fused kernels in either mode (per-point scales): still bit for bit the request alone."""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(6, mode="he")
    fast = SdfEngine(store)
    strict = SdfEngine(None, weights=fast.weights, strict=True)
    jobs = _pipeline_jobs(B, N, seed0=80)
    imgs, pts, tms = (torch.cat([j[k] for j in jobs]) for k in range(3))
    enc_s, pred_s = strict.encode_query(imgs, pts, tms)
    enc_f, pred_f = fast.encode_query(imgs, pts, tms)
    torch.cuda.synchronize()
    assert not torch.equal(enc_s.taps[2], enc_f.taps[2]), "strict and default ran the same convolution form"
    worst = 0.0
    for b in (0, B - 1):
        enc1, pred1 = fast.encode_query(*jobs[b])
        for k in range(5):
            assert torch.equal(enc_s.taps[k][b], enc1.taps[k][0]), "tap %d of request %d" % (k, b)
        assert torch.equal(enc_s.embedding[b], enc1.embedding[0]), "embedding of request %d" % b
        worst = max(worst, float((pred_s[b] - pred1[0]).abs().max()))
    print("strict call of %d x %d: max |pred - request alone| %.3g; |strict - default| %.3g" % (
        B, N, worst, float((pred_s - pred_f).abs().max())))
    assert worst == 0.0, "strict call: pred_sdf differs from the request alone by %g" % worst
    d = {"imgs": jobs[0][0].cpu().numpy(), "sample_pc": jobs[0][1].cpu().numpy(), "sample_pc_rot": jobs[0][1].cpu().numpy(),
         "trans_mat": jobs[0][2].cpu().numpy()}
    ref = O.get_model(d, store.arrays, dtype=np.float64)["pred_sdf"][0, :, 0]
    assert float(np.abs(pred_s[0].cpu().numpy() - ref).max()) <= PRED_ATOL


def _pipeline_jobs(njobs, npts, seed0=60):
    jobs = []
    for k in range(njobs):
        d = O.synth_inputs(seed0 + k, 1, npts)
        d["imgs"] *= np.float32(0.25 + 0.25 * (k % 4))    # different brightness (pixels stay in [0, 1]): different activation maxima
        jobs.append((torch.from_numpy(d["imgs"]).cuda(), torch.from_numpy(d["sample_pc"]).cuda(),
                     torch.from_numpy(d["trans_mat"]).cuda()))
    return jobs


def test_batched_calls_are_batch_invariant_and_within_the_bar():
    """bench.py's configuration: eight steps per disn_encode_query call.  Calls of four images and more run the
    BATCHED form of the 28..224-pixel convolutions (conv_h2w.hip: another K summation order than the single-image
    kernels -- include/disn_amd.h, disn_conv3x3_h2).  So: (1) an image's result does not depend on its companions, its
    position in the call or the number of images (>= 4) of the call -- bit for bit; (2) it agrees with the step run
    alone to fp32 rounding; (3) it is within north_star's 1e-5 of the float64 oracle."""
    from disn_amd.engine import SdfEngine, StepPipeline
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(6, mode="he")
    pipe8 = StepPipeline(store, in_flight=2, batch=8)
    jobs = _pipeline_jobs(16, 2048)
    got = pipe8.run(jobs)                                   # calls of 8 + 8
    torch.cuda.synchronize()
    eng = SdfEngine(None, weights=pipe8.engines[0].weights)
    # other companions / positions / call sizes: jobs reversed in calls of 5 + 5 + 6 through one engine
    order = list(range(15, -1, -1))
    got2 = [None] * 16
    for lo, hi in ((0, 5), (5, 10), (10, 16)):
        idx = order[lo:hi]
        sdf = eng.encode_query(torch.cat([jobs[k][0] for k in idx]), torch.cat([jobs[k][1] for k in idx]),
                               torch.cat([jobs[k][2] for k in idx]))[1]
        for i, k in enumerate(idx):
            got2[k] = sdf[i:i + 1].clone()
    torch.cuda.synchronize()
    # ... and one call of all sixteen (bench.py's default call size: sixteen fc rows cross the split-K reduce's
    # parallelism threshold -- the batched fc layers fix the reduce's lane count for every B)
    sdf16 = eng.encode_query(torch.cat([j[0] for j in jobs]), torch.cat([j[1] for j in jobs]),
                             torch.cat([j[2] for j in jobs]))[1]
    torch.cuda.synchronize()
    for k in range(16):
        assert torch.equal(got[k], got2[k]), "job %d: bits depend on the call it travels in" % k
        assert torch.equal(got[k], sdf16[k:k + 1]), "job %d: bits of a 16-request call differ from an 8-request call's" % k
    worst1 = worst64 = 0.0
    for k in (0, 7, 13):
        one = eng.encode_query(*jobs[k])[1]
        worst1 = max(worst1, float((got[k] - one).abs().max()))
        d = {"imgs": jobs[k][0].cpu().numpy(), "sample_pc": jobs[k][1].cpu().numpy(),
             "sample_pc_rot": jobs[k][1].cpu().numpy(), "trans_mat": jobs[k][2].cpu().numpy()}
        ref = O.get_model(d, store.arrays, dtype=np.float64)["pred_sdf"][..., 0]
        worst64 = max(worst64, float(np.abs(got[k].cpu().numpy() - ref).max()))
        print("job %d: |pred| max %.3g, batched vs alone %.3g, batched vs float64 %.3g" % (
            k, float(np.abs(ref).max()), float((got[k] - one).abs().max()), float(np.abs(got[k].cpu().numpy() - ref).max())))
    assert worst1 <= 1e-5 and worst64 <= PRED_ATOL


def test_kernel_selection_thresholds_stay_within_the_bar():
    """ADVICE r2 / include/disn_amd.h (disn_encode_query, WHICH KERNELS RUN): across the N = 8184 | 8192 boundary of a
    single request the point MLPs switch between the layer-by-layer two-term f16 kernels and the fused small-set kernels
    (until round 4: the three-term GEMM chain) -- results of the two sides agree to fp32 rounding, each within 1e-5 of the float64 oracle; the old B = 32 | 33 boundary is gone
    (round 4: calls of >= 4 requests with N % 128 == 0 run the fused small-set kernels whatever B): bit-identical"""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(6, mode="he")
    eng = SdfEngine(store)
    d = O.synth_inputs(77, 1, 8192)
    img, tm = torch.from_numpy(d["imgs"]).cuda(), torch.from_numpy(d["trans_mat"]).cuda()
    pts = torch.from_numpy(d["sample_pc"]).cuda()
    ref = O.get_model(d, store.arrays, dtype=np.float64)["pred_sdf"][0, :, 0]
    big = eng.encode_query(img, pts, tm)[1][0].cpu().numpy()                      # N = 8192: fused kernels
    small = eng.encode_query(img, pts[:, :8184].contiguous(), tm)[1][0].cpu().numpy()   # N = 8184: dense_h2
    print("N = 8192 vs float64 %.3g; N = 8184 vs float64 %.3g; the two forms on the shared points %.3g" % (
        np.abs(big - ref).max(), np.abs(small - ref[:8184]).max(), np.abs(big[:8184] - small).max()))
    assert np.abs(big - ref).max() <= PRED_ATOL and np.abs(small - ref[:8184]).max() <= PRED_ATOL
    # B = 33 against B = 32 (round 3: two forms; now one): image 0 against the oracle, the shared images bit for bit
    d2 = O.synth_inputs(78, 1, 256)
    imgs = torch.from_numpy(np.repeat(d2["imgs"], 33, axis=0) * np.linspace(0.5, 1.0, 33, dtype=np.float32).reshape(33, 1, 1, 1)).cuda()
    p17 = torch.from_numpy(np.repeat(d2["sample_pc"], 33, axis=0)).cuda()
    t17 = torch.from_numpy(np.repeat(d2["trans_mat"], 33, axis=0)).cuda()
    s17 = eng.encode_query(imgs, p17, t17)[1].cpu().numpy()
    s16 = eng.encode_query(imgs[:32].contiguous(), p17[:32].contiguous(), t17[:32].contiguous())[1].cpu().numpy()
    d2["imgs"] = d2["imgs"] * np.float32(0.5)
    r0 = O.get_model(d2, store.arrays, dtype=np.float64)["pred_sdf"][0, :, 0]
    print("B = 33 vs float64 %.3g; B = 32 vs float64 %.3g; 33 vs 32 %.3g" % (
        np.abs(s17[0] - r0).max(), np.abs(s16[0] - r0).max(), np.abs(s17[:32] - s16).max()))
    assert np.abs(s17[0] - r0).max() <= PRED_ATOL and np.abs(s16[0] - r0).max() <= PRED_ATOL
    assert np.array_equal(s17[:32], s16)
