"""Fused point-MLP kernels (disn_amd/csrc/mlp_fused.hip; disn_mlp_fused_pack, disn_query_fused,
disn_query_grid_fused): both MLP streams with every activation in registers, fp32-accurate products from
a two-term fp16 split.  Bars: the device pack equals the CPU restatement of the layout bit for bit; the
prediction is within 1e-5 of the float64 oracle (the north-star bar) and as close to it as the
layer-by-layer three-term path is; grid mode == point mode bit for bit; runs are bit-reproducible; the
counted-wait build equals the wait-for-everything build bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, internal_arrays, report_close
from oracle import disn_oracle as O
from tests import fused_emulation as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_store():
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(2, mode="he")
    return SdfEngine(store), store


def _raw(store, scope):
    w = lambda l: store["%s/%s/weights" % (scope, l)][0, 0]
    return w("fold1/conv2"), w("fold1/conv3"), w("fold2/conv1")[:512], w("fold2/conv2")


@pytest.mark.parametrize("scope", ["sdfprediction", "sdfprediction_imgfeat"])
def test_device_pack_equals_the_layout_restatement(eng_store, scope):
    from disn_amd import ops
    _, store = eng_store
    ws = _raw(store, scope)
    img = ops.mlp_fused_pack(*[torch.from_numpy(np.ascontiguousarray(w, np.float32)).cuda() for w in ws])
    torch.cuda.synchronize()
    raw = img.cpu().numpy()
    ref_img, ref_meta = E.pack_image(*ws)
    nb = E.PAIRS * 2048
    got_img = raw[:nb].view(np.float16).reshape(E.PAIRS, 2, 64, 8)
    got_meta = raw[nb:nb + 4 * E.META].view(np.float32)
    assert np.array_equal(got_meta[64:], ref_meta[64:])                       # inverse weight scales per output feature
    np.testing.assert_allclose(got_meta[8:10], ref_meta[8:10], rtol=1e-5)     # column 1-norms: summation order
    assert np.array_equal(got_img.view(np.uint16), ref_img.view(np.uint16))


def _oracle(store, enc, pts, tms):
    xy = O.get_img_points(pts, tms)
    feat = O.resampler(enc.featmap.cpu().numpy(), xy)[:, :, None, :]
    W = internal_arrays(store)     # the engine's own feature map is in ITS units (the equalised copy of the variables)
    return (O.get_sdf_basic2(pts, enc.embedding.cpu().numpy(), W, dtype=np.float64)
            + O.get_sdf_basic2_imgfeat_twostream(pts, feat, W, dtype=np.float64))[..., 0]


@pytest.mark.parametrize("B,N", [(1, 1), (1, 127), (1, 4097), (2, 1000), (1, 70001)])
def test_fused_query_vs_float64_oracle_and_unfused(eng_store, B, N):
    eng, store = eng_store
    rng = np.random.default_rng(B * 11 + N)
    imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    tms = np.stack([O.DEMO_TRANS_MAT[0], O.synth_trans_mat(30, 25, 0.8)])[:B]
    enc = eng.encode(imgs)
    eng.featmap_of(enc)
    f = eng.query(enc, pts, tms, fold=True, fused=True)
    u = eng.query(enc, pts, tms, fold=True, fused=False)
    f2 = eng.query(enc, pts, tms, fold=True, fused=True)
    torch.cuda.synchronize()
    assert torch.equal(f, f2), "fused kernel is not bit-reproducible"
    idx = np.unique(np.concatenate([np.arange(0, N, max(1, N // 400)), [N - 1]]))
    ref = _oracle(store, enc, pts[:, idx], tms)
    gi = torch.from_numpy(idx).cuda()
    ef = report_close("fused vs float64", f[:, gi].cpu().numpy(), ref, 1e-5)
    eu = report_close("unfused folded vs float64", u[:, gi].cpu().numpy(), ref, 1e-5)
    d = float((f - u).abs().max())
    print("N=%d: |fused - f64| %.3g, |unfused - f64| %.3g, |fused - unfused| %.3g (scale %.3g)" % (
        N, ef, eu, d, float(np.abs(ref).max())))
    assert ef <= 2.0 * eu + 2e-6      # as close to the truth as the three-term path
    assert d <= 6e-6 * max(1.0, float(np.abs(ref).max()))


def test_fused_small_activations_keep_relative_accuracy():
    """xavier weights shrink the activations to << 1: the per-point scales must keep the
    RELATIVE error at fp32 level, not just the absolute one"""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(5, mode="xavier")
    eng = SdfEngine(store)
    rng = np.random.default_rng(3)
    imgs = rng.random((1, 137, 137, 3), dtype=np.float32)
    pts = rng.uniform(-1, 1, (1, 3000, 3)).astype(np.float32)
    enc = eng.encode(imgs)
    eng.featmap_of(enc)
    f = eng.query(enc, pts, O.DEMO_TRANS_MAT, fold=True, fused=True).cpu().numpy()
    ref = _oracle(store, enc, pts, O.DEMO_TRANS_MAT)
    sc = float(np.abs(ref).max())
    err = float(np.abs(f - ref).max())
    print("xavier: scale %.3g, max |fused - f64| %.3g (%.3g relative)" % (sc, err, err / sc))
    assert err <= 3e-6 * sc


def test_fused_grid_equals_points_and_slices(eng_store):
    from disn_amd import ops
    eng, _ = eng_store
    enc = eng.encode(O.synth_inputs(3, 1, 8)["imgs"])
    R, sp = 44, [-1, -0.9, -0.8, 1, 0.9, 0.8]          # 45^3 = 91125 points
    total = (R + 1) ** 3
    g = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, fused=True)
    part = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, 70001, total, fused=True)
    u = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, fused=False)
    pts = ops.grid_points(sp, R, 0, total, "cuda")
    # true IEEE division as the kernel's `/ out_div` does (torch's `tensor / python_scalar` multiplies by 1/10 on the GPU)
    q = torch.div(eng.query(enc, pts[None], O.DEMO_TRANS_MAT, fold=True, fused=True)[0],
                  torch.tensor(10.0, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(part, g[70001:]), "a slice of the grid differs from the whole grid"
    assert torch.equal(q, g), "grid mode differs from point mode on the same points"
    d = float((g - u).abs().max())
    print("fused grid vs layer-by-layer grid: max |d| %.3g (values are pred/10)" % d)
    assert d <= 6e-7 * max(1.0, float(u.abs().max()) * 10)


def test_fused_entry_points_need_the_images(eng_store):
    import ctypes as C
    from disn_amd import _lib, ops
    eng, _ = eng_store
    w = _lib.MlpWeights()
    C.memmove(C.byref(w), C.byref(eng.weights.mlp), C.sizeof(w))
    w.l_fused = None
    z = torch.zeros((137 * 137, 512), device="cuda")
    with pytest.raises(_lib.DisnError) as e:
        ops.query_fused(w, z[None], torch.ones(1, device="cuda"), torch.zeros((1, 1024), device="cuda"),
                        torch.from_numpy(O.DEMO_TRANS_MAT).cuda(), torch.zeros((1, 8, 3), device="cuda"))
    assert e.value.status == -1


def test_counted_waits_equal_wait_for_everything():
    """the same launches in a tuning build with every LDS-DMA wait replaced by vmcnt(0): any under-counted
    wait in the product kernel shows as a difference (tools/fused_check.py runs both modes in one process)"""
    lib = os.path.join(ROOT, "disn_amd", "csrc", "libdisn_amd_tuning.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(
            os.path.join(ROOT, "disn_amd", "csrc", "libdisn_amd.so")) - 3600:
        pytest.skip("no (or a stale) tuning build: python -m disn_amd.csrc.build --tuning")
    env = dict(os.environ, DISN_AMD_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_check.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "FUSED_CHECK_OK" in r.stdout


# ---------------------------------------------------------------------------------------------------------------
# round 4: the FEAT form -- small point sets straight from the taps (split-form gather + one launch per stream)
# ---------------------------------------------------------------------------------------------------------------
def test_device_feat_pack_equals_the_layout_restatement(eng_store):
    from disn_amd import ops
    _, store = eng_store
    scope = "sdfprediction_imgfeat"
    w = lambda l: np.ascontiguousarray(store["%s/%s/weights" % (scope, l)][0, 0], np.float32)
    ws = (w("fold1/conv2"), w("fold1/conv3"), w("fold2/conv1"), w("fold2/conv2"))
    img = ops.mlp_fused_feat_pack(*[torch.from_numpy(a).cuda() for a in ws])
    torch.cuda.synchronize()
    raw = img.cpu().numpy()
    ref_img, ref_meta = E.pack_image(*ws)
    nb = E.PAIRS_FEAT * 2048
    got_meta = raw[nb:nb + 4 * E.META].view(np.float32)
    assert np.array_equal(got_meta[64:], ref_meta[64:])
    np.testing.assert_allclose(got_meta[8:11], ref_meta[8:11], rtol=1e-5)
    assert got_meta[11] == ref_meta[11]
    assert np.array_equal(raw[:nb].view(np.uint16).reshape(E.PAIRS_FEAT, 2, 64, 8), ref_img.view(np.uint16))


def test_split_form_gather_is_the_fp32_gather_split(eng_store):
    """disn_gather_taps_split = disn_gather_taps, then x * 2^k, h = f16(x), l = f16(x - h), [h8 | l8] per 8 channels"""
    from disn_amd import ops
    eng, _ = eng_store
    rng = np.random.default_rng(5)
    enc = eng.encode(rng.random((2, 137, 137, 3), dtype=np.float32))
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (2, 333, 3)).astype(np.float32)).cuda()
    tms = torch.from_numpy(np.stack([O.DEMO_TRANS_MAT[0], O.synth_trans_mat(30, 25, 0.8)])).cuda()
    feat = ops.gather_taps(enc.taps, tms, pts).cpu().numpy()
    amax = torch.stack([torch.stack([t[b].abs().max() for t in enc.taps]).max() for b in range(2)]).contiguous()
    got = ops.gather_taps_split(enc.taps, tms, pts, amax).cpu().numpy()
    for b in range(2):
        ref = E.split_rows(feat[b], float(amax[b]))
        assert np.array_equal(got[b], ref), "image %d" % b
    # a bound above the true maximum only moves the power of two
    got2 = ops.gather_taps_split(enc.taps, tms, pts, (amax * 3.0).contiguous()).cpu().numpy()
    assert np.array_equal(got2[1], E.split_rows(feat[1], float(amax[1]) * 3.0))


def test_one_wave_per_point_gather_equals_the_thread_per_float4_gather(eng_store):
    """from 10 240 points per call on the gather from the taps runs one wave per point (round 6: wave-uniform geometry,
    duplicate tap rows / columns skipped by scalar branches) -- the SAME BITS as the thread-per-float4 kernel that serves
    smaller calls: 6 images x 2048 points in one call (wave kernel) against the same requests image by image (thread
    kernel), fp32 rows and split rows; points far outside the image, on its border and NaN cameras included"""
    from disn_amd import ops
    eng, _ = eng_store
    rng = np.random.default_rng(11)
    B, N = 6, 2048
    enc = eng.encode(rng.random((B, 137, 137, 3), dtype=np.float32))
    pts = rng.uniform(-0.6, 0.6, (B, N, 3)).astype(np.float32)
    pts[:, :16] *= 40.0                                   # far outside the image
    tms = np.stack([O.DEMO_TRANS_MAT[0]] + [O.synth_trans_mat(30.0 + 50.0 * k, 25.0, 0.8) for k in range(B - 1)]).astype(np.float32)
    tms[3, 3, 2] = np.nan                                 # a degenerate camera: every point of image 3 projects to NaN
    pts_d, tms_d = torch.from_numpy(pts).cuda(), torch.from_numpy(tms).cuda()
    amax = torch.stack([torch.stack([t[b].abs().max() for t in enc.taps]).max() for b in range(B)]).contiguous()
    wave = ops.gather_taps(enc.taps, tms_d, pts_d)
    wave_split = ops.gather_taps_split(enc.taps, tms_d, pts_d, amax)
    assert bool(torch.isfinite(wave).all())
    for b in range(B):
        taps_b = [t[b:b + 1].contiguous() for t in enc.taps]
        one = ops.gather_taps(taps_b, tms_d[b:b + 1].contiguous(), pts_d[b:b + 1].contiguous())
        assert torch.equal(one[0], wave[b]), "fp32 rows, image %d" % b
        one_s = ops.gather_taps_split(taps_b, tms_d[b:b + 1].contiguous(), pts_d[b:b + 1].contiguous(), amax[b:b + 1].contiguous())
        assert torch.equal(one_s[0], wave_split[b]), "split rows, image %d" % b
    assert float(wave[3].abs().max()) == 0.0              # NaN projection -> the resampler's zeros (sample4's `ok` test)


def _oracle_taps(store, enc, pts, tms):
    fm = np.concatenate([O.resize_bilinear_legacy(t.cpu().numpy(), 137, 137) for t in enc.taps], axis=3)
    xy = O.get_img_points(pts, tms)
    feat = O.resampler(fm, xy)[:, :, None, :]
    W = internal_arrays(store)     # the engine's own taps are in ITS units (the equalised copy of the variables)
    return (O.get_sdf_basic2(pts, enc.embedding.cpu().numpy(), W, dtype=np.float64)
            + O.get_sdf_basic2_imgfeat_twostream(pts, feat, W, dtype=np.float64))[..., 0]


@pytest.mark.parametrize("B,N", [(1, 128), (2, 256), (3, 1152), (4, 2048), (5, 640), (16, 2048)])
def test_query_taps_fused_vs_float64_oracle(eng_store, B, N):
    from disn_amd import ops
    eng, store = eng_store
    rng = np.random.default_rng(B * 13 + N)
    imgs = rng.random((B, 137, 137, 3), dtype=np.float32) * rng.uniform(0.3, 1.0, (B, 1, 1, 1)).astype(np.float32)
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    tms = np.stack([O.DEMO_TRANS_MAT[0] if b % 2 == 0 else O.synth_trans_mat(30 + 20 * b, 25, 0.8) for b in range(B)])
    enc = eng.encode(imgs)
    dp, dt = torch.from_numpy(pts).cuda(), torch.from_numpy(tms).cuda()
    f = ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, dt, dp)
    f2 = ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, dt, dp)
    torch.cuda.synchronize()
    assert torch.equal(f, f2), "the fused small-set kernels are not bit-reproducible"
    nb = min(B, 3)                                     # the oracle on the first images (137 x 137 x 1472 maps on the CPU)
    sub = type(enc)(enc.resized[:nb], [t[:nb] for t in enc.taps], enc.embedding[:nb], None)
    ref = _oracle_taps(store, sub, pts[:nb], tms[:nb])
    ef = report_close("fused small-set vs float64", f[:nb].cpu().numpy(), ref, 1e-5)
    u = eng.query(enc, dp, dt, fold=False, fused=False)            # the layer-by-layer path on the materialised map
    d = float((f - u).abs().max())
    print("B=%d N=%d: |fused small - f64| %.3g, |fused small - layer by layer| %.3g (|pred| max %.3g)" % (
        B, N, ef, d, float(np.abs(ref).max())))
    assert d <= 1e-5
    # an image's bits do not depend on its companions or its position in the call -- within one FORM of the fc head:
    # calls of < 4 images fold the global bias with the row kernels, calls of >= 4 with the split-K stream kernel
    # (include/disn_amd.h, "WHICH KERNELS RUN"), so a single image is compared with a call of < 4 and a permuted /
    # shortened batch with a call of >= 4
    if 2 <= B < 4:
        one = ops.query_taps_fused(eng.weights.mlp, [t[B - 1:B].contiguous() for t in enc.taps],
                                   enc.embedding[B - 1:B].contiguous(), dt[B - 1:B].contiguous(), dp[B - 1:B].contiguous())
        assert torch.equal(one[0], f[B - 1])
    if B >= 5:
        perm = torch.tensor(list(range(B - 1, 0, -1)), device="cuda")          # reversed, image 0 dropped
        g = ops.query_taps_fused(eng.weights.mlp, [t[perm].contiguous() for t in enc.taps], enc.embedding[perm].contiguous(),
                                 dt[perm].contiguous(), dp[perm].contiguous())
        assert torch.equal(g, f[perm])


@pytest.mark.parametrize("B,N", [(4, 2048), (6, 640)])
def test_batched_encode_query_runs_the_fused_small_set_kernels(eng_store, B, N):
    """disn_encode_query of >= 4 images (N % 128 == 0) = disn_encode + disn_query_taps_fused, bit for bit: the split
    scale comes from the convolutions' epilogue maxima there and from a pass over the taps here -- the same numbers"""
    from disn_amd import ops
    eng, _ = eng_store
    rng = np.random.default_rng(B + N)
    imgs = torch.from_numpy(rng.random((B, 137, 137, 3), dtype=np.float32)).cuda()
    pts = torch.from_numpy(rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)).cuda()
    tms = torch.from_numpy(np.repeat(O.DEMO_TRANS_MAT, B, axis=0)).cuda()
    enc, pred = eng.encode_query(imgs, pts, tms)
    enc2 = eng.encode(imgs)
    for a, b in zip(enc.taps, enc2.taps):
        assert torch.equal(a, b)
    q = ops.query_taps_fused(eng.weights.mlp, enc2.taps, enc2.embedding, tms, pts)
    assert torch.equal(pred, q), float((pred - q).abs().max())


@pytest.mark.parametrize("B,N", [(5, 1000), (4, 1), (1, 9000)])
def test_ragged_point_sets_are_padded_inside_the_library(eng_store, B, N):
    """a batched call (or a single request of >= 8192 points) whose N is not a multiple of 128 runs
    the fused small-set kernels all the same -- the library pads every point set with (0, 0, 0) (test/create_sdf.py:241,256)
    and drops those results.  Bit for bit the caller-padded call's first N results, through disn_encode_query and
    through disn_query_taps_fused; within the bar of the float64 oracle."""
    from disn_amd import ops
    eng, store = eng_store
    rng = np.random.default_rng(B * 17 + N)
    Np = (N + 127) // 128 * 128
    imgs = torch.from_numpy(rng.random((B, 137, 137, 3), dtype=np.float32)).cuda()
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    padded = np.zeros((B, Np, 3), np.float32)
    padded[:, :N] = pts
    tms = np.stack([O.DEMO_TRANS_MAT[0] if b % 2 == 0 else O.synth_trans_mat(40 + 25 * b, 20, 0.8) for b in range(B)])
    dp, dpp, dt = torch.from_numpy(pts).cuda(), torch.from_numpy(padded).cuda(), torch.from_numpy(tms).cuda()
    enc, a = eng.encode_query(imgs, dp, dt)
    _, b = eng.encode_query(imgs, dpp, dt)
    assert a.shape == (B, N)
    assert torch.equal(a, b[:, :N]), float((a - b[:, :N]).abs().max())
    q = ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, dt, dp)
    assert torch.equal(q, a)
    rot = torch.from_numpy((pts * np.float32(0.5)).astype(np.float32)).cuda()         # pts_rot != pts: its own pad buffer
    padrot = torch.zeros((B, Np, 3), device="cuda")
    padrot[:, :N] = rot
    r1 = eng.encode_query(imgs, dp, dt, pts_rot=rot)[1]
    r2 = eng.encode_query(imgs, dpp, dt, pts_rot=padrot)[1]
    assert torch.equal(r1, r2[:, :N])
    d = {"imgs": imgs[:1].cpu().numpy(), "sample_pc": pts[:1, :256], "sample_pc_rot": pts[:1, :256], "trans_mat": tms[:1]}
    ref = O.get_model(d, store.arrays, dtype=np.float64)["pred_sdf"][0, :, 0]
    report_close("padded call vs float64", a[0, :256].cpu().numpy(), ref[:min(N, 256)], 1e-5)


def test_batched_call_with_a_degenerate_camera(eng_store):
    """nulls on the fused small-set path: request 2 of a five-request call has an all-zero camera (0 / 0 -> NaN image
    coordinates -> the resampler's zeros for every feature, models/model_normalization.py:172-190 through
    tf.contrib.resampler's range test) and request 3 projects every point outside the image (clamped to the border as the
    reference's tf.minimum / tf.maximum do).  Each request against the float64 oracle; the OTHER requests bit for bit
    what they are in a call without the degenerate ones"""
    eng, store = eng_store
    B, N = 5, 256
    rng = np.random.default_rng(99)
    imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    tms = np.repeat(O.DEMO_TRANS_MAT, B, axis=0).copy()
    tms_ok = tms.copy()
    tms[2] = 0.0
    tms[3, 3, :2] += 4000.0                            # far outside: coordinates clamp to 136
    pred = eng.encode_query(torch.from_numpy(imgs).cuda(), torch.from_numpy(pts).cuda(), torch.from_numpy(tms).cuda())[1]
    base = eng.encode_query(torch.from_numpy(imgs).cuda(), torch.from_numpy(pts).cuda(), torch.from_numpy(tms_ok).cuda())[1]
    assert bool(torch.isfinite(pred).all())
    for b in (0, 1, 4):
        assert torch.equal(pred[b], base[b]), "request %d depends on its companions' cameras" % b
    worst = 0.0
    for b in range(B):
        d = {"imgs": imgs[b:b + 1], "sample_pc": pts[b:b + 1], "sample_pc_rot": pts[b:b + 1], "trans_mat": tms[b:b + 1]}
        ref = O.get_model(d, store.arrays, dtype=np.float64)["pred_sdf"][0, :, 0]
        err = float(np.abs(pred[b].cpu().numpy() - ref).max())
        worst = max(worst, err)
        print("request %d: |pred| max %.3g, |gpu - f64| %.3g" % (b, float(np.abs(ref).max()), err))
    assert worst <= 1e-5
