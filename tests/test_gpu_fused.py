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

from conftest import ROOT, report_close
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
    return (O.get_sdf_basic2(pts, enc.embedding.cpu().numpy(), store.arrays, dtype=np.float64)
            + O.get_sdf_basic2_imgfeat_twostream(pts, feat, store.arrays, dtype=np.float64))[..., 0]


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
