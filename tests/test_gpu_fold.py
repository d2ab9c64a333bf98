"""Folded local stream (disn_fold_local, disn_query_folded, disn_query_grid_folded): the local
fold2/conv1 pre-multiplied into the feature map.  Same math re-associated, so the bar is fp32 rounding
against the unfolded HIP path and the usual 1e-5 against the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import internal_arrays, report_close
from oracle import disn_oracle as O

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-5


@pytest.fixture(scope="module")
def eng_store():
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(2, mode="he")
    return SdfEngine(store), store


def test_fold_local_vs_float64(eng_store):
    eng, store = eng_store
    rng = np.random.default_rng(1)
    enc = eng.encode(rng.random((2, 137, 137, 3), dtype=np.float32))
    # (the engine's feature map AND its fold2/conv1 -- rows and columns -- are in equalised units: the oracle product is
    # formed from the same copy of the variables)
    w = internal_arrays(store)["sdfprediction_imgfeat/fold2/conv1/weights"][0, 0].astype(np.float64)
    for b in range(2):
        pm = eng.pmap_of(enc, b).cpu().numpy()
        ref = enc.featmap[b].reshape(-1, 1472).cpu().numpy().astype(np.float64) @ w[512:]
        assert pm.shape == (137 * 137, 512)
        err = np.abs(pm - ref).max() / np.abs(ref).max()
        print("fold_local image %d: max err / scale %.3g" % (b, err))
        assert err < 4e-6


@pytest.mark.parametrize("B,N", [(1, 70001), (2, 33000), (3, 257)])
def test_query_folded_vs_unfolded_and_oracle(eng_store, B, N):
    eng, store = eng_store
    rng = np.random.default_rng(B * 7 + N)
    imgs = rng.random((B, 137, 137, 3), dtype=np.float32)
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    tms = np.stack([O.DEMO_TRANS_MAT[0], O.synth_trans_mat(30, 25, 0.8), O.synth_trans_mat(201.5, 30, 0.65)])[:B]
    enc = eng.encode(imgs)
    a = eng.query(enc, pts, tms, fold=False)
    f = eng.query(enc, pts, tms, fold=True, fused=False)
    torch.cuda.synchronize()
    scale = float(a.abs().max())
    d = float((a - f).abs().max())
    print("folded vs unfolded: max |d| %.3g, scale %.3g" % (d, scale))
    assert d <= 4e-6 * max(scale, 1.0)
    # a strided sample against the float64 oracle MLP on the GPU's own features
    idx = np.arange(0, N, max(1, N // 300))
    sub = pts[:, idx]
    xy = O.get_img_points(sub, tms)
    feat = O.resampler(eng.true_features(enc.featmap).cpu().numpy(), xy)[:, :, None, :]
    ref = (O.get_sdf_basic2(sub, enc.embedding.cpu().numpy(), store.arrays, dtype=np.float64)
           + O.get_sdf_basic2_imgfeat_twostream(sub, feat, store.arrays, dtype=np.float64))[..., 0]
    report_close("folded vs oracle", f[:, torch.from_numpy(idx).cuda()].cpu().numpy(), ref, ATOL, RTOL)


def test_grid_folded_vs_unfolded(eng_store):
    eng, _ = eng_store
    enc = eng.encode(O.synth_inputs(3, 1, 8)["imgs"])
    R, sp = 44, [-1, -0.9, -0.8, 1, 0.9, 0.8]          # 45^3 = 91125 points: one full chunk + a ragged one
    total = (R + 1) ** 3
    a = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, fold=False)
    f = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, fused=False)            # folded, layer-by-layer GEMMs
    part = eng.query_grid(enc, 0, O.DEMO_TRANS_MAT, sp, R, 70000, total, fused=False)
    torch.cuda.synchronize()
    assert f.shape == (total,)
    d = float((a - f).abs().max())
    print("grid folded vs unfolded: max |d| %.3g (values are pred/10)" % d)
    assert d <= 5e-7 * max(1.0, float(a.abs().max()) * 10)
    report_close("slice of the folded grid", part.cpu().numpy(), f[70000:].cpu().numpy(), 2e-6, 1e-6)


def test_folded_entry_points_need_the_split_weights(eng_store):
    import ctypes as C
    from disn_amd import _lib, ops
    eng, _ = eng_store
    w = _lib.MlpWeights()
    C.memmove(C.byref(w), C.byref(eng.weights.mlp), C.sizeof(w))
    w.l_w4_point = None
    fm = torch.zeros((137, 137, 1472), device="cuda")
    with pytest.raises(_lib.DisnError) as e:
        ops.fold_local(w, fm)
    assert e.value.status == -1
