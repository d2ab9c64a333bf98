"""CPU check of the fused point-MLP kernel's layout contract (disn_amd/csrc/mlp_fused.hip) through the
lane-level emulation in tests/fused_emulation.py: packed stream order, slot <-> feature permutation,
C-layout chaining, power-of-two scaling and the two-term fp16 split reproduce the plain MLP."""
import numpy as np
import pytest

from tests import fused_emulation as E


def _mlp_ref(w, consts, pts, add4, dtype=np.float64):
    c = {k: np.asarray(v, dtype) for k, v in consts.items() if k != "add4max"}
    h = np.maximum(pts.astype(dtype) @ c["w1"] + c["b1"], 0)
    h = np.maximum(h @ w[0].astype(dtype) + c["b2"], 0)
    h = np.maximum(h @ w[1].astype(dtype) + c["b3"], 0)
    z = h @ w[2].astype(dtype) + c["b4"]
    if add4 is not None:
        z = z + add4.astype(dtype)
    h = np.maximum(z, 0)
    h = np.maximum(h @ w[3].astype(dtype) + c["b5"], 0)
    return h @ c["w6"] + c["b6"]


def _make(seed, scale=1.0, with_add=True):
    rng = np.random.default_rng(seed)
    w = [(rng.standard_normal((k, n)) * np.sqrt(2.0 / k) * scale).astype(np.float32) for k, n in E.LAYER_DIMS]
    consts = {"w1": (rng.standard_normal((3, 64)) * 0.8).astype(np.float32),
              "b1": (rng.standard_normal(64) * 0.1).astype(np.float32),
              "b2": (rng.standard_normal(256) * 0.1).astype(np.float32),
              "b3": (rng.standard_normal(512) * 0.1).astype(np.float32),
              "b4": (rng.standard_normal(512) * 0.1).astype(np.float32),
              "b5": (rng.standard_normal(256) * 0.1).astype(np.float32),
              "w6": (rng.standard_normal(256) * np.sqrt(1.0 / 256)).astype(np.float32),
              "b6": np.float32(0.05)}
    pts = (rng.random((32, 3)) * 2 - 1).astype(np.float32)
    add4 = (rng.standard_normal((32, 512)) * 0.7).astype(np.float32) if with_add else None
    consts["add4max"] = float(np.abs(add4).max()) if with_add else 0.0
    return w, consts, pts, add4


def test_pair_coords_cover_every_block_once():
    seen = set()
    for p in range(E.PAIRS):
        seen.add(E.pair_coords(p))
    assert len(seen) == E.PAIRS
    for layer, (K, N) in enumerate(E.LAYER_DIMS):
        assert sum(1 for c in seen if c[0] == layer) == (K // 16) * (N // 32)


def test_phi_is_a_permutation_of_each_block():
    g, t = np.meshgrid(np.arange(2), np.arange(8), indexing="ij")
    for kb in (0, 3, 31):
        assert sorted(E.phi(kb, g, t).ravel().tolist()) == list(range(16 * kb, 16 * kb + 16))


def test_split_is_23_bit():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(100000) * 3000).astype(np.float32)
    h, l = E.split16(v)
    err = np.abs(v.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
    big = np.abs(v) > 0.125            # below 2^-3 the low term is subnormal: absolute error <= 2^-25
    assert (err[big] <= np.abs(v[big]) * 2.0 ** -23).all()
    assert (err[~big] <= 2.0 ** -25).all()


@pytest.mark.parametrize("seed,scale,with_add", [(0, 1.0, True), (1, 1.0, False), (2, 0.05, True), (3, 30.0, True)])
def test_emulated_kernel_matches_the_mlp(seed, scale, with_add):
    w, consts, pts, add4 = _make(seed, scale, with_add)
    img, meta = E.pack_image(*w)
    got = E.fused_stream(img, meta, consts, pts, add4)
    ref = _mlp_ref(w, consts, pts, add4)
    ref32 = _mlp_ref(w, consts, pts, add4, np.float32)
    sc = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max()) / sc
    err32 = float(np.abs(ref32 - ref).max()) / sc
    # the two-term path is as close to float64 as a plain float32 evaluation is (same order of magnitude)
    assert err <= 3e-6, (err, err32)
    assert err <= 4 * err32 + 2e-7, (err, err32)


def test_feat_pair_coords_cover_every_block_once():
    seen = {E.pair_coords(p, True) for p in range(E.PAIRS_FEAT)}
    assert len(seen) == E.PAIRS_FEAT
    assert sum(1 for c in seen if c[0] == 2) == ((512 + E.FEAT_COLS) // 16) * 16
    assert sum(1 for c in seen if c[0] == 2 and c[2] >= 32) == 96 * 16
    # every later pair keeps its position mod 48: fragment register set (mod 3) and ring-slot parity (mod 16)
    assert (16 * E.PAIRS_A2) % 48 == 0 and E.PAIRS_FEAT % 8 == 0 and (E.PAIRS_FEAT // 8) % 2 == 0


@pytest.mark.parametrize("seed,fscale", [(0, 1.0), (1, 300.0), (2, 1e-3)])
def test_emulated_feat_form_matches_the_mlp(seed, fscale):
    """the FEAT form (the gathered features as 96 extra reduction blocks of fold2/conv1, split form from memory, one
    rescale s_feat / s3 of the accumulators) against the plain MLP on [point 512 | feature 1472]"""
    w, consts, pts, _ = _make(seed, 1.0, False)
    rng = np.random.default_rng(100 + seed)
    w4f = (rng.standard_normal((E.FEAT_REAL, 512)) * np.sqrt(2.0 / 1984) / fscale).astype(np.float32)
    w4 = np.concatenate([w[2] * np.float32(np.sqrt(512 / 1984.0)), w4f])
    feat = (np.abs(rng.standard_normal((32, E.FEAT_REAL))) * fscale).astype(np.float32)
    feat[:, rng.integers(0, E.FEAT_REAL, 15)] *= 40.0                 # outlier channels
    featmax = float(np.abs(feat).max()) * 1.7                          # the taps' maximum: a bound, not the exact maximum
    img, meta = E.pack_image(w[0], w[1], w4, w[3])
    assert img.shape[0] == E.PAIRS_FEAT
    got = E.fused_stream(img, meta, consts, pts, None, E.split_rows(feat, featmax), featmax)
    add4 = feat.astype(np.float64) @ w4f.astype(np.float64)
    wref = [w[0], w[1], w4[:512], w[3]]
    ref = _mlp_ref(wref, consts, pts, add4)
    ref32 = _mlp_ref(wref, consts, pts, (feat @ w4f).astype(np.float32), np.float32)
    sc = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max()) / sc
    err32 = float(np.abs(ref32 - ref).max()) / sc
    assert err <= 3e-6, (err, err32)
    assert err <= 4 * err32 + 2e-7, (err, err32)


def test_equalised_image_survives_trained_like_channel_gains():
    """the streams on the oracle's trained-like weights (log-normal channel gains, 1 % outliers x 1000): with raw
    weights the column 1-norms behind the activation-scale bounds are 10^4 above typical and compound (max error 12 on
    |pred| 100); the equalised image keeps the kernel at fp32 level"""
    from oracle import disn_oracle as O
    W = O.trained_like_weights(11)
    rng = np.random.default_rng(0)
    pts = (rng.random((32, 3)) * 2 - 1).astype(np.float32)
    for scope in ("sdfprediction", "sdfprediction_imgfeat"):
        g = lambda l: W["%s/%s/weights" % (scope, l)][0, 0]
        b = lambda l: W["%s/%s/biases" % (scope, l)]
        ws = (g("fold1/conv2"), g("fold1/conv3"), g("fold2/conv1")[:512], g("fold2/conv2"))
        consts = {"w1": g("fold1/conv1"), "b1": b("fold1/conv1"), "b2": b("fold1/conv2"), "b3": b("fold1/conv3"),
                  "b4": b("fold2/conv1"), "b5": b("fold2/conv2"), "w6": g("fold2/conv5").reshape(-1),
                  "b6": np.float32(b("fold2/conv5")[0])}
        img, meta = E.pack_image(*ws)
        assert meta[8] < 4096 and meta[9] < 4096, (meta[8], meta[9])      # normalised columns: 1-norm <= 2 K
        got = E.fused_stream(img, meta, consts, pts)
        ref = _mlp_ref(ws, consts, pts, None)
        ref32 = _mlp_ref(ws, consts, pts, None, np.float32)
        sc = max(1.0, float(np.abs(ref).max()))
        err, err32 = float(np.abs(got - ref).max()) / sc, float(np.abs(ref32 - ref).max()) / sc
        assert err <= 3e-6 and err <= 4 * err32 + 2e-7, (scope, err, err32)
