"""GPU parity of the small-batch point-MLP layer (disn_amd/csrc/dense_h2.hip, through the C ABI) against float64:
the layer shapes of models/sdfnet.py:71-88,173-186 at a 2048-point batch, ragged row counts, the two-source
(tf.concat read in place) form, the deferred bias + ReLU on load, run-to-run reproducibility."""
import numpy as np
import pytest
import torch

from conftest import report_close

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from disn_amd import ops as _ops
    return _ops


def case(M, K, N, seed, positive=True):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((M, K)).astype(np.float32)
    if positive:
        a = np.maximum(a, 0) * 1.5
    w = (rng.standard_normal((K, N)) * np.sqrt(2.0 / K)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    return a, w, b


@pytest.mark.parametrize("M,K,N,relu", [(2048, 64, 256, True), (2048, 256, 512, True), (2048, 512, 512, False),
                                        (2048, 512, 256, True), (777, 256, 512, True), (33, 512, 256, True),
                                        (5000, 64, 256, True), (1, 512, 512, False), (4096, 1024, 512, True),
                                        (96, 128, 2048, True), (64, 4096, 4096, True)])   # > 1024 columns / rows: ADVICE r4
def test_layer_shapes_vs_float64(ops, M, K, N, relu):
    a, w, b = case(M, K, N, M + K + N)
    ref = a.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        ref = np.maximum(ref, 0)
    img = ops.pack_dense_h2(dev(w))
    out, amax = ops.dense_h2(dev(a), img, dev(b), N, relu, want_amax=True)
    got = host(out)
    report_close("dense_h2 %s" % ((M, K, N),), got, ref, atol=1e-5, rtol=1e-5)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    print("M %d K %d N %d: max err %.3g of scale %.3g (%.3g relative)" % (M, K, N, err, scale, err / scale))
    assert err <= 2e-6 * scale
    assert float(amax) == float(np.abs(got).max())
    assert np.array_equal(got, host(ops.dense_h2(dev(a), img, dev(b), N, relu)))


@pytest.mark.parametrize("M,k1,k2,N", [(2048, 512, 1472, 512), (700, 512, 1472, 512), (2048, 256, 256, 512),
                                       (100, 64, 128, 256)])
def test_two_sources_read_in_place(ops, M, k1, k2, N):
    """[a1 | a2] . W without materialising the concat (models/sdfnet.py:180): the 1984-deep local fold2/conv1"""
    a, w, b = case(M, k1 + k2, N, M + k1)
    a1, a2 = np.ascontiguousarray(a[:, :k1]), np.ascontiguousarray(a[:, k1:]) * 4.0   # different magnitudes
    a = np.concatenate([a1, a2], axis=1)
    ref = np.maximum(a.astype(np.float64) @ w.astype(np.float64) + b, 0)
    got = host(ops.dense_h2(dev(a1), ops.pack_dense_h2(dev(w)), dev(b), N, True, a2=dev(a2)))
    # the bar of a 1984-term fp32 chain: relative to the output scale (an absolute 1e-5 is below the rounding of
    # one fp32 product at |out| ~ 15)
    scale = float(np.abs(ref).max())
    print("two sources %s: max err %.3g of scale %.3g" % ((M, k1, k2, N), np.abs(got - ref).max(), scale))
    assert np.abs(got - ref).max() <= 2e-6 * scale


def test_deferred_bias_and_relu_on_load(ops):
    """out = relu(relu(pre + in_bias) . W + b): the global fold2/conv1 -> fold2/conv2 hand-over of disn_encode_query"""
    M, K, N = 2048, 512, 256
    pre, w, b = case(M, K, N, 5, positive=False)
    rng = np.random.default_rng(6)
    ib = (rng.standard_normal(K) * 0.7).astype(np.float32)
    h = np.maximum(pre.astype(np.float64) + ib, 0)
    ref = np.maximum(h @ w.astype(np.float64) + b, 0)
    got = host(ops.dense_h2(dev(pre), ops.pack_dense_h2(dev(w)), dev(b), N, True, in_bias=dev(ib)))
    report_close("dense_h2 deferred bias", got, ref, atol=1e-5, rtol=1e-5)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_argument_checks(ops):
    from disn_amd import _lib
    a, w, b = case(64, 64, 64, 1)
    with pytest.raises(ValueError):
        ops.pack_dense_h2(dev(np.zeros((48, 64), np.float32)))
    img = ops.pack_dense_h2(dev(w))
    with pytest.raises(_lib.DisnError) as e:       # k1 = 32 is not a multiple of the 64-column chunk
        ops.dense_h2(dev(a[:, :32]), img, dev(b), 64, True, a2=dev(a[:, 32:]))
    assert e.value.status == -2


# ---- the batched form (disn_amd/csrc/dense_h2w.hip): rows of >= 4 images, a multiple of 128 rows each ----------------
def _ref(a, w, b, relu, in_bias=None, rows=0):
    a = a.astype(np.float64)
    if in_bias is not None:
        a = np.maximum(a + np.repeat(in_bias.astype(np.float64), rows, axis=0), 0)
    r = a @ w.astype(np.float64) + b
    return np.maximum(r, 0) if relu else r


@pytest.mark.parametrize("imgs,rows,K,N,relu", [(4, 2048, 256, 512, True), (8, 2048, 512, 512, False),
                                                (8, 2048, 512, 256, True), (5, 384, 1024, 256, True), (4, 128, 128, 512, True)])
def test_batched_form_vs_float64_and_the_single_image_form(ops, imgs, rows, K, N, relu):
    a, w, b = case(imgs * rows, K, N, imgs + rows + K + N)
    a = a.reshape(imgs, rows, K) * (0.5 + np.arange(imgs, dtype=np.float32)).reshape(imgs, 1, 1)   # every image its own scale
    a = np.ascontiguousarray(a.reshape(imgs * rows, K))
    ref = _ref(a, w, b, relu)
    img = ops.pack_dense_h2(dev(w))
    out, amax = ops.dense_h2(dev(a), img, dev(b), N, relu, want_amax=True, rows_per_image=rows)
    got = host(out)
    for i in range(imgs):
        sl = slice(i * rows, (i + 1) * rows)
        sc = float(np.abs(ref[sl]).max())
        assert np.abs(got[sl] - ref[sl]).max() <= 2e-6 * sc, i
        one = host(ops.dense_h2(dev(a[sl]), img, dev(b), N, relu))            # the four-k-wave tiles
        assert np.abs(got[sl] - one).max() <= 2e-6 * sc, i
    assert float(amax) == float(np.abs(got).max())
    assert np.array_equal(got, host(ops.dense_h2(dev(a), img, dev(b), N, relu, rows_per_image=rows)))
    # other companions, another position, another image count: the same bits
    perm = [imgs - 1, 0] + list(range(1, imgs - 1)) + [0]
    a2 = np.concatenate([a[p * rows:(p + 1) * rows] for p in perm])
    got2 = host(ops.dense_h2(dev(a2), img, dev(b), N, relu, rows_per_image=rows))
    assert np.array_equal(got2[rows:2 * rows], got[:rows]) and np.array_equal(got2[:rows], got[(imgs - 1) * rows:])
    assert np.array_equal(got2[-rows:], got[:rows])


def test_batched_form_two_sources_and_deferred_bias(ops):
    """the local fold2/conv1 ([point 512 | feat 1536 zero-padded] . W, per-image scales of both sources) and the global
    fold2/conv2 (relu(pre + bias[image]) . W) of an eight-step call"""
    imgs, rows = 8, 2048
    M = imgs * rows
    a, w, b = case(M, 2048, 512, 901)
    a1, a2 = np.ascontiguousarray(a[:, :512]), np.ascontiguousarray(a[:, 512:]) * 4.0
    a2[:, 1472:] = 0
    ref = _ref(np.concatenate([a1, a2], axis=1), w, b, True)
    got = host(ops.dense_h2(dev(a1), ops.pack_dense_h2(dev(w)), dev(b), 512, True, a2=dev(a2), rows_per_image=rows))
    sc = float(np.abs(ref).max())
    print("batched two sources: max err %.3g of scale %.3g" % (np.abs(got - ref).max(), sc))
    assert np.abs(got - ref).max() <= 2e-6 * sc
    pre, w, b = case(M, 512, 256, 902, positive=False)
    ib = (np.random.default_rng(903).standard_normal((imgs, 512)) * 0.7).astype(np.float32)
    ref = _ref(pre, w, b, True, in_bias=ib, rows=rows)
    got = host(ops.dense_h2(dev(pre), ops.pack_dense_h2(dev(w)), dev(b), 256, True, in_bias=dev(ib), rows_per_image=rows))
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_per_image_scales_below_the_batched_threshold(ops):
    """three images (or rows not a multiple of 128): the four-k-wave tiles with per-image scales -- every image bit for
    bit what it gets alone"""
    rows, K, N = 192, 256, 512
    a, w, b = case(3 * rows, K, N, 77)
    a[rows:2 * rows] *= 8.0
    img = ops.pack_dense_h2(dev(w))
    got = host(ops.dense_h2(dev(a), img, dev(b), N, True, rows_per_image=rows))
    for i in range(3):
        one = host(ops.dense_h2(dev(a[i * rows:(i + 1) * rows]), img, dev(b), N, True))
        assert np.array_equal(got[i * rows:(i + 1) * rows], one), i


def test_batched_form_k_range_with_addend_equals_the_layer_in_one_piece(ops):
    """how disn_encode_query cuts the local fold2/conv1 of a batched call: rows [0, 1408) of the packed matrix on
    [point | first 896 feature columns] -> pre; rows [1408, 2048) on the remaining 640 columns + pre + bias, ReLU
    (in place).  Against float64 and against the layer run in one piece (another summation order: fp32 rounding)."""
    imgs, rows = 4, 256
    M = imgs * rows
    a, w, b = case(M, 2048, 512, 1234)
    a[:, 1984:] = 0
    point, feat = np.ascontiguousarray(a[:, :512]), np.ascontiguousarray(a[:, 512:])
    img = ops.pack_dense_h2(dev(w))
    zero = dev(np.zeros(512, np.float32))
    pre = ops.dense_h2(dev(point), img, zero, 512, False, a2=dev(np.ascontiguousarray(feat[:, :896])), rows_per_image=rows,
                       image_k=2048, k_begin=0)
    out = ops.dense_h2(dev(np.ascontiguousarray(feat[:, 896:])), img, dev(b), 512, True, rows_per_image=rows, image_k=2048,
                       k_begin=1408, add_in=pre, out=pre)
    got = host(out)
    ref = _ref(a, w, b, True)
    sc = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-6 * sc
    whole = host(ops.dense_h2(dev(point), img, dev(b), 512, True, a2=dev(feat), rows_per_image=rows))
    assert np.abs(got - whole).max() <= 2e-6 * sc
    from disn_amd import _lib
    with pytest.raises(_lib.DisnError) as e:       # a K range needs the batched form (>= 4 images)
        ops.dense_h2(dev(feat[:rows, 896:]), img, dev(b), 512, True, image_k=2048, k_begin=1408)
    assert e.value.status == -2
