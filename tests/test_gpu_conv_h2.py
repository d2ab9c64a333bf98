"""GPU parity of the single-image convolution (disn_amd/csrc/conv_h2.hip, through the C ABI) against the
oracle's float64 3x3 SAME convolution (oracle/disn_oracle.py:conv2d; models/CNN/vgg.py:187-196): every VGG
layer shape at B = 1, every workgroup tiling on ragged shapes, the fused 2x2 max pool, the activation maximum
the next layer scales by, run-to-run bit reproducibility, and the device weight image against its CPU
restatement (tests/conv_h2_emulation.py)."""
import numpy as np
import pytest
import torch

import conv_h2_emulation as E
from conftest import report_close
from oracle import disn_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from disn_amd import ops as _ops
    return _ops


def case(B, H, W, Cin, Cout, seed, relu_input=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    if relu_input:
        x = np.maximum(x, 0) * 2.0
    w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    return x, w, b


def test_weight_image_equals_the_layout_restatement(ops):
    _, w, _ = case(1, 2, 2, 128, 64, 5)
    img = host(ops.pack_conv_h2(dev(w)))
    ref, s = E.pack(w)
    n = ref.size * 2
    assert np.array_equal(img[:n].view(np.float16).reshape(ref.shape), ref)
    inv_sw = img[n:n + 4 * 64].view(np.float32)             # the tail: 1 / s_w per output channel
    assert np.array_equal(inv_sw, np.float32(1.0) / s)


VGG_SHAPES = [(224, 64, 64), (112, 64, 128), (112, 128, 128), (56, 128, 256), (56, 256, 256), (28, 256, 512),
              (28, 512, 512), (14, 512, 512)]


@pytest.mark.parametrize("hw,cin,cout", VGG_SHAPES)
def test_vgg_layer_shapes_vs_float64(ops, hw, cin, cout):
    x, w, b = case(1, hw, hw, cin, cout, hw + cin)
    ref = O.conv2d(x, w, b, "SAME", True, dtype=np.float64)
    out, pooled, amax = ops.conv3x3_h2(dev(x), ops.pack_conv_h2(dev(w)), dev(b), cout, True, pool=True,
                                       want_amax=True)
    got = host(out)
    # products carry 2^-22-relative operand errors, accumulation is fp32 over K <= 4608: same bar as the
    # three-term kernel (tests/test_gpu_kernels.py)
    report_close("conv3x3_h2 %s" % ((hw, cin, cout),), got, ref, atol=1e-5, rtol=1e-5)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    print("hw %d cin %d cout %d: max err %.3g, scale %.3g, relative %.3g" % (hw, cin, cout, err, scale, err / scale))
    assert err <= 2e-6 * scale
    assert np.array_equal(host(pooled), got.reshape(1, hw // 2, 2, hw // 2, 2, cout).max(axis=(2, 4)))
    assert float(amax) == float(np.abs(got).max())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 14, 14, 512, 512), (3, 14, 14, 64, 128), (2, 10, 12, 192, 64),
                                            (5, 14, 14, 512, 512), (16, 14, 14, 512, 512)])
def test_whole_image_tiling_of_14_pixel_layers(ops, B, H, W, Cin, Cout):
    """tiling 10: one workgroup per image and n-block, the image as ONE patch (seven 32-pixel blocks), four k-waves --
    what tiling 0 takes for 14 x 14 layers when B * Cout / 32 >= 200 (conv5_x of a 16-image call).  Against float64;
    pool and maximum from the same values; an image's bits do not depend on its companions; every call of four images
    and more (tiling 0) gives these bits: four k-waves, as a whole image or in two-row patches."""
    x, w, b = case(B, H, W, Cin, Cout, 1000 + B + H + Cin, relu_input=Cin == 512)
    ref = O.conv2d(x, w, b, "SAME", True, dtype=np.float64)
    img = ops.pack_conv_h2(dev(w))
    out, pooled, amax = ops.conv3x3_h2(dev(x), img, dev(b), Cout, True, pool=True, want_amax=True, tiling=10)
    got = host(out)
    scale = float(np.abs(ref).max())
    report_close("conv3x3_h2 whole-image tiling %s" % ((B, H, W, Cin, Cout),), got, ref, atol=2e-6 * scale)
    assert np.array_equal(host(pooled), got.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4)))
    assert float(amax) == float(np.abs(got).max())
    alone = host(ops.conv3x3_h2(dev(x[B - 1:]), img, dev(b), Cout, True, tiling=10))
    assert np.array_equal(alone[0], got[B - 1])
    if B >= 4:
        # tiling 0 (inference): four k-waves in SEGMENTS of two chunks -- tiling 19's bits whatever the patch; tiling 18
        # (the training step's selection): the plain four-k-wave chains of tiling 10
        seg = host(ops.conv3x3_h2(dev(x), img, dev(b), Cout, True, tiling=19))
        assert float(np.abs(seg - ref).max()) <= 1e-6 * scale
        assert np.array_equal(host(ops.conv3x3_h2(dev(x), img, dev(b), Cout, True, tiling=0)), seg)
        assert np.array_equal(host(ops.conv3x3_h2(dev(x), img, dev(b), Cout, True, tiling=18)), got)
    # four k-waves: the bits of the other four-k-wave tilings
    if Cin % 128:
        assert np.array_equal(host(ops.conv3x3_h2(dev(x), img, dev(b), Cout, True, tiling=1)), got)


@pytest.mark.parametrize("tiling", [1, 2, 3, 4])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 20, 18, 64, 128), (1, 30, 44, 128, 64), (3, 6, 8, 192, 64)])
def test_every_tiling_on_ragged_shapes(ops, tiling, B, H, W, Cin, Cout):
    x, w, b = case(B, H, W, Cin, Cout, 100 * tiling + H, relu_input=False)
    ref = O.conv2d(x, w, b, "SAME", False, dtype=np.float64)
    out, pooled, amax = ops.conv3x3_h2(dev(x), ops.pack_conv_h2(dev(w)), dev(b), Cout, False, pool=True,
                                       want_amax=True, tiling=tiling)
    got = host(out)
    report_close("conv3x3_h2 tiling %d %s" % (tiling, (B, H, W, Cin, Cout)), got, ref, atol=1e-5, rtol=1e-5)
    assert np.array_equal(host(pooled), got.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4)))
    assert float(amax) == float(np.abs(got).max())


def test_tilings_agree_bit_for_bit_and_runs_repeat(ops):
    """the K order of an output element (k16 block, tap; the fixed tree over the k-waves) depends on the number of
    k-waves only: tilings with four k-waves agree bit for bit, as do those with eight (Cin % 128 == 0: tilings 1,
    2), and every run repeats"""
    x, w, b = case(1, 28, 28, 128, 128, 77)
    img, xd, bd = ops.pack_conv_h2(dev(w)), dev(x), dev(b)
    o = {t: host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=t)) for t in (1, 2, 3, 4)}
    assert np.array_equal(o[1], o[2]) and np.array_equal(o[3], o[4])
    assert np.abs(o[1] - o[3]).max() <= 1e-6 * np.abs(o[3]).max()
    for t in (2, 3):
        assert np.array_equal(o[t], host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=t)))
    x, w, b = case(1, 28, 28, 64, 64, 78)        # Cin = 64: four k-waves everywhere
    img, xd, bd = ops.pack_conv_h2(dev(w)), dev(x), dev(b)
    o = [host(ops.conv3x3_h2(xd, img, bd, 64, True, tiling=t)) for t in (1, 2, 3, 4)]
    for v in o[1:]:
        assert np.array_equal(o[0], v)


def test_14_pixel_layers_of_a_batched_call_run_with_four_k_waves(ops):
    """14-pixel layers (conv5_x) stay on conv_h2.hip; in a call of four images and more they run with FOUR k-waves in
    segments of two chunks (two-row patches, two workgroups per CU, or the whole-image tiling in large calls): the same
    bits whatever the tiling -- here four copies of one image against that image through tiling 19 -- and within fp32
    rounding of the single-image call's eight-k-wave tree."""
    hw, cin, cout = 14, 512, 512
    x, w, b = case(1, hw, hw, cin, cout, 77 + hw)
    img = ops.pack_conv_h2(dev(w))
    one, pool1, _ = ops.conv3x3_h2(dev(x), img, dev(b), cout, True, pool=True, want_amax=True, tiling=19)
    four, pool4, amax4 = ops.conv3x3_h2(dev(np.repeat(x, 4, axis=0)), img, dev(b), cout, True, pool=True, want_amax=True)
    for k in range(4):
        assert torch.equal(four[k], one[0]) and torch.equal(pool4[k], pool1[0]), k
    assert float(amax4) == float(one.abs().max())
    single = ops.conv3x3_h2(dev(x), img, dev(b), cout, True)      # eight k-waves
    assert float((single - one).abs().max()) <= 2e-6 * float(one.abs().max())


# ---- the batched form (disn_amd/csrc/conv_h2w.hip): tiling 5..9 force its variants 1..5, tiling 0 takes it from
# four images per call on (28-pixel layers and larger) --------------------------------------------------------------
WIDE_ONE_KWAVE, WIDE_TWO_KWAVES = (5, 6, 7), (8, 9)
WIDE_SEGMENTED = (12, 13)   # two K halves in segments of two chunks (round 6; one k-wave + LDS park | two k-waves): what tiling 0 takes for Cin >= 128


@pytest.mark.parametrize("tiling", WIDE_ONE_KWAVE + WIDE_TWO_KWAVES + WIDE_SEGMENTED)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 30, 44, 64, 128), (1, 36, 62, 128, 128), (3, 28, 28, 192, 128)])
def test_every_batched_variant_on_ragged_shapes(ops, tiling, B, H, W, Cin, Cout):
    if tiling in WIDE_SEGMENTED and Cin < 128:
        pytest.skip("the segmented variants need two K halves of whole segments (Cin >= 128)")
    x, w, b = case(B, H, W, Cin, Cout, 100 * tiling + H, relu_input=False)
    ref = O.conv2d(x, w, b, "SAME", False, dtype=np.float64)
    out, pooled, amax = ops.conv3x3_h2(dev(x), ops.pack_conv_h2(dev(w)), dev(b), Cout, False, pool=True,
                                       want_amax=True, tiling=tiling)
    got = host(out)
    report_close("conv3x3_h2 batched variant %d %s" % (tiling - 4, (B, H, W, Cin, Cout)), got, ref, atol=1e-5, rtol=1e-5)
    assert float(np.abs(got - ref).max()) <= 2e-6 * float(np.abs(ref).max())
    assert np.array_equal(host(pooled), got.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4)))
    assert float(amax) == float(np.abs(got).max())


def test_batched_variants_with_the_same_k_waves_agree_bit_for_bit_and_runs_repeat(ops):
    """the K order of an output element depends on the number of k-waves only (one: k16 blocks ascending, taps inside;
    two: p0 + p1), not on the patch or the n-waves per workgroup"""
    x, w, b = case(2, 40, 60, 128, 128, 311)
    img, xd, bd = ops.pack_conv_h2(dev(w)), dev(x), dev(b)
    o = {t: host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=t)) for t in WIDE_ONE_KWAVE + WIDE_TWO_KWAVES}
    assert np.array_equal(o[5], o[6]) and np.array_equal(o[6], o[7])
    assert np.array_equal(o[8], o[9])
    assert np.abs(o[5] - o[8]).max() <= 2e-6 * np.abs(o[8]).max()      # two fp32 summation orders
    for t in (5, 6, 8, 9):
        assert np.array_equal(o[t], host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=t)))
    # the segmented variant (tiling 0's choice for Cin >= 128): its own order -- segments of two chunks summed in fp32 --
    # closer to the float64 convolution than the 216-MFMA chain of the one-k-wave variants; runs repeat
    seg = host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=12))
    assert np.array_equal(seg, host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=12)))
    assert np.array_equal(seg, host(ops.conv3x3_h2(xd, img, bd, 128, True, tiling=13))), "one k-wave + park vs two k-waves"
    assert np.array_equal(seg, host(ops.conv3x3_h2(dev(np.concatenate([x, x])), img, bd, 128, True, tiling=0))[:2])   # B = 4: tiling 0 takes it
    ref = O.conv2d(x, w, b, "SAME", True, dtype=np.float64)
    e_seg, e_one = np.sqrt(((seg - ref) ** 2).mean()), np.sqrt(((o[5] - ref) ** 2).mean())
    print("rms error / layer maximum: segmented %.3g, one k-wave %.3g" % (e_seg / np.abs(ref).max(), e_one / np.abs(ref).max()))
    assert e_seg < e_one and np.abs(seg - ref).max() <= 1e-6 * np.abs(ref).max()


@pytest.mark.parametrize("hw,cin,cout", [(224, 64, 64), (112, 64, 128), (112, 128, 128), (56, 128, 256), (56, 256, 256),
                                         (28, 256, 512), (28, 512, 512)])
def test_batched_form_on_the_vgg_layer_shapes(ops, hw, cin, cout):
    """tiling 0 with four images per call = the batched form.  Against float64 (first image), against the
    single-image kernel (every image: another summation order, fp32 rounding only), the fused pool, the maximum,
    and: an image's bits do not depend on the images it travels with nor on its position in the call."""
    x, w, b = case(4, hw, hw, cin, cout, hw + cin + 1)
    x *= np.array([1.0, 0.25, 3.0, 1e-3], np.float32).reshape(4, 1, 1, 1)        # every image its own scale
    img, bd = ops.pack_conv_h2(dev(w)), dev(b)
    out, pooled, amax = ops.conv3x3_h2(dev(x), img, bd, cout, True, pool=True, want_amax=True)
    got = host(out)
    ref0 = O.conv2d(x[:1], w, b, "SAME", True, dtype=np.float64)
    sc0 = float(np.abs(ref0).max())
    err0 = float(np.abs(got[:1] - ref0).max())
    print("batched hw %d cin %d cout %d: max err %.3g of scale %.3g (%.3g)" % (hw, cin, cout, err0, sc0, err0 / sc0))
    assert err0 <= 2e-6 * sc0
    for k in range(4):
        one = host(ops.conv3x3_h2(dev(x[k:k + 1]), img, bd, cout, True))
        assert np.abs(got[k] - one[0]).max() <= 2e-6 * np.abs(one).max(), k
    assert np.array_equal(host(pooled), got.reshape(4, hw // 2, 2, hw // 2, 2, cout).max(axis=(2, 4)))
    assert float(amax) == float(np.abs(got).max())
    if cin >= 128:   # tiling 0 = the segmented variants: one k-wave + LDS park (12) and two k-waves (13) -- the same bits
        assert np.array_equal(host(ops.conv3x3_h2(dev(x), img, bd, cout, True, tiling=12)), got)
        assert np.array_equal(host(ops.conv3x3_h2(dev(x), img, bd, cout, True, tiling=13)), got)
    # other companions, other position, other batch size (5): the same bits
    x2 = np.concatenate([x[3:4], x[3:4] * 0.5, x[0:1], x[1:2], x[2:3]], axis=0)
    got2 = host(ops.conv3x3_h2(dev(x2), img, bd, cout, True))
    assert np.array_equal(got2[2], got[0]) and np.array_equal(got2[0], got[3]) and np.array_equal(got2[4], got[2])


def test_zero_input_and_tiny_activations(ops):
    x, w, b = case(1, 14, 14, 64, 64, 9)
    img = ops.pack_conv_h2(dev(w))
    out0 = host(ops.conv3x3_h2(dev(np.zeros_like(x)), img, dev(b), 64, True))
    assert np.array_equal(out0, np.broadcast_to(np.maximum(b, 0), out0.shape))
    xs = x * np.float32(1e-12)                        # the scale follows the tensor: relative accuracy is kept
    ref = O.conv2d(xs, w, np.zeros_like(b), "SAME", True, dtype=np.float64)
    got = host(ops.conv3x3_h2(dev(xs), img, dev(np.zeros_like(b)), 64, True))
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_argument_checks(ops):
    from disn_amd import _lib
    x, w, b = case(1, 8, 8, 64, 64, 1)
    img = ops.pack_conv_h2(dev(w))
    with pytest.raises(_lib.DisnError) as e:
        ops.conv3x3_h2(dev(x[:, :7]), img, dev(b), 64, True, pool=True)       # odd height with a pool
    assert e.value.status == -2
    with pytest.raises(ValueError):
        ops.pack_conv_h2(dev(np.zeros((3, 3, 32, 64), np.float32)))


@pytest.mark.parametrize("B,H,W", [(1, 224, 224), (2, 37, 45), (1, 5, 3)])
def test_conv1_1_direct_vs_float64(ops, B, H, W):
    rng = np.random.default_rng(H)
    x = rng.random((B, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64)) * np.sqrt(2.0 / 27)).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    ref = O.conv2d(x, w, b, "SAME", True, dtype=np.float64)
    out, amax = ops.conv1_1(dev(x), dev(w), dev(b), True, want_amax=True)
    got = host(out)
    report_close("conv1_1 direct %s" % ((B, H, W),), got, ref, atol=2e-6, rtol=2e-6)   # 27-term fp32 FMA chain
    assert float(amax) == float(np.abs(got).max())


def test_encoder_through_the_h2_kernels_equals_the_three_term_path():
    """same weights, same image: the single-image kernels (engine default) against the implicit-GEMM path"""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    store = WeightStore.random_init(3, mode="he")
    img = O.synth_inputs(5, 1, 8)["imgs"]
    a = SdfEngine(store, conv_h2=True).encode(img)
    b = SdfEngine(store, conv_h2=False).encode(img)
    torch.cuda.synchronize()
    for ta, tb, name in zip(a.taps, b.taps, ("conv1_2", "conv2_2", "conv3_3", "conv4_3", "conv5_3")):
        sc = float(tb.abs().max())
        d = float((ta - tb).abs().max())
        print("%s: max |h2 - x3| %.3g of scale %.3g" % (name, d, sc))
        assert d <= 4e-6 * sc
    sc = float(b.embedding.abs().max())
    assert float((a.embedding - b.embedding).abs().max()) <= 4e-6 * sc
