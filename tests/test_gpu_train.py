"""GPU parity of the training step: every backward building block and the whole
disn_train_step, through the C ABI, against torch-CPU float64 autograd (oracle/train_oracle.py).
Gradients are sums of up to 4e5 products: the tolerance of each test is relative to the scale of the
reference gradient (max |ref|), written in the test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from conftest import report_close
from oracle import disn_oracle as O
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from disn_amd import ops as _ops
    return _ops


def rel_close(name, got, ref, rtol_of_max):
    ref = np.asarray(ref, np.float64)
    report_close(name, got, ref, atol=rtol_of_max * max(float(np.abs(ref).max()), 1e-30))


# ------------------------------------------------------------------ dense layer ----------------
@pytest.mark.parametrize("M,K,N,relu", [(1000, 64, 256, True), (4096, 512, 512, True), (777, 1472, 512, True),
                                        (2048, 256, 64, False), (33, 512, 256, True), (20000, 512, 512, True)])
def test_dense_backward(ops, M, K, N, relu):
    rng = np.random.default_rng(M + K + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    wd = 1e-3
    at = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    y = at @ wt + bt
    if relu:
        y = torch.relu(y)
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    y32 = host(torch.relu(dev(a) @ dev(w) + dev(b))) if relu else None
    # the mask must come from the SAME activations the gradient is checked against
    if relu:
        y32 = np.where(y.detach().numpy() > 0, np.maximum(y32, 1e-30), 0.0).astype(np.float32)
    da, dw, db = ops.dense_backward(dev(a), dev(w), dev(y32) if relu else None, dev(dy), wd=wd)
    rel_close("da", host(da), at.grad.numpy(), 2e-6)
    rel_close("dw", host(dw), wt.grad.numpy() + wd * w.astype(np.float64), 2e-6)
    rel_close("db", host(db), bt.grad.numpy(), 2e-6)


# ------------------------------------------------------------------ 3x3 conv -------------------
@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 14, 64, 128), (1, 28, 128, 64), (3, 20, 3, 64), (2, 56, 64, 64),
                                          (2, 14, 512, 512), (8, 14, 512, 512)])
def test_conv3x3_backward(ops, B, H, Cin, Cout):
    rng = np.random.default_rng(B * 1000 + H + Cin)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / math.sqrt(9 * Cin)).astype(np.float32)
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    dy = rng.standard_normal((B, H, H, Cout)).astype(np.float32)
    wd = 1e-3
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    y = torch.relu(Fnn.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), bt, padding=1).permute(0, 2, 3, 1))
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    y32 = y.detach().numpy().astype(np.float32)
    y32 = np.where(y.detach().numpy() > 0, np.maximum(y32, 1e-30), 0.0).astype(np.float32)
    dx, dw, db = ops.conv3x3_backward(dev(x), dev(w), dev(y32), dev(dy), wd=wd, need_dx=Cin != 3)
    if Cin != 3:
        rel_close("dx", host(dx), xt.grad.numpy(), 2e-6)
    rel_close("dw", host(dw), wt.grad.numpy() + wd * w.astype(np.float64), 2e-6)
    rel_close("db", host(db), bt.grad.numpy(), 2e-6)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 14, 512, 512), (8, 14, 512, 512), (3, 28, 256, 512), (4, 28, 256, 512),
                                          (4, 28, 512, 512), (4, 56, 128, 256), (5, 56, 256, 256), (4, 112, 64, 128),
                                          (4, 112, 128, 128), (4, 224, 64, 64)])
def test_conv3x3_data_gradient_through_conv_h2(ops, B, H, Cin, Cout):
    """compute mode 2 (and 1) of the step: dx = conv(dz, mirrored transposed kernel) on the inference kernels -- the
    single-image form below four samples, the batched form from four on (all VGG layer shapes with a data gradient,
    channels swapped).  fp32-accurate: 2e-6 of the gradient's scale against float64; sample b's dx does not depend on
    the other samples (per-sample operand scales): checked by re-running sample 0 with the others zeroed."""
    rng = np.random.default_rng(B * 1000 + H + Cin + 7)
    x = rng.standard_normal((B, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / math.sqrt(9 * Cin)).astype(np.float32)
    dz = (rng.standard_normal((B, H, H, Cout)) * rng.uniform(0.01, 10.0, (B, 1, 1, 1))).astype(np.float32)
    dzt = torch.tensor(dz, dtype=torch.float64)
    wt = torch.tensor(w, dtype=torch.float64)
    # dx = conv_transpose of dz: the full correlation with the mirrored kernel
    ref = Fnn.conv_transpose2d(dzt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1).numpy()
    dx, dw, db = ops.conv3x3_backward(dev(x), dev(w), None, dev(dz), wd=1e-3, compute_bf16=2)
    got = host(dx)
    for b in range(B):   # per sample: each has its own scale
        rel_close("dx[%d]" % b, got[b], ref[b], 2e-6)
    # the weight gradient of the same mode: two-term f16 split of x and dz (one scale per tensor) in the TN GEMM
    xt = torch.tensor(x, dtype=torch.float64)
    wt.requires_grad_(True)
    out = Fnn.conv2d(xt.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1)
    (out * dzt).sum().backward()
    rel_close("dw", host(dw), wt.grad.numpy() + 1e-3 * w.astype(np.float64), 2e-6)
    rel_close("db", host(db), dz.astype(np.float64).sum((0, 1, 2)), 2e-6)
    dz0 = dz.copy()
    dz0[1:] = 0
    dx0, _, _ = ops.conv3x3_backward(dev(x), dev(w), None, dev(dz0), wd=0.0, compute_bf16=2)
    assert np.array_equal(host(dx0)[0], got[0])
    assert not host(dx0)[1:].any()


# ------------------------------------------------------------------ pool / resize / gather ------
def test_maxpool_backward(ops):
    rng = np.random.default_rng(5)
    x = np.maximum(rng.standard_normal((2, 28, 28, 64)), 0).astype(np.float32)  # ties at zero, as after ReLU
    dy = rng.standard_normal((2, 14, 14, 64)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    p = Fnn.max_pool2d(xt.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    (p * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    got = host(ops.maxpool2x2_backward(dev(x), dev(dy)))
    ref = xt.grad.numpy()
    nz = x > 0  # among tied zeros the route is irrelevant (ReLU' = 0 there); compare where x > 0
    assert np.array_equal(got[nz], ref[nz].astype(np.float32))
    # and the total routed gradient is conserved window by window
    assert np.array_equal(got.reshape(2, 14, 2, 14, 2, 64).sum((2, 4)), dy)


@pytest.mark.parametrize("hin,c,coff,cs", [(224, 64, 0, 64), (112, 128, 64, 256), (56, 8, 4, 16), (28, 4, 0, 4),
                                           (14, 512, 960, 1472)])
def test_resize_backward(ops, hin, c, coff, cs):
    rng = np.random.default_rng(hin + c)
    B = 2
    dout = rng.standard_normal((B, 137, 137, cs)).astype(np.float32)
    xt = torch.zeros((B, hin, hin, c), dtype=torch.float64, requires_grad=True)
    y = T._resize_legacy(xt, 137, 137)
    (y * torch.tensor(dout[..., coff:coff + c], dtype=torch.float64)).sum().backward()
    got = host(ops.resize_bilinear_backward(dev(dout), hin, hin, channels=c, out_coff=coff))
    rel_close("din", got, xt.grad.numpy(), 1e-6)
    base = rng.standard_normal((B, hin, hin, c)).astype(np.float32)
    acc = dev(base)
    ops.resize_bilinear_backward(dev(dout), hin, hin, channels=c, out_coff=coff, din=acc, accumulate=True)
    rel_close("din+=", host(acc), xt.grad.numpy() + base, 1e-6)


def test_gather_backward(ops):
    rng = np.random.default_rng(9)
    B, N = 2, 300
    xy = rng.uniform(-2.0, 139.0, (B, N, 2)).astype(np.float32)
    xy[0, :8] = [[0, 0], [136, 136], [136.5, 3], [-0.5, 7], [5, 136.9], [-1, 5], [137, 5], [20, 20]]
    xy[1, :40] = xy[1, 40:41]  # many points on one pixel: the atomics must accumulate
    dfeat = rng.standard_normal((B, N, 1472)).astype(np.float32)
    mt = torch.zeros((B, 137, 137, 1472), dtype=torch.float64, requires_grad=True)
    f = T._resampler(mt, xy)
    (f * torch.tensor(dfeat, dtype=torch.float64)).sum().backward()
    got = host(ops.gather_backward(dev(dfeat), dev(xy)))
    rel_close("dfeatmap", got, mt.grad.numpy(), 1e-6)


# ------------------------------------------------------------------ Adam -----------------------
def test_adam_update(ops):
    rng = np.random.default_rng(11)
    n = 4 * 1000 + 64
    w = rng.standard_normal(n).astype(np.float32)
    m = (0.1 * rng.standard_normal(n)).astype(np.float32)
    v = (0.01 * rng.random(n)).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    t, lr = 7, 1e-4
    w2, m2, v2 = T.adam_step(w.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                             v.astype(np.float64), t, lr)
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
    dw, dm, dv = dev(w), dev(m), dev(v)
    ops.adam_update(dw, dev(g), dm, dv, lr_t, grad_scale=0.5)
    report_close("m", host(dm), m2, atol=1e-7)
    report_close("v", host(dv), v2, atol=1e-7)
    report_close("w", host(dw), w2, atol=1e-7, rtol=1e-6)


# ------------------------------------------------------------------ the whole step --------------
def _feed(B, N, seed):
    feed = O.synth_inputs(seed=seed, batch=B, n_points=N)
    rng = np.random.default_rng(seed + 1)
    feed["sample_pc_rot"] = (feed["sample_pc"] @ np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32)).astype(np.float32)
    feed["sdf"] = (0.05 * rng.standard_normal((B, N, 1))).astype(np.float32)
    return feed


def _dev_feed(feed):
    return {k: dev(feed[k]) for k in ("imgs", "trans_mat", "sample_pc", "sample_pc_rot", "sdf")}


@pytest.fixture(scope="module")
def small_step():
    """B=2, N=256, He weights: oracle float64 autograd vs disn_train_step"""
    from disn_amd.train_sdf import Trainer
    from disn_amd.weights import WeightStore
    B, N = 2, 256
    weights = O.init_weights(3, "he")
    feed = _feed(B, N, 21)
    L, grads, pred = T.loss_and_grads(feed, weights, np.float64)
    tr = Trainer(WeightStore(weights), batch_size=B)
    dpred, dl = tr.forward_backward(_dev_feed(feed))
    torch.cuda.synchronize()
    return dict(tr=tr, L=L, grads=grads, pred=pred, dpred=host(dpred), dl=host(dl), feed=feed, weights=weights)


def test_train_step_forward_and_losses(small_step):
    s = small_step
    report_close("pred", s["dpred"], s["pred"].reshape(s["dpred"].shape), atol=2e-5, rtol=1e-5)
    from disn_amd.train_sdf import LOSS_NAMES
    for i, n in enumerate(LOSS_NAMES):
        assert abs(s["dl"][i] - s["L"][n]) <= 1e-5 * max(abs(s["L"][n]), 1.0) + 1e-6, (n, s["dl"][i], s["L"][n])


def test_train_step_gradients(small_step):
    s = small_step
    tr = s["tr"]
    got = tr.flat.to_arrays(tr.grads)
    # The gradients depend on ~1e7 discrete decisions (ReLU sign, pool arg-max) taken on fp32
    # activations; a float64 run takes a handful of them differently and each flip moves a few
    # weight-gradient entries by ~1% of the maximum -- the CPU oracle run in float32 differs from
    # its own float64 run by 1.1e-2 (conv4_2) on this very input; with no flip the agreement is
    # 1e-7 .. 3e-6 (tools/train_debug.py).  The test is therefore flip-tolerant: relative L2 error
    # < 2e-2 and cosine > 0.9995 per variable (a wrong kernel or a missing term gives O(1)); the
    # kernels themselves are held to 2e-6 in the unit tests above, where the masks are inputs.
    # The point MLPs have ReLUs too (512 rows here: one flipped unit is a rank-1 change of every
    # upstream weight gradient, ~2e-3 of the maximum), so they get the same flip-tolerant test.
    rows = []
    for name, ref in s["grads"].items():
        g = got[name].astype(np.float64).ravel()
        r = np.asarray(ref, np.float64).ravel()
        scale = max(float(np.abs(r).max()), 1e-12)
        l2 = float(np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30))
        cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        med = float(np.median(np.abs(g - r))) / scale
        rows.append((l2, cos, med, name))
    rows.sort(reverse=True)
    print("worst (rel L2, cos, median err / max ref):", rows[:4])
    assert rows[0][0] < 2e-2 and min(c for _, c, _, _ in rows) > 0.9995, rows[:6]


def test_train_step_is_repeatable_except_atomics(small_step):
    s = small_step
    tr = s["tr"]
    g1 = tr.grads.clone()
    tr.forward_backward(_dev_feed(s["feed"]))
    torch.cuda.synchronize()
    mlp = slice(int(tr.flat.layout.offset[32]), tr.flat.total)
    assert torch.equal(g1[mlp], tr.grads[mlp])  # no atomics on the MLP side: bit-identical
    assert torch.allclose(g1, tr.grads, rtol=0, atol=1e-4 * float(g1.abs().max()))


def test_trainer_steps_match_oracle_adam(small_step):
    """two optimizer steps from the same state: parameters follow TF Adam on the oracle's gradients"""
    from disn_amd.train_sdf import Trainer, get_learning_rate
    from disn_amd.weights import WeightStore
    s = small_step
    tr = Trainer(WeightStore(s["weights"]), batch_size=2)
    feed = _dev_feed(s["feed"])
    w = {k: np.asarray(v, np.float64) for k, v in s["weights"].items()}
    m = {k: np.zeros_like(v) for k, v in w.items()}
    v_ = {k: np.zeros_like(v) for k, v in w.items()}
    _, losses, lr = tr.step(feed)
    assert lr == get_learning_rate(0, 2)
    for k in w:
        w[k], m[k], v_[k] = T.adam_step(w[k], s["grads"][k], m[k], v_[k], 1, lr)
    got = tr.flat.to_arrays(tr.params)
    # step 1 of Adam moves every weight by ~lr*sign(g): compare the UPDATE, which is what the kernel computes
    for k in ("vgg_16/conv1/conv1_1/weights", "vgg_16/fc8/weights", "sdfprediction/fold2/conv1/weights",
              "sdfprediction_imgfeat/fold2/conv5/biases"):
        upd_ref = w[k] - np.asarray(s["weights"][k], np.float64)
        upd_got = got[k].astype(np.float64) - np.asarray(s["weights"][k], np.float64)
        big = np.abs(s["grads"][k]) > 1e-3 * np.abs(s["grads"][k]).max()  # sign of a ~0 gradient is noise
        report_close(k, upd_got[big], upd_ref[big], atol=2e-6)
    assert float(losses["overall_loss"]) > 0 and tr.step_count == 1


def test_fp32_mfma_mode_tracks_the_default_mode(small_step):
    """precision="f32_mfma" (compute_bf16 = 0: every product on the f32-input MFMA, implicit-GEMM convolutions, a
    launch per sample for the global stream's fold2/conv1) against the default fp32-accurate mode (two-term f16 /
    three-term bf16 kernels): same losses, predictions within 2e-5 of each other, every gradient aligned"""
    from disn_amd.train_sdf import Trainer
    from disn_amd.weights import WeightStore
    s = small_step
    tr = Trainer(WeightStore(s["weights"]), batch_size=2, precision="f32_mfma")
    dpred, dl = tr.forward_backward(_dev_feed(s["feed"]))
    torch.cuda.synchronize()
    report_close("pred", host(dpred), s["dpred"], atol=2e-5, rtol=1e-5)
    for i in range(len(s["dl"])):
        assert abs(float(dl[i]) - float(s["dl"][i])) <= 1e-5 * max(abs(float(s["dl"][i])), 1.0) + 1e-6
    got, ref = tr.flat.to_arrays(tr.grads), s["tr"].flat.to_arrays(s["tr"].grads)
    for name in ref:
        g, r = got[name].astype(np.float64).ravel(), ref[name].astype(np.float64).ravel()
        cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        assert cos > 0.9995, (name, cos)
    tr.close()


def test_config5_shape_step_vs_oracle():
    """BASELINE config 5's per-GPU shape (train/train_sdf.py:371-387: 8 samples x 2048 points per rank), one
    forward + backward against the float64 autograd oracle (VERDICT r3 #2d) -- in the fp32-accurate mode AND in the
    bf16 mode the config names.  Flip-tolerant metric as in test_train_step_gradients (~4e7 ReLU / arg-max decisions
    here; a float64 run takes a handful differently from any fp32 run): fp32-accurate mode relative L2 < 3e-2 and
    cosine > 0.999 per variable; bf16 mode (2^-9 relative operand rounding through 13 + 6 layers) prediction within
    3 % of its scale, sdf loss within 3 %, every gradient cosine > 0.9."""
    from disn_amd.train_sdf import LOSS_NAMES, Trainer
    from disn_amd.weights import WeightStore
    B, N = 8, 2048
    weights = O.init_weights(3, "he")
    feed = _feed(B, N, 55)
    L, grads, pred = T.loss_and_grads(feed, weights, np.float64)
    scale = float(np.abs(pred).max())

    def compare(tr):
        dpred, dl = tr.forward_backward(_dev_feed(feed))
        torch.cuda.synchronize()
        got = tr.flat.to_arrays(tr.grads)
        rows = []
        for name, ref in grads.items():
            g = got[name].astype(np.float64).ravel()
            r = np.asarray(ref, np.float64).ravel()
            l2 = float(np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30))
            cos = float(g @ r / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
            rows.append((l2, cos, name))
        rows.sort(reverse=True)
        perr = float(np.abs(host(dpred).reshape(pred.shape) - pred).max())
        return rows, perr, host(dl)

    tr = Trainer(WeightStore(weights), batch_size=B)
    rows, perr, dl = compare(tr)
    tr.close()
    print("\n[parity cfg5 8x2048 fp32-accurate] max |pred - f64| %.3g (|pred| max %.3g); worst (rel L2, cos): %s" % (
        perr, scale, [(round(a, 5), round(b, 6), n) for a, b, n in rows[:3]]))
    assert perr <= 2e-5 + 1e-5 * scale
    for i, n in enumerate(LOSS_NAMES):
        assert abs(dl[i] - L[n]) <= 1e-5 * max(abs(L[n]), 1.0) + 1e-6, (n, dl[i], L[n])
    assert rows[0][0] < 3e-2 and min(c for _, c, _ in rows) > 0.999, rows[:6]
    del tr
    torch.cuda.empty_cache()

    tb = Trainer(WeightStore(weights), batch_size=B, precision="bf16")
    rows, perr, dl = compare(tb)
    tb.close()
    worst_cos = sorted((c, n) for _, c, n in rows)[:3]
    print("[parity cfg5 8x2048 bf16] max |pred - f64| %.3g = %.3g of scale; sdf_loss %.6g vs %.6g; lowest cosines %s" % (
        perr, perr / scale, dl[2], L[LOSS_NAMES[2]], [(round(c, 4), n) for c, n in worst_cos]))
    assert perr < 3e-2 * scale
    assert abs(dl[2] - L[LOSS_NAMES[2]]) < 3e-2 * abs(L[LOSS_NAMES[2]])
    assert worst_cos[0][0] > 0.9, worst_cos


def test_training_reduces_the_loss():
    from disn_amd.train_sdf import Trainer
    from disn_amd.weights import WeightStore
    B, N = 2, 512
    tr = Trainer(WeightStore(O.init_weights(5, "he")), batch_size=B)
    feed = _dev_feed(_feed(B, N, 33))
    first = None
    for i in range(12):
        _, losses, _ = tr.step(feed)
        if first is None:
            first = float(losses["sdf_loss"])
    last = float(losses["sdf_loss"])
    assert math.isfinite(last) and last < 0.7 * first, (first, last)


def test_checkpoint_roundtrip_with_adam_slots(tmp_path, small_step):
    from disn_amd import tf_checkpoint as tfc_mod
    from disn_amd.train_sdf import Trainer
    from disn_amd.weights import WeightStore
    s = small_step
    tr = Trainer(WeightStore(s["weights"]), batch_size=2)
    feed = _dev_feed(s["feed"])
    tr.step(feed); tr.step(feed)
    prefix = str(tmp_path / "model.ckpt")
    tr.save(prefix)
    tr2 = Trainer(WeightStore(O.init_weights(99, "he")), batch_size=2)
    assert tr2.restore(prefix) == 3 * 56
    # as a restored reference run: Adam's timestep back from beta2_power, the learning-rate schedule at step 0
    # (the reference's Saver leaves `batch` out of the bundle, train/train_sdf.py:285-286)
    assert tr2.adam_t == 2 and tr2.step_count == 0 and "batch" not in tfc_mod.load_checkpoint(prefix)
    assert torch.equal(tr.params, tr2.params) and torch.equal(tr.m, tr2.m) and torch.equal(tr.v, tr2.v)
    # saver.save also writes the `checkpoint` state file: the latest-checkpoint lookup finds the bundle
    from disn_amd import tf_checkpoint as tfc
    assert tfc.get_checkpoint_state(str(tmp_path)) == prefix
    tr.save(str(tmp_path / "model2.ckpt"), include_step=True)          # the opt-in extension + the state file's history
    assert int(tfc.load_checkpoint(str(tmp_path / "model2.ckpt"))["batch"]) == 2
    assert tfc.all_checkpoint_paths(str(tmp_path)) == ["model.ckpt", "model2.ckpt"]
    tr3 = Trainer(WeightStore(O.init_weights(98, "he")), batch_size=2)
    tr3.restore(str(tmp_path / "model2.ckpt"))
    assert tr3.step_count == 2 and tr3.adam_t == 2
    tr3.close()
    tr.step(feed); tr2.step(feed)
    mlp = slice(int(tr.flat.layout.offset[32]), tr.flat.total)
    assert torch.equal(tr.params[mlp], tr2.params[mlp])


def test_trainer_with_one_rank_rccl_group_overlapped_exchange():
    """The data-parallel call sequence on one GPU (tools/ddp_one_rank.py): a one-rank 'nccl' (= RCCL)
    group with the exchange forced on -- head bucket on the side stream gated by the library's
    head-ready event, tail bucket after the step, Adam with grad_scale 1/world.  After one step the
    result must equal the exchange-free trainer bit for bit on everything that is reproducible
    (fc + MLP variables)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ddp_one_rank.py")], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0 and "DDP1_OK" in r.stdout, "\n".join(r.stderr.splitlines()[-25:])


def test_train_one_epoch_from_the_loader(tmp_path):
    """the reference's epoch loop (train/train_sdf.py:349-440): loader thread -> feed -> step -> running
    means, on a synthetic on-disk dataset in the reference's directory layout"""
    import types
    from disn_amd import data_sdf as D
    from disn_amd.train_sdf import LOSS_NAMES, Trainer, train_one_epoch
    from disn_amd.weights import WeightStore
    rng = np.random.default_rng(0)
    info = {"rendered_dir": str(tmp_path / "img"), "sdf_dir": str(tmp_path / "sdf")}
    listinfo = []
    tm = O.DEMO_TRANS_MAT[0]
    for o in range(4):
        pts = rng.uniform(-0.5, 0.5, (600, 3)).astype(np.float32)
        sdf = np.linalg.norm(pts, axis=1, keepdims=True) - 0.3          # a sphere
        D.save_sample(info["sdf_dir"], "03001627", "o%d" % o, np.concatenate([pts, sdf], 1),
                      np.concatenate([pts, sdf], 1), [0, 0, 0, 1], [-1, -1, -1, 1, 1, 1])
        img = rng.integers(0, 256, (137, 137, 4), dtype=np.uint8)
        D.save_view(info["rendered_dir"], "03001627", "o%d" % o, 0, img, tm, np.eye(3), tm)
        listinfo.append(["03001627", "o%d" % o, 0])
    flags = types.SimpleNamespace(num_points=64, num_sample_points=512, batch_size=2, img_h=137, img_w=137,
                                  rot=False, max_epoch=3, cat_limit=100, backcolorwhite=False, alpha=False)
    ds = D.Pt_sdf_img(flags, listinfo=listinfo, info=info, qsize=4, shuffle=True, seed=1)
    ds.start()
    tr = Trainer(WeightStore(O.init_weights(7, "he")), batch_size=2)
    lines = []
    first = train_one_epoch(tr, ds, ds.num_batches, log=lines.append, log_every=1)
    for _ in range(2):
        last = train_one_epoch(tr, ds, ds.num_batches, log=lines.append, log_every=1)
    ds.shutdown()
    tr.close()
    assert set(first) == set(LOSS_NAMES) and tr.step_count == 6 and len(lines) == 6
    assert all(math.isfinite(v) for v in last.values()) and last["sdf_loss"] < first["sdf_loss"]
