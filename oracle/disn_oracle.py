"""CPU oracle for the DISN per-point SDF query path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``disn_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do.  The product path runs on hand-written HIP kernels and fails loudly
when the extension is missing.

PARITY UNPINNED: the arithmetic of this path lives in TensorFlow 1.10
(``tf.image.resize_bilinear``, ``tf.contrib.resampler``, slim ``vgg_16``,
``tf.nn.conv2d``), which is neither vendored in /root/reference nor installable
here, and the reference ships no golden vector for the path (its only golden
artefact, demo/result.obj, is listed in .MISSING_LARGE_BLOBS).  This file is a
from-spec restatement; each function cites the reference file:line it follows
and the TF-1.10 kernel semantics it encodes.  What *can* be
pinned is pinned in tests/golden (see tests/golden/make_golden.py): the pure
numpy/struct pieces of the reference (``to_binary``, ``getBlenderProj``, the
split-size arithmetic) are executed from the reference sources themselves, and
the bilinear/conv pieces are cross-checked against independent torch ops.

Arithmetic conventions: everything is float32 with one rounding per operation
(numpy never fuses multiply-add), in the operation order written here.  The
HIP kernels for the element-wise rows (A, D, E, F) follow exactly this order
with FMA contraction disabled, so they are compared bit-for-bit; GEMM-shaped
rows (B, C, G) are compared within a stated tolerance.  Every GEMM-shaped
function takes ``dtype`` so a float64 shadow can be produced to bound the
float32 noise floor.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------
# constants of the path
# --------------------------------------------------------------------------
IMG_H = IMG_W = 137                    # models/model_normalization.py:249-250 clamp [0,136]
VGG_SIZE = 224                         # models/model_normalization.py:47 (img_size default)
TAP_NAMES = ("conv1_2", "conv2_2", "conv3_3", "conv4_3", "conv5_3")  # :171-183
TAP_CHANNELS = (64, 128, 256, 512, 512)                              # Σ = 1472 (:41)
FEAT_DIM = 1472
VGG_CFG = (  # models/CNN/vgg.py:187-196  (scope, n_convs, channels)
    ("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512))
SDF_WEIGHT = 10.0                      # test/create_sdf.py:285
DEMO_TRANS_MAT = np.asarray(           # demo/demo.py:272-276
    [[[-68.453156, 5.5086656, -0.37556022],
      [-17.138561, -84.685486, -0.250198],
      [-47.284092, -3.6569588, 0.2493176],
      [101.133705, 101.34268, 1.4305686]]], dtype=np.float32)


# --------------------------------------------------------------------------
# variable namespace -- shapes keyed by TF variable name
# --------------------------------------------------------------------------
def variable_shapes(num_classes: int = 1024) -> Dict[str, Tuple[int, ...]]:
    """TF variable names -> shapes for the two-stream regression model.

    VGG: models/CNN/vgg.py:187-214 under scope 'vgg_16'
    (models/model_normalization.py:76).  Decoder: models/sdfnet.py:71-88,173-186
    under scopes 'sdfprediction' / 'sdfprediction_imgfeat'
    (models/model_normalization.py:194,199); variable leaf names 'weights' /
    'biases' from utils/tf_util.py:163,173.
    """
    shapes: Dict[str, Tuple[int, ...]] = {}
    cin = 3
    for scope, n, cout in VGG_CFG:
        for j in range(1, n + 1):
            nm = "vgg_16/%s/%s_%d" % (scope, scope, j)
            shapes[nm + "/weights"] = (3, 3, cin, cout)
            shapes[nm + "/biases"] = (cout,)
            cin = cout
    shapes["vgg_16/fc6/weights"] = (7, 7, 512, 4096)
    shapes["vgg_16/fc6/biases"] = (4096,)
    shapes["vgg_16/fc7/weights"] = (1, 1, 4096, 4096)
    shapes["vgg_16/fc7/biases"] = (4096,)
    shapes["vgg_16/fc8/weights"] = (1, 1, 4096, num_classes)
    shapes["vgg_16/fc8/biases"] = (num_classes,)
    for scope, k_concat in (("sdfprediction", 512 + num_classes),
                            ("sdfprediction_imgfeat", 512 + FEAT_DIM)):
        for nm, ci, co in (("fold1/conv1", 3, 64), ("fold1/conv2", 64, 256),
                           ("fold1/conv3", 256, 512), ("fold2/conv1", k_concat, 512),
                           ("fold2/conv2", 512, 256), ("fold2/conv5", 256, 1)):
            shapes["%s/%s/weights" % (scope, nm)] = (1, 1, ci, co)
            shapes["%s/%s/biases" % (scope, nm)] = (co,)
    return shapes


def init_weights(seed: int = 0, mode: str = "xavier", num_classes: int = 1024) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights.

    mode 'xavier': uniform ±sqrt(6/(fan_in+fan_out)), fan = kh*kw*C, zero biases
    -- tf.contrib.layers.xavier_initializer (utils/tf_util.py:41) and the slim
    default.  mode 'he': N(0, 2/fan_in) weights and N(0, 0.1) biases, so that
    activations stay O(1) through 16 ReLU layers and a 1e-5 absolute tolerance
    is meaningful.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shp in variable_shapes(num_classes).items():
        if name.endswith("/weights"):
            kh, kw, ci, co = shp
            fan_in, fan_out = kh * kw * ci, kh * kw * co
            if mode == "xavier":
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                w = rng.uniform(-lim, lim, size=shp)
            elif mode == "he":
                w = rng.normal(0.0, math.sqrt(2.0 / fan_in), size=shp)
            else:
                raise ValueError(mode)
            out[name] = w.astype(np.float32)
        else:
            if mode == "xavier":
                out[name] = np.zeros(shp, np.float32)
            else:
                out[name] = rng.normal(0.0, 0.1, size=shp).astype(np.float32)
    return out


def trained_like_weights(seed: int = 0, sigma: float = 1.0, outlier_frac: float = 0.01, outlier_gain: float = 1.0e3,
                         heavy_tail_df: float = 4.0, num_classes: int = 1024) -> Dict[str, np.ndarray]:
    """Synthetic weights with the STATISTICS of trained ones (the released SDF_DISN / vgg_16.ckpt checkpoints of
    README.md:27-39 are not available offline): heavy-tailed entries (Student-t, `heavy_tail_df` degrees of freedom,
    He variance 2 / fan_in), N(0, 0.1) biases, and per-channel gains -- every output channel c of every hidden layer
    is multiplied by g_c = exp(sigma * N(0,1)), a fraction `outlier_frac` of the channels by a further `outlier_gain`,
    and the rows every consumer of that channel reads are divided by g_c (next convolution, fc6, the local stream's
    fold2/conv1 rows of the five taps, the global stream's fold2/conv1 rows of the embedding).  ReLU and max-pool are
    positively homogeneous and resize / resampler are linear, so in exact arithmetic the network function is that of
    the ungained weights (|pred| stays O(1), the 1e-5 absolute bar stays meaningful), while every activation tensor
    now has log-normal channel magnitudes with outlier channels 10^3 above the rest -- what a per-tensor power-of-two
    operand scale (conv_h2w / dense_h2w / mlp_fused) has to survive."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    t_std = math.sqrt(heavy_tail_df / (heavy_tail_df - 2.0))
    for name, shp in variable_shapes(num_classes).items():
        if name.endswith("/weights"):
            kh, kw, ci, co = shp
            w = rng.standard_t(heavy_tail_df, size=shp) * (math.sqrt(2.0 / (kh * kw * ci)) / t_std)
            out[name] = w.astype(np.float64)
        else:
            out[name] = rng.normal(0.0, 0.1, size=shp).astype(np.float64)
    convs = ["vgg_16/%s/%s_%d" % (sc, sc, j) for sc, n, _ in VGG_CFG for j in range(1, n + 1)]
    chain = []          # (producer, [(consumer, first input row of the producer's channels)])
    tap_row = {nm: 512 + sum(TAP_CHANNELS[:i]) for i, nm in enumerate(TAP_NAMES)}
    for i, nm in enumerate(convs):
        cons = [(convs[i + 1], 0)] if i + 1 < len(convs) else [("vgg_16/fc6", 0)]
        leaf = nm.rsplit("/", 1)[1]
        if leaf in tap_row:
            cons.append(("sdfprediction_imgfeat/fold2/conv1", tap_row[leaf]))
        chain.append((nm, cons))
    chain += [("vgg_16/fc6", [("vgg_16/fc7", 0)]), ("vgg_16/fc7", [("vgg_16/fc8", 0)]),
              ("vgg_16/fc8", [("sdfprediction/fold2/conv1", 512)])]
    for sc in ("sdfprediction", "sdfprediction_imgfeat"):
        seq = ["fold1/conv1", "fold1/conv2", "fold1/conv3", "fold2/conv1", "fold2/conv2", "fold2/conv5"]
        for a, b in zip(seq[:-1], seq[1:]):
            chain.append(("%s/%s" % (sc, a), [("%s/%s" % (sc, b), 0)]))
    for prod, cons in chain:
        co = out[prod + "/biases"].shape[0]
        g = np.exp(sigma * rng.standard_normal(co))
        g = np.where(rng.random(co) < outlier_frac, g * outlier_gain, g)
        out[prod + "/weights"] = out[prod + "/weights"] * g
        out[prod + "/biases"] = out[prod + "/biases"] * g
        for cname, row0 in cons:
            w = out[cname + "/weights"]
            w[:, :, row0:row0 + co, :] = w[:, :, row0:row0 + co, :] / g[None, None, :, None]
    return {k: v.astype(np.float32) for k, v in out.items()}


# --------------------------------------------------------------------------
# row A / E : tf.image.resize_bilinear, TF1 legacy (align_corners=False)
# --------------------------------------------------------------------------
def resize_index_table(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(lower, upper, lerp) per output index -- TF-1.10 resize_bilinear_op.cc
    compute_interpolation_weights: scale = in/float(out) (float32);
    in = i*scale; lower = floor(in); upper = min(lower+1, in_size-1);
    lerp = in - lower.  No half-pixel offset (call sites
    models/model_normalization.py:72,171-183 never pass align_corners)."""
    scale = F32(in_size) / F32(out_size)
    idx = np.arange(out_size, dtype=np.float32)
    src = (idx * scale).astype(np.float32)
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, in_size - 1)
    lerp = (src - lo.astype(np.float32)).astype(np.float32)
    return lo, hi, lerp


def resize_bilinear_legacy(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img [B,H,W,C] float32 -> [B,out_h,out_w,C].
    top = tl + (tr-tl)*xl; bot = bl + (br-bl)*xl; out = top + (bot-top)*yl."""
    img = np.asarray(img, dtype=np.float32)
    B, H, W, C = img.shape
    ylo, yhi, yl = resize_index_table(H, out_h)
    xlo, xhi, xl = resize_index_table(W, out_w)
    xl = xl[None, None, :, None]
    yl = yl[None, :, None, None]
    rows_lo = img[:, ylo]           # [B,out_h,W,C]
    rows_hi = img[:, yhi]
    tl, tr = rows_lo[:, :, xlo], rows_lo[:, :, xhi]
    bl, br = rows_hi[:, :, xlo], rows_hi[:, :, xhi]
    top = (tl + (tr - tl) * xl).astype(np.float32)
    bot = (bl + (br - bl) * xl).astype(np.float32)
    return (top + (bot - top) * yl).astype(np.float32)


def resize_bilinear_legacy_mt(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """resize_bilinear_legacy with torch-CPU element-wise ops (multi-threaded): the same expressions in the
    same order, one float32 rounding per operation, so the result is bit-identical to the numpy form
    (tests/test_oracle.py).  Only bench.py's cpu_baseline leg uses it, so that the CPU figure is not
    dominated by single-threaded numpy fancy indexing."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    B, H, W, C = t.shape
    ylo, yhi, yl = resize_index_table(H, out_h)
    xlo, xhi, xl = resize_index_table(W, out_w)
    ylo, yhi, xlo, xhi = (torch.from_numpy(a) for a in (ylo, yhi, xlo, xhi))
    xl = torch.from_numpy(xl)[None, None, :, None]
    yl = torch.from_numpy(yl)[None, :, None, None]
    rows_lo, rows_hi = t.index_select(1, ylo), t.index_select(1, yhi)
    tl, tr = rows_lo.index_select(2, xlo), rows_lo.index_select(2, xhi)
    bl, br = rows_hi.index_select(2, xlo), rows_hi.index_select(2, xhi)
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).numpy()


def resize_bilinear_legacy_blocks(img: np.ndarray, out_h: int, out_w: int, workers: int = 1) -> np.ndarray:
    """resize_bilinear_legacy in two passes over row blocks on a thread pool (numpy releases the GIL inside its
    loops).  Pass 1 forms the x-interpolated SOURCE rows T[h] = row[xlo] + (row[xhi] - row[xlo]) * xl once per source
    row; an output row's `top` / `bot` of the one-pass form are exactly T[ylo] / T[yhi] (the same float32 expression
    on the same operands), so pass 2's T[ylo] + (T[yhi] - T[ylo]) * yl reproduces the one-pass result bit for bit
    (tests/test_oracle.py) -- TF's CPU kernel caches the x weights per row the same way.  bench.py's cpu_baseline
    leg uses it for the five 137x137 tap up-samples (110 MB of output: memory bound, scales with the threads)."""
    from concurrent.futures import ThreadPoolExecutor
    img = np.ascontiguousarray(img, dtype=np.float32)
    B, H, W, C = img.shape
    ylo, yhi, yl = resize_index_table(H, out_h)
    xlo, xhi, xl = resize_index_table(W, out_w)
    xl4 = xl[None, None, :, None]
    T = np.empty((B, H, out_w, C), np.float32)
    out = np.empty((B, out_h, out_w, C), np.float32)
    workers = max(1, int(workers)) if B * out_h * out_w * C >= (1 << 21) else 1   # small outputs: pool start-up dominates

    def blocks(n):
        nb = max(1, min(n, 4 * workers))
        edges = np.linspace(0, n, nb + 1).astype(int)
        return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]

    def pass1(rng):
        a, b = rng
        rows = img[:, a:b]
        tl = rows[:, :, xlo]
        d = rows[:, :, xhi]
        np.subtract(d, tl, out=d)
        np.multiply(d, xl4, out=d)
        np.add(tl, d, out=T[:, a:b])

    def pass2(rng):
        a, b = rng
        top = T[:, ylo[a:b]]
        d = T[:, yhi[a:b]]
        np.subtract(d, top, out=d)
        np.multiply(d, yl[None, a:b, None, None], out=d)
        np.add(top, d, out=out[:, a:b])

    if workers == 1:
        for r in blocks(H):
            pass1(r)
        for r in blocks(out_h):
            pass2(r)
    else:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(pass1, blocks(H)))
            list(ex.map(pass2, blocks(out_h)))
    return out


def resampler_mt(data: np.ndarray, warp: np.ndarray) -> np.ndarray:
    """resampler with torch-CPU ops (multi-threaded), same expressions / order as the numpy form: bit-identical"""
    import torch
    d = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
    w = torch.from_numpy(np.ascontiguousarray(warp, dtype=np.float32))
    B, H, W, C = d.shape
    out = torch.zeros((B, w.shape[1], C), dtype=torch.float32)
    for b in range(B):
        x, y = w[b, :, 0], w[b, :, 1]
        ok = (x > -1.0) & (y > -1.0) & (x < W) & (y < H)
        zero = torch.zeros((), dtype=torch.float32)
        xs, ys = torch.where(ok, x, zero), torch.where(ok, y, zero)
        fx, fy = torch.floor(xs), torch.floor(ys)
        cx, cy = fx + 1.0, fy + 1.0
        dx, dy = cx - xs, cy - ys
        ifx, ify, icx, icy = (a.to(torch.int64) for a in (fx, fy, cx, cy))
        flat = d[b].reshape(H * W, C)

        def get(ix, iy):
            inb = (ix >= 0) & (iy >= 0) & (ix < W) & (iy < H)
            v = flat.index_select(0, iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1))
            return torch.where(inb[:, None], v, zero)

        v = (dx * dy)[:, None] * get(ifx, ify)
        v = v + ((1.0 - dx) * (1.0 - dy))[:, None] * get(icx, icy)
        v = v + (dx * (1.0 - dy))[:, None] * get(ifx, icy)
        v = v + ((1.0 - dx) * dy)[:, None] * get(icx, ify)
        out[b] = torch.where(ok[:, None], v, zero)
    return out.numpy()


# --------------------------------------------------------------------------
# row D : get_img_points  (models/model_normalization.py:241-251)
# --------------------------------------------------------------------------
def get_img_points(sample_pc: np.ndarray, trans_mat_right: np.ndarray) -> np.ndarray:
    """[B,N,3] x [B,4,3] -> [B,N,2] pixel coords (x=col, y=row), clamped to
    [0,136].  homo = [x,y,z,1]; p = homo @ T; xy = p[:2]/p[2];
    min(136, max(0, xy)).  Summation order pinned here (and in the HIP kernel)
    as ((x*T0 + y*T1) + z*T2) + T3, float32, no FMA.  A NaN quotient
    (0/0) is undefined by the reference (Appendix C #16); this build defines it
    as 'outside the image': NaN is propagated so the resampler's range test
    fails and the features are zero."""
    pc = np.asarray(sample_pc, np.float32)
    T = np.asarray(trans_mat_right, np.float32)
    x, y, z = pc[..., 0:1], pc[..., 1:2], pc[..., 2:3]
    p = ((x * T[:, None, 0, :] + y * T[:, None, 1, :]).astype(np.float32)
         + z * T[:, None, 2, :]).astype(np.float32)
    p = (p + T[:, None, 3, :]).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy = (p[..., :2] / p[..., 2:3]).astype(np.float32)
    # np.maximum/minimum propagate NaN (the documented choice above)
    return np.minimum(F32(136.0), np.maximum(F32(0.0), xy)).astype(np.float32)


# --------------------------------------------------------------------------
# row F : tf.contrib.resampler.resampler  (TF-1.10 resampler_ops.cc, CPU functor)
# --------------------------------------------------------------------------
def resampler(data: np.ndarray, warp: np.ndarray) -> np.ndarray:
    """data [B,H,W,C], warp [B,N,2] (x,y) in pixels -> [B,N,C].
    if x>-1 and y>-1 and x<W and y<H:
        fx=floor(x); cx=fx+1; dx=cx-x (same for y)
        out = dx*dy*D(fx,fy) + (1-dx)(1-dy)*D(cx,cy) + dx(1-dy)*D(fx,cy) + (1-dx)dy*D(cx,fy)
    with D = 0 outside the image; else 0.  Sum order ((a+b)+c)+d as in the
    TF functor (img_fxfy + img_cxcy + img_fxcy + img_cxfy)."""
    data = np.asarray(data, np.float32)
    warp = np.asarray(warp, np.float32)
    B, H, W, C = data.shape
    N = warp.shape[1]
    out = np.zeros((B, N, C), np.float32)
    one = F32(1.0)
    for b in range(B):
        x, y = warp[b, :, 0], warp[b, :, 1]
        with np.errstate(invalid="ignore"):
            ok = (x > -1.0) & (y > -1.0) & (x < W) & (y < H)
        xs = np.where(ok, x, F32(0.0)).astype(np.float32)
        ys = np.where(ok, y, F32(0.0)).astype(np.float32)
        fx = np.floor(xs); fy = np.floor(ys)
        cx = fx + one; cy = fy + one
        dx = (cx - xs).astype(np.float32); dy = (cy - ys).astype(np.float32)
        ifx, ify, icx, icy = (a.astype(np.int64) for a in (fx, fy, cx, cy))

        def get(ix, iy):
            inb = (ix >= 0) & (iy >= 0) & (ix < W) & (iy < H)
            v = data[b, np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)]
            return np.where(inb[:, None], v, F32(0.0)).astype(np.float32)

        w_ff = (dx * dy).astype(np.float32)[:, None]
        w_cc = ((one - dx) * (one - dy)).astype(np.float32)[:, None]
        w_fc = (dx * (one - dy)).astype(np.float32)[:, None]
        w_cf = ((one - dx) * dy).astype(np.float32)[:, None]
        v = (w_ff * get(ifx, ify)).astype(np.float32)
        v = (v + w_cc * get(icx, icy)).astype(np.float32)
        v = (v + w_fc * get(ifx, icy)).astype(np.float32)
        v = (v + w_cf * get(icx, ify)).astype(np.float32)
        out[b] = np.where(ok[:, None], v, F32(0.0))
    return out


# --------------------------------------------------------------------------
# rows B, C : slim VGG-16  (models/CNN/vgg.py:187-217)
# --------------------------------------------------------------------------
def conv2d_numpy(x: np.ndarray, w: np.ndarray, b: np.ndarray, padding: str, relu: bool,
                 dtype=np.float32) -> np.ndarray:
    """Reference-of-the-oracle: im2col + matmul, NHWC x HWIO, stride 1.
    Slow; used on small shapes to validate conv2d()."""
    x = np.asarray(x, dtype); w = np.asarray(w, dtype); b = np.asarray(b, dtype)
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C
    if padding == "SAME":
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        xp = np.zeros((B, H + kh - 1, W + kw - 1, C), dtype)
        xp[:, ph:ph + H, pw:pw + W] = x
        Ho, Wo = H, W
    else:
        xp = x
        Ho, Wo = H - kh + 1, W - kw + 1
    cols = np.empty((B, Ho, Wo, kh, kw, C), dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + Ho, j:j + Wo, :]
    y = cols.reshape(B * Ho * Wo, kh * kw * C) @ w.reshape(kh * kw * C, co) + b
    y = y.reshape(B, Ho, Wo, co)
    return np.maximum(y, 0) if relu else y


def conv2d(x: np.ndarray, w: np.ndarray, b: np.ndarray, padding: str = "SAME", relu: bool = True,
           dtype=np.float32) -> np.ndarray:
    """slim.conv2d / tf_util.conv2d (utils/tf_util.py:169-183): NHWC x HWIO
    stride-1 conv + bias (+ReLU).  Executed with torch-CPU's conv
    (validated against conv2d_numpy in tests)."""
    import torch
    import torch.nn.functional as Fnn
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    xt = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype))).permute(3, 2, 0, 1).contiguous()
    bt = torch.from_numpy(np.asarray(b, dtype))
    kh, kw = w.shape[0], w.shape[1]
    pad = ((kh - 1) // 2, (kw - 1) // 2) if padding == "SAME" else 0
    y = Fnn.conv2d(xt.to(tdt), wt.to(tdt), bt.to(tdt), stride=1, padding=pad)
    if relu:
        y = torch.relu(y)
    return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())


def max_pool_2x2(x: np.ndarray) -> np.ndarray:
    """slim.max_pool2d [2,2] stride 2 VALID (models/CNN/vgg.py:188)."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    v = x[:, :Ho * 2, :Wo * 2].reshape(B, Ho, 2, Wo, 2, C)
    return v.max(axis=(2, 4))


def vgg16(img224: np.ndarray, weights: Dict[str, np.ndarray], dtype=np.float32
          ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """vgg_16(num_classes=1024, is_training=False, spatial_squeeze=False)
    (call: models/model_normalization.py:76; arch: models/CNN/vgg.py:187-214).
    Returns (embedding [B,1024], end_points with every conv output keyed
    'vgg_16/convX/convX_Y').  Dropout is inactive (is_training=False)."""
    net = np.asarray(img224, dtype)
    end_points: Dict[str, np.ndarray] = {}
    for scope, n, _ in VGG_CFG:
        for j in range(1, n + 1):
            nm = "vgg_16/%s/%s_%d" % (scope, scope, j)
            net = conv2d(net, weights[nm + "/weights"], weights[nm + "/biases"], "SAME", True, dtype)
            end_points[nm] = net
        net = max_pool_2x2(net)
        end_points["vgg_16/pool%s" % scope[-1]] = net
    net = conv2d(net, weights["vgg_16/fc6/weights"], weights["vgg_16/fc6/biases"], "VALID", True, dtype)
    end_points["vgg_16/fc6"] = net
    net = conv2d(net, weights["vgg_16/fc7/weights"], weights["vgg_16/fc7/biases"], "SAME", True, dtype)
    end_points["vgg_16/fc7"] = net
    net = conv2d(net, weights["vgg_16/fc8/weights"], weights["vgg_16/fc8/biases"], "SAME", False, dtype)
    end_points["vgg_16/fc8"] = net
    return net.reshape(net.shape[0], -1), end_points


# --------------------------------------------------------------------------
# row G : point MLPs  (models/sdfnet.py:69-92, 171-190; utils/tf_util.py:119-184)
# --------------------------------------------------------------------------
def _pointwise(x: np.ndarray, w: np.ndarray, b: np.ndarray, relu: bool, dtype) -> np.ndarray:
    """tf_util.conv2d with a [1,1] kernel == per-point x @ W[0,0] + b (+ReLU)."""
    y = np.asarray(x, dtype) @ np.asarray(w, dtype)[0, 0] + np.asarray(b, dtype)
    return np.maximum(y, 0) if relu else y


def get_sdf_basic2(src_pc: np.ndarray, globalfeats: np.ndarray, weights: Dict[str, np.ndarray],
                   scope: str = "sdfprediction", dtype=np.float32) -> np.ndarray:
    """models/sdfnet.py:69-92.  [B,N,3],[B,1024] -> [B,N,1].
    concat order [point512, global1024] (:78-82)."""
    W = lambda n: weights["%s/%s/weights" % (scope, n)]
    Bv = lambda n: weights["%s/%s/biases" % (scope, n)]
    B, N, _ = src_pc.shape
    net = _pointwise(src_pc, W("fold1/conv1"), Bv("fold1/conv1"), True, dtype)
    net = _pointwise(net, W("fold1/conv2"), Bv("fold1/conv2"), True, dtype)
    net = _pointwise(net, W("fold1/conv3"), Bv("fold1/conv3"), True, dtype)
    g = np.broadcast_to(np.asarray(globalfeats, dtype).reshape(B, 1, -1), (B, N, globalfeats.reshape(B, -1).shape[1]))
    net = np.concatenate([net, g], axis=2)
    net = _pointwise(net, W("fold2/conv1"), Bv("fold2/conv1"), True, dtype)
    net = _pointwise(net, W("fold2/conv2"), Bv("fold2/conv2"), True, dtype)
    pred = _pointwise(net, W("fold2/conv5"), Bv("fold2/conv5"), False, dtype)
    return pred.reshape(B, -1, 1)


def get_sdf_basic2_imgfeat_twostream(src_pc: np.ndarray, point_feat: np.ndarray,
                                     weights: Dict[str, np.ndarray],
                                     scope: str = "sdfprediction_imgfeat", dtype=np.float32) -> np.ndarray:
    """models/sdfnet.py:171-190.  [B,N,3],[B,N,1,1472] -> [B,N,1].
    concat order [point512, feat1472] (:180)."""
    W = lambda n: weights["%s/%s/weights" % (scope, n)]
    Bv = lambda n: weights["%s/%s/biases" % (scope, n)]
    B, N, _ = src_pc.shape
    net = _pointwise(src_pc, W("fold1/conv1"), Bv("fold1/conv1"), True, dtype)
    net = _pointwise(net, W("fold1/conv2"), Bv("fold1/conv2"), True, dtype)
    net = _pointwise(net, W("fold1/conv3"), Bv("fold1/conv3"), True, dtype)
    net = np.concatenate([net, np.asarray(point_feat, dtype).reshape(B, N, -1)], axis=2)
    net = _pointwise(net, W("fold2/conv1"), Bv("fold2/conv1"), True, dtype)
    net = _pointwise(net, W("fold2/conv2"), Bv("fold2/conv2"), True, dtype)
    pred = _pointwise(net, W("fold2/conv5"), Bv("fold2/conv5"), False, dtype)
    return pred.reshape(B, -1, 1)


# --------------------------------------------------------------------------
# the full graph : get_model  (models/model_normalization.py:47-221, twostream regression)
# --------------------------------------------------------------------------
def upsampled_taps(vgg_end_points: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """5 x resize_bilinear(tap, (137,137))  (models/model_normalization.py:171-183)."""
    maps = []
    for nm in TAP_NAMES:
        tap = vgg_end_points["vgg_16/%s/%s" % (nm[:5], nm)]
        maps.append(resize_bilinear_legacy(np.asarray(tap, np.float32), IMG_H, IMG_W))
    return maps


def gather_point_feat(maps: Sequence[np.ndarray], sample_img_points: np.ndarray) -> np.ndarray:
    """5 x resampler + concat(axis=2) + expand_dims (models/model_normalization.py:172-190)
    -> [B,N,1,1472]."""
    feats = [resampler(m, sample_img_points) for m in maps]
    return np.concatenate(feats, axis=2)[:, :, None, :]


def encode(imgs: np.ndarray, weights: Dict[str, np.ndarray], dtype=np.float32):
    """Rows A-C and E: resize 137->224, VGG-16, 5 up-sampled tap maps.
    Returns (resized_img, embedding [B,1024], maps[5] float32)."""
    imgs = np.asarray(imgs, np.float32)
    if imgs.shape[1] != VGG_SIZE or imgs.shape[2] != VGG_SIZE:      # :65-72
        resized = resize_bilinear_legacy(imgs, VGG_SIZE, VGG_SIZE)
    else:
        resized = imgs
    emb, vgg_eps = vgg16(resized, weights, dtype)
    maps = upsampled_taps({k: np.asarray(v, np.float32) for k, v in vgg_eps.items()})
    return resized, emb, maps, vgg_eps


def get_model(feed: Dict[str, np.ndarray], weights: Dict[str, np.ndarray], dtype=np.float32,
              tanh: bool = False) -> Dict[str, np.ndarray]:
    """The --img_feat_twostream regression branch of get_model
    (models/model_normalization.py:47-221, branch :169-206) as one eager
    function.  feed keys as placeholder_inputs (:14-35).  Returns the
    end_points dict with the reference's keys."""
    ep: Dict[str, np.ndarray] = {}
    imgs = np.asarray(feed["imgs"], np.float32)
    ep["ref_pc"] = feed.get("pc")
    ep["ref_sdf"] = feed.get("sdf")
    ep["ref_img"] = imgs                                           # :62 (un-resized)
    resized, emb, maps, vgg_eps = encode(imgs, weights, dtype)
    ep["resized_ref_img"] = resized                                # :73
    ep["img_embedding"] = np.asarray(emb, np.float32)              # :78
    ep["ref_feats_embedding_cnn"] = ep["img_embedding"]
    ep["vgg_end_points"] = vgg_eps
    xy = get_img_points(feed["sample_pc"], feed["trans_mat"])      # :170
    ep["sample_img_points"] = xy
    feat = gather_point_feat(maps, xy)                             # :171-190
    ep["point_img_feat"] = feat
    g = get_sdf_basic2(np.asarray(feed["sample_pc_rot"], np.float32), emb, weights, dtype=dtype)      # :194-197
    l = get_sdf_basic2_imgfeat_twostream(np.asarray(feed["sample_pc_rot"], np.float32), feat, weights,
                                         dtype=dtype)                                                # :199-202
    pred = g + l                                                   # :204
    if tanh:
        pred = np.tanh(pred)                                       # :214-215
    ep["pred_sdf_value_global"] = g
    ep["pred_sdf_value_local"] = l
    ep["pred_sdf"] = pred
    return ep


# --------------------------------------------------------------------------
# row K : get_loss (models/model_normalization.py:254-300), regression branch
# --------------------------------------------------------------------------
def get_loss(pred_sdf: np.ndarray, gt_sdf: np.ndarray, weights: Dict[str, np.ndarray] = None,
             sdf_weight: float = 10.0, mask_weight: float = 4.0, wd: float = 1e-5) -> Dict[str, float]:
    pred = np.asarray(pred_sdf, np.float32); gt = np.asarray(gt_sdf, np.float32)
    losses = {}
    losses["accuracy"] = float(np.mean(((gt > 0) == (pred > 0)).astype(np.float32)))
    wmask = (gt <= F32(0.01)).astype(np.float32) * F32(mask_weight) + (gt > F32(0.01)).astype(np.float32)
    sdf_loss = np.mean(np.abs(gt * F32(sdf_weight) - pred) * wmask)
    losses["sdf_loss_realvalue"] = float(np.mean(np.abs(gt - pred / F32(sdf_weight))))
    losses["sdf_loss"] = float(sdf_loss * 1000.0)
    reg = 0.0
    if weights is not None:
        for k, v in weights.items():
            if k.endswith("/weights"):
                reg += wd * 0.5 * float(np.sum(np.asarray(v, np.float64) ** 2))
    losses["regularization"] = reg
    losses["overall_loss"] = losses["sdf_loss"] + reg
    return losses


# --------------------------------------------------------------------------
# row J : dense grid construction + chunking  (test/create_sdf.py:69-77, 241-285)
# --------------------------------------------------------------------------
def split_plan(sdf_res: int, twostream: bool = True) -> Tuple[int, int, int, int]:
    """(TOTAL_POINTS, SPLIT_SIZE, NUM_SAMPLE_POINTS, pad) -- test/create_sdf.py:69-77."""
    resolution = sdf_res + 1
    total = resolution ** 3
    split = int(np.ceil(total / (214669.0 if twostream else 274625.0)))
    nsp = int(np.ceil(total / split))
    return total, split, nsp, split * nsp - total


def grid_points(sdf_params: Sequence[float], sdf_res: int) -> np.ndarray:
    """[(R+1)^3, 3] float32 in the flat (iz,iy,ix) order -- test/create_sdf.py:246-256:
    linspace in float64, meshgrid(z_,y_,x_,'ij'), concat (x,y,z), cast float32."""
    res = sdf_res + 1
    # float64 explicitly: what numpy 1.x linspace computes for int / float64 scalars (demo/demo.py:278 passes
    # ints).  For float32 scalars numpy 1.x rounds delta = stop - start and step = delta / div to float32 and only
    # then multiplies the float64 arange (grid_points_numpy1_float32 below): identical for the +-1 box and any
    # dyadic box, up to R * ulp32(step) / 2 (~1e-7 of the box) apart otherwise.  numpy >= 2 (NEP 50) computes
    # everything in float32 for float32 inputs -- neither is this function.
    p = np.asarray(sdf_params, dtype=np.float64)
    x_ = np.linspace(p[0], p[3], num=res)
    y_ = np.linspace(p[1], p[4], num=res)
    z_ = np.linspace(p[2], p[5], num=res)
    z, y, x = np.meshgrid(z_, y_, x_, indexing="ij")
    return np.stack((x, y, z), axis=3).astype(np.float32).reshape(-1, 3)


def grid_points_numpy1_float32(sdf_params: Sequence[float], sdf_res: int) -> np.ndarray:
    """the grid numpy 1.x builds when sdf_params are FLOAT32 scalars (numpy/core/function_base.py linspace, 1.14):
    start, stop stay float32, delta = stop - start and step = delta / div are float32, y = arange(float64) * step
    + start, y[-1] = stop.  Restated with explicit casts so that it does not depend on the installed numpy."""
    res = sdf_res + 1
    p = np.asarray(sdf_params, dtype=np.float32)
    axes = []
    for a in range(3):
        start, stop = np.float32(p[a]), np.float32(p[a + 3])
        step = np.float32(np.float32(stop - start) / np.float32(res - 1))
        y = np.arange(res, dtype=np.float64) * np.float64(step) + np.float64(start)
        y[-1] = np.float64(stop)
        axes.append(y)
    z, y, x = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
    return np.stack((x, y, z), axis=3).astype(np.float32).reshape(-1, 3)


def dense_grid_sdf(img: np.ndarray, trans_mat: np.ndarray, sdf_params: Sequence[float], sdf_res: int,
                   weights: Dict[str, np.ndarray], per_split_encode: bool = True,
                   max_splits: int = None, dtype=np.float32) -> np.ndarray:
    """test_one_epoch for one image (test/create_sdf.py:224-285): build grid,
    pad with (0,0,0), run get_model per split (the WHOLE graph, VGG included,
    is re-run on each split when per_split_encode -- the reference's
    structure), concatenate, drop padding, divide by SDF_WEIGHT.
    max_splits bounds the work for timing samples."""
    total, split, nsp, pad = split_plan(sdf_res)
    pts = np.concatenate([grid_points(sdf_params, sdf_res), np.zeros((pad, 3), np.float32)], 0)
    pts = pts.reshape(split, 1, nsp, 3)
    out = np.zeros((split, 1, nsp, 1), np.float64)                 # :260 (float64 accumulator)
    cached = None
    n_run = split if max_splits is None else min(split, max_splits)
    for sp in range(n_run):
        feed = {"imgs": img, "sample_pc": pts[sp], "sample_pc_rot": pts[sp], "trans_mat": trans_mat}
        if per_split_encode or cached is None:
            ep = get_model(feed, weights, dtype)
            cached = ep
        out[sp] = ep["pred_sdf"]
    res = out.reshape(1, -1, 1)[:, :total, :] / SDF_WEIGHT
    return res[0, :, 0]


def to_binary(res: int, pos: Sequence[float], sdf_vals: np.ndarray) -> bytes:
    """The .dist wire format -- test/create_sdf.py:292-303: int32 -res, res,
    res; 6 x float64 bbox; (res+1)^3 float32, x fastest."""
    vals = np.asarray(sdf_vals, np.float32).ravel()
    return (struct.pack("i", -res) + struct.pack("i", res) + struct.pack("i", res)
            + struct.pack("d" * len(pos), *[float(p) for p in pos]) + vals.astype("<f4").tobytes())


# --------------------------------------------------------------------------
# camera convention (preprocessing/create_img_h5.py:14-63, 156-201)
# --------------------------------------------------------------------------
def blender_proj(az: float, el: float, distance_ratio: float, img_w: int = 137, img_h: int = 137):
    """getBlenderProj restated: returns (K [3,3], RT [3,4]) float64."""
    F_MM, SENSOR, CAM_MAX_DIST = 35.0, 32.0, 1.75
    cam_rot = np.asarray([[1.910685676922942e-15, 4.371138828673793e-08, 1.0],
                          [1.0, -4.371138828673793e-08, -0.0],
                          [4.371138828673793e-08, 1.0, -4.371138828673793e-08]])
    f_u = F_MM * img_w / SENSOR
    f_v = F_MM * img_h / SENSOR
    K = np.array([[f_u, 0.0, img_w / 2.0], [0.0, f_v, img_h / 2.0], [0.0, 0.0, 1.0]])
    sa, ca = np.sin(np.radians(-az)), np.cos(np.radians(-az))
    se, ce = np.sin(np.radians(-el)), np.cos(np.radians(-el))
    R_world2obj = np.array([[ca * ce, -sa, ca * se], [sa * ce, ca, sa * se], [-se, 0.0, ce]]).T
    R_obj2cam = cam_rot.T
    R_world2cam = R_obj2cam @ R_world2obj
    cam_location = np.array([[distance_ratio * CAM_MAX_DIST], [0.0], [0.0]])
    T_world2cam = -1.0 * R_obj2cam @ cam_location
    R_camfix = np.diag([1.0, -1.0, -1.0])
    R_world2cam = R_camfix @ R_world2cam
    T_world2cam = R_camfix @ T_world2cam
    return K, np.hstack((R_world2cam, T_world2cam))


def synth_trans_mat(az: float, el: float, distance_ratio: float = 0.8) -> np.ndarray:
    """A [4,3] right-multiply projection in the stored convention
    (preprocessing/create_img_h5.py:184-185: trans_mat = (K.RT.[rot|0;0 1]).T
    with rot = rot90y, object normalisation omitted = identity)."""
    K, RT = blender_proj(az, el, distance_ratio)
    rot90y = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]], dtype=np.float64)
    rot4 = np.eye(4); rot4[:3, :3] = rot90y
    M = K @ RT @ rot4                        # [3,4]
    return M.T.astype(np.float32)            # [4,3]


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synth_inputs(seed: int = 0, batch: int = 1, n_points: int = 2048) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    imgs = rng.random((batch, IMG_H, IMG_W, 3), dtype=np.float32)
    pts = (rng.random((batch, n_points, 3), dtype=np.float32) * F32(2.0) - F32(1.0)).astype(np.float32)
    tm = np.repeat(DEMO_TRANS_MAT, batch, axis=0)
    return {"imgs": imgs, "sample_pc": pts, "sample_pc_rot": pts.copy(), "trans_mat": tm,
            "sdf_params": np.tile(np.array([[-1, -1, -1, 1, 1, 1]], np.float32), (batch, 1))}


def load_demo_image(path: str) -> np.ndarray:
    """demo/demo.py:262-264: cv2.imread(IMREAD_UNCHANGED)[:,:,:3] (=> BGR,
    alpha dropped) / 255.  PIL reads RGBA; reverse RGB to mimic cv2."""
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
    bgr = a[:, :, [2, 1, 0]]
    return (bgr.astype(np.float32) / F32(255.0))[None]
