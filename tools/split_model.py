"""CPU model of the two-term f16 operand split on trained-like statistics (VERDICT r3 #2c): the whole forward in
float64 EXCEPT that every MFMA operand is replaced by h + l, h = f16(x s), l = f16(x s - h), s = the power of two that
puts the tensor's (per image: activations; per tensor: weights) largest magnitude in [2^14, 2^15) / [2^13, 2^14), and
the l.l product is dropped -- i.e. the GPU path's arithmetic with exact accumulation.  What it prints is the part of
|gpu - f64| that the SPLIT is responsible for (the fp32 accumulation error comes on top, ~3e-6 on He weights).

    python tools/split_model.py [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import disn_oracle as O   # noqa: E402


def pow2_scale(amax, target):
    if not amax > 0:
        return 1.0
    e = int(np.floor(np.log2(amax)))
    return 2.0 ** (target - e)


PER_COLUMN = [False]      # weight scale per output channel (column) instead of per tensor


def wscale(w):
    """w [..., K, N] -> scale broadcastable to w"""
    if PER_COLUMN[0]:
        m = np.abs(w).reshape(-1, w.shape[-1]).max(0)
        return np.array([pow2_scale(v, 13) for v in m])
    return pow2_scale(np.abs(w).max(), 13)


def split(x, s):
    v = np.asarray(x, np.float64) * s
    h = v.astype(np.float16).astype(np.float64)
    l = (v - h).astype(np.float16).astype(np.float64)
    return h, l


def conv_split(x, w, b, relu=True):
    """x [1,H,W,C] float64 (one image), 3x3 SAME"""
    sa, sw = pow2_scale(np.abs(x).max(), 14), wscale(w)
    ah, al = split(x, sa)
    wh, wl = split(w, sw)
    z = np.zeros(w.shape[3])
    y = (O.conv2d(al, wh, z, relu=False, dtype=np.float64) + O.conv2d(ah, wl, z, relu=False, dtype=np.float64)
         + O.conv2d(ah, wh, z, relu=False, dtype=np.float64)) / (sa * sw) + np.asarray(b, np.float64)
    return np.maximum(y, 0) if relu else y


def dense_split(x, w, b, relu=True, row_scale=False):
    """x [n,K] float64; per-image scale (row_scale: per-point scale, the fused kernels)"""
    w = np.asarray(w, np.float64)[0, 0]
    sw = wscale(w)
    if row_scale:
        sa = np.array([pow2_scale(m, 14) for m in np.abs(x).max(1)])[:, None]
    else:
        sa = pow2_scale(np.abs(x).max(), 14)
    ah, al = split(x, sa)
    wh, wl = split(w, sw)
    y = (al @ wh + ah @ wl + ah @ wh) / (sa * sw) + np.asarray(b, np.float64)
    return np.maximum(y, 0) if relu else y


def forward_split(feed, W, row_scale=False):
    f64 = np.float64
    img = O.resize_bilinear_legacy(feed["imgs"], 224, 224)
    x = np.asarray(img, f64)
    taps = {}
    first = True
    for scope, n, _ in O.VGG_CFG:
        for j in range(1, n + 1):
            nm = "vgg_16/%s/%s_%d" % (scope, scope, j)
            if first:      # conv1_1: fp32 FMA on the GPU (K = 27): exact here
                x = O.conv2d(x, W[nm + "/weights"], W[nm + "/biases"], relu=True, dtype=f64)
                first = False
            else:
                x = conv_split(x, np.asarray(W[nm + "/weights"], f64), W[nm + "/biases"])
            taps["%s_%d" % (scope, j)] = x
        x = O.max_pool_2x2(x)
    # fc head: fp32 FMA GEMV on the GPU: exact here
    v = x.reshape(1, -1)
    for nm, relu in (("fc6", True), ("fc7", True), ("fc8", False)):
        w = np.asarray(W["vgg_16/%s/weights" % nm], f64)
        v = v @ w.reshape(-1, w.shape[3]) + np.asarray(W["vgg_16/%s/biases" % nm], f64)
        v = np.maximum(v, 0) if relu else v
    emb = v
    maps = [O.resize_bilinear_legacy(np.asarray(taps[nm], np.float32), 137, 137) for nm in O.TAP_NAMES]
    xy = O.get_img_points(feed["sample_pc"], feed["trans_mat"])
    feat = O.gather_point_feat(maps, xy)[0, :, 0, :].astype(f64)
    pts = np.asarray(feed["sample_pc_rot"], f64)[0]
    out = 0.0
    for scope, extra in (("sdfprediction", None), ("sdfprediction_imgfeat", feat)):
        Wn = lambda n: W["%s/%s/weights" % (scope, n)]
        Bn = lambda n: W["%s/%s/biases" % (scope, n)]
        h = np.maximum(pts @ np.asarray(Wn("fold1/conv1"), f64)[0, 0] + np.asarray(Bn("fold1/conv1"), f64), 0)   # VALU
        h = dense_split(h, Wn("fold1/conv2"), Bn("fold1/conv2"), row_scale=row_scale)
        h = dense_split(h, Wn("fold1/conv3"), Bn("fold1/conv3"), row_scale=row_scale)
        w4 = np.asarray(Wn("fold2/conv1"), f64)
        if extra is None:   # global: the embedding block is a per-image bias (fp32 GEMV: exact here)
            bias = emb[0] @ w4[0, 0, 512:] + np.asarray(Bn("fold2/conv1"), f64)
            h = dense_split(h, w4[:, :, :512], bias, row_scale=row_scale)
        else:
            h = dense_split(np.concatenate([h, extra], 1), w4, Bn("fold2/conv1"), row_scale=row_scale)
        h = dense_split(h, Wn("fold2/conv2"), Bn("fold2/conv2"), row_scale=row_scale)
        out = out + h @ np.asarray(Wn("fold2/conv5"), f64)[0, 0] + np.asarray(Bn("fold2/conv5"), f64)
    return out[None], emb, taps


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    PER_COLUMN[0] = len(sys.argv) > 2 and sys.argv[2] == "percol"
    print("weight scale per %s" % ("output channel" if PER_COLUMN[0] else "tensor"))
    for label, W in (("he", O.init_weights(seed, "he")), ("trained-like", O.trained_like_weights(seed)),
                     ("trained-like, outliers 1e4", O.trained_like_weights(seed, outlier_gain=1e4)),
                     ("trained-like, sigma 2", O.trained_like_weights(seed, sigma=2.0))):
        feed = O.synth_inputs(seed, 1, 1024)
        ref = O.get_model(feed, W, dtype=np.float64)
        got, emb, taps = forward_split(feed, W)
        e_emb = np.abs(emb - ref["img_embedding"].astype(np.float64)).max()
        print("%-28s |pred| max %.3g  split-only error: pred %.3g  (embedding rel %.3g)" % (
            label, np.abs(ref["pred_sdf"]).max(), np.abs(got - ref["pred_sdf"]).max(),
            e_emb / np.abs(ref["img_embedding"]).max()), flush=True)
