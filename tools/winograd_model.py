"""VERDICT r4 #3d, "emulate first, then decide": F(2x2, 3x3) Winograd for the 3x3 convolutions of the VGG stack
(models/CNN/vgg.py:187-196) ON the two-term f16 operand split -- 16 transformed multiplications per 2x2 outputs instead of
36, i.e. 2.25x fewer MFMAs, the only lever that moves the 833 TFLOP/s two-term ceiling.  CPU model, no GPU:

  direct   : x, w split into h + l (22 bits, power-of-two scales per image / per output channel), l.l dropped -- what
             conv_h2 / conv_h2w compute -- with (a) exact accumulation (the split alone) and (b) float32 accumulation
  winograd : V = B^T d B per 4x4 input tile in float32 (as the kernel's loader would), U = G g G^T at pack time in float64,
             both split the same way (V per image, U per output channel AND tile position), the 16 channel contractions
             with exact / float32 accumulation, Y = A^T M A in float32

against the float64 convolution, error relative to the layer's output maximum (the taps' bar: <= 2e-6; build only if
winograd stays <= 3e-6 on He AND trained-like statistics).  Inputs: the float64 oracle's activations of the demo-size
random image on He weights and on trained-like weights (sigma 2, equalised as the engine uploads them).

    python tools/winograd_model.py            (~2 minutes)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import disn_oracle as O   # noqa: E402

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def pow2(amax, target):
    amax = np.asarray(amax, np.float64)
    e = np.floor(np.log2(np.where(amax > 0, amax, 1.0)))
    return np.where(amax > 0, 2.0 ** (target - e), 1.0)


def split(v, s):
    x = np.asarray(v, np.float64) * s
    h = x.astype(np.float16).astype(np.float64)
    l = (x - h).astype(np.float16).astype(np.float64)
    return h, l


def contract(ah, al, wh, wl, f32):
    """sum_k a[.., k] w[k, n] with the l.l term dropped; f32: float32 GEMMs (fp32 accumulation, BLAS order)"""
    if f32:
        m = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float64)
        return m(al, wh) + m(ah, wl) + m(ah, wh)
    return al @ wh + ah @ wl + ah @ wh


def direct(x, w, f32):
    """x [H, W, C] float64, w [3, 3, C, N]: SAME conv through the split, as an im2col contraction"""
    H, W, C = x.shape
    N = w.shape[3]
    sa = pow2(np.abs(x).max(), 14)
    sw = pow2(np.abs(w).reshape(-1, N).max(0), 13)
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    cols = np.concatenate([xp[r:r + H, c:c + W] for r in range(3) for c in range(3)], axis=2).reshape(H * W, 9 * C)
    ah, al = split(cols, sa)
    wh, wl = split(w.reshape(9 * C, N), sw[None, :])
    return (contract(ah, al, wh, wl, f32) / (sa * sw[None, :])).reshape(H, W, N)


def winograd(x, w, f32):
    H, W, C = x.shape
    N = w.shape[3]
    th, tw = H // 2, W // 2
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    # input tiles d [th, tw, 4, 4, C] (stride 2), V = B^T d B in float32
    d = np.stack([np.stack([xp[r:r + 2 * th:2, c:c + 2 * tw:2] for c in range(4)], axis=2) for r in range(4)], axis=2)
    d32 = d.astype(np.float32)
    V = np.einsum("ir,abrsc,js->abijc", BT.astype(np.float32), d32, BT.astype(np.float32)).astype(np.float64)
    U = np.einsum("ir,rscn,js->ijcn", G, w, G)                         # [4, 4, C, N] at pack time (float64)
    sa = pow2(np.abs(V).max(), 14)                                     # one scale per image
    su = pow2(np.abs(U).max(axis=2), 13)                               # per (tile position, output channel)
    Vh, Vl = split(V, sa)
    M = np.zeros((th, tw, 4, 4, N))
    for i in range(4):
        for j in range(4):
            uh, ul = split(U[i, j], su[i, j][None, :])
            a_h, a_l = Vh[:, :, i, j].reshape(-1, C), Vl[:, :, i, j].reshape(-1, C)
            M[:, :, i, j] = (contract(a_h, a_l, uh, ul, f32) / (sa * su[i, j][None, :])).reshape(th, tw, N)
    M32 = M.astype(np.float32)
    Y = np.einsum("pi,abijn,qj->abpqn", AT.astype(np.float32), M32, AT.astype(np.float32)).astype(np.float64)
    return Y.transpose(0, 2, 1, 3, 4).reshape(H, W, N)


def main():
    from disn_amd.weights import WeightStore
    img = O.synth_inputs(5, 1, 8)["imgs"]
    layers = [("conv2/conv2_2", "conv2/conv2_1", 112), ("conv3/conv3_2", "conv3/conv3_1", 56), ("conv4/conv4_2", "conv4/conv4_1", 28),
              ("conv5/conv5_2", "conv5/conv5_1", 14)]
    for label, W in (("he", O.init_weights(3, "he")),
                     ("trained-like sigma 2 (equalised)", WeightStore(O.trained_like_weights(21, sigma=2.0)).equalised()[0].arrays)):
        _, _, _, eps = O.encode(img, W, np.float64)
        for name, prev, hw in layers:
            x = np.asarray(eps["vgg_16/" + prev], np.float64)[0]
            if x.shape[0] != hw:
                x = O.max_pool_2x2(x[None])[0]
            w = np.asarray(W["vgg_16/%s/weights" % name], np.float64)
            if hw > 56:                                     # a 56 x 56 crop keeps the im2col of the big layers small
                x = x[:56, :56]
            ref = O.conv2d(x[None], w, np.zeros(w.shape[3]), relu=False, dtype=np.float64)[0]
            sc = np.abs(ref).max()
            row = []
            for f32 in (False, True):
                ed = np.abs(direct(x, w, f32) - ref).max() / sc
                ew = np.abs(winograd(x, w, f32) - ref).max() / sc
                row.append((ed, ew))
            print("%-34s %-14s split only: direct %.2e winograd %.2e | fp32 accumulate: direct %.2e winograd %.2e" % (
                label, name, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)


if __name__ == "__main__":
    main()
