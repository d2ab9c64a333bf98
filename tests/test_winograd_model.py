"""tools/winograd_model.py (VERDICT r4 #3d: F(2x2, 3x3) Winograd on the two-term f16 split, emulated before any kernel is
written) stays runnable: on a small layer the modelled Winograd convolution is the convolution (float64 reference) to the
split's accuracy, with exact and with float32 accumulation.  (-m "not gpu")"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_winograd_on_the_split_is_the_convolution():
    import winograd_model as WM
    from oracle import disn_oracle as O
    rng = np.random.default_rng(0)
    x = np.maximum(rng.standard_normal((16, 20, 64)), 0) * np.exp(rng.standard_normal(64))      # post-ReLU, channel gains
    w = rng.standard_normal((3, 3, 64, 32)) * np.sqrt(2.0 / (9 * 64))
    ref = O.conv2d(x[None], w, np.zeros(32), relu=False, dtype=np.float64)[0]
    sc = np.abs(ref).max()
    for f32 in (False, True):
        ed = np.abs(WM.direct(x, w, f32) - ref).max() / sc
        ew = np.abs(WM.winograd(x, w, f32) - ref).max() / sc
        print("fp32 accumulate %s: direct %.2e winograd %.2e of the output maximum" % (f32, ed, ew))
        assert ed <= 1e-6 and ew <= 1e-6
    # the transforms themselves: exact arithmetic reproduces the convolution
    U = np.einsum("ir,rscn,js->ijcn", WM.G, w, WM.G)
    xp = np.zeros((18, 22, 64)); xp[1:-1, 1:-1] = x
    d = np.stack([np.stack([xp[r:r + 16:2, c:c + 20:2] for c in range(4)], axis=2) for r in range(4)], axis=2)
    V = np.einsum("ir,abrsc,js->abijc", WM.BT, d, WM.BT)
    M = np.einsum("abijc,ijcn->abijn", V, U)
    Y = np.einsum("pi,abijn,qj->abpqn", WM.AT, M, WM.AT).transpose(0, 2, 1, 3, 4).reshape(16, 20, 32)
    assert np.abs(Y - ref).max() <= 1e-12 * sc
