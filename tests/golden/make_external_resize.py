"""Hand derivation, by exact rational arithmetic, of legacy-bilinear resize vectors (tf.image.resize_bilinear,
align_corners=False, no half-pixel centres: tensorflow/core/kernels/resize_bilinear_op.cc of TF 1.x, the kernel the
reference calls at models/model_normalization.py:72,171-183) -> the `resize_exact_*` entries of external_kat.json.

Nothing of this repo's oracle or kernels is used.  The formula as the TF kernel states it, per output index i of an axis:
    scale = (float) in_size / (float) out_size            one float32 division
    src   = (float) i * scale                             one float32 multiplication
    lo = floor(src), hi = min(lo + 1, in_size - 1), frac = src - lo          (exact in float32)
    out   = top + (bottom - top) * y_frac,  top = tl + (tr - tl) * x_frac,  bottom likewise
The two ROUNDED operations (scale, src) are evaluated with numpy float32 scalars -- that is the formula's statement, one
IEEE operation each; everything after them is evaluated with fractions.Fraction, and the inputs are chosen (0 or a power
of two, alternating) so that every product and sum of the lerps is EXACT in float32: the script asserts that each
expected value is a float32 number, so no rounding-order assumption enters the vector.

    python tests/golden/make_external_resize.py   (rewrites the resize_exact_* entries of external_kat.json)
"""
import json
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def axis_table(n_in, n_out):
    scale = np.float32(n_in) / np.float32(n_out)
    tab = []
    for i in range(n_out):
        src = np.float32(i) * scale
        lo = int(np.floor(src))
        hi = min(lo + 1, n_in - 1)
        frac = Fraction(float(src)) - lo            # float32 -> exact rational
        tab.append((lo, hi, frac))
    return tab


def exact32(fr):
    v = np.float32(float(fr))
    assert Fraction(float(v)) == fr, "not exact in float32: %s" % fr
    return float(v)


def row_case(n_in, n_out):
    """1 x n_in x 1 input (values 0 / 2^k alternating), resized along x only"""
    vals = [0.0 if j % 2 == 0 else float(2 ** (1 + (j // 2) % 5)) for j in range(n_in)]
    out = []
    for lo, hi, fr in axis_table(n_in, n_out):
        a, b = Fraction(vals[lo]), Fraction(vals[hi])
        out.append(exact32(a + (b - a) * fr))
    return {"in_shape": [1, 1, n_in, 1], "in": vals, "out_hw": [1, n_out], "out": out}


def grid_case(h_in, w_in, h_out, w_out):
    """2-D: values 2^(y) on even columns, 0 on odd ones (every lerp has one zero operand or equal operands)"""
    img = [[0.0 if x % 2 else float(2 ** (y % 4)) for x in range(w_in)] for y in range(h_in)]
    ty, tx = axis_table(h_in, h_out), axis_table(w_in, w_out)
    out = []
    for ylo, yhi, yf in ty:
        for xlo, xhi, xf in tx:
            tl, tr, bl, br = (Fraction(img[ylo][xlo]), Fraction(img[ylo][xhi]), Fraction(img[yhi][xlo]), Fraction(img[yhi][xhi]))
            top = tl + (tr - tl) * xf
            exact32(top)
            bot = bl + (br - bl) * xf
            exact32(bot)
            out.append(exact32(top + (bot - top) * yf))
    return {"in_shape": [1, h_in, w_in, 1], "in": [v for r in img for v in r], "out_hw": [h_out, w_out], "out": out}


def main():
    p = os.path.join(HERE, "external_kat.json")
    k = json.load(open(p))
    k["resize_exact_row_137_224"] = row_case(137, 224)      # row A: the 137 -> 224 image resize
    k["resize_exact_row_14_137"] = row_case(14, 137)        # row E: the coarsest tap, 14 -> 137
    k["resize_exact_row_224_137"] = row_case(224, 137)      # row E: the finest tap, 224 -> 137 (a down-sampling)
    k["resize_exact_grid_4x6_8x12"] = grid_case(4, 6, 8, 12)      # scale 1/2: fractions 0 and 1/2
    k["resize_exact_grid_6x4_3x2"] = grid_case(6, 4, 3, 2)        # scale 2: pure sub-sampling
    k["_about_resize_exact"] = ("resize_exact_*: derived by tests/golden/make_external_resize.py with exact rational arithmetic "
                                "from the TF-1.x legacy bilinear formula (scale and source coordinate: one float32 operation "
                                "each; the lerps in fractions.Fraction on inputs that make every intermediate exact in "
                                "float32). They replace reliance on resize_down, which was quoted from memory of "
                                "TensorFlow's image_ops_test.py (kept: its values equal the formula's sub-sampling result).")
    json.dump(k, open(p, "w"), indent=1)
    print("rows:", {n: len(v["out"]) for n, v in k.items() if n.startswith("resize_exact")})


if __name__ == "__main__":
    main()
