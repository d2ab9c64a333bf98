"""Adds TensorFlow's own conv2d unit-test vectors to external_kat.json (tensorflow/python/kernel_tests/conv_ops_test.py,
Conv2DTest.testConv2D1x1Filter / testConv2D2x2Filter: inputs and filters are 1, 2, 3, ... in NHWC / HWIO memory order,
stride 1, VALID).  The expected values are the ones that file states; this script re-derives them with integer
arithmetic and refuses to write anything that differs -- what they pin is the ORIENTATION of slim's conv2d on the path
(models/CNN/vgg.py:187-196 -> tf.nn.conv2d: cross-correlation, NHWC activations, HWIO filters, channel-minor sums).
usage: python tests/golden/make_external_conv.py"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

QUOTED = {
    "conv2d_tf_1x1": {"in_shape": [1, 2, 3, 3], "filter_shape": [1, 1, 3, 3],
                      "out": [30, 36, 42, 66, 81, 96, 102, 126, 150, 138, 171, 204, 174, 216, 258, 210, 261, 312]},
    "conv2d_tf_2x2": {"in_shape": [1, 2, 3, 3], "filter_shape": [2, 2, 3, 3],
                      "out": [2271, 2367, 2463, 2901, 3033, 3165]},
}


def derive(in_shape, filter_shape):
    _, H, W, C = in_shape
    kh, kw, ci, co = filter_shape
    x = lambda y, xx, c: (y * W + xx) * C + c + 1
    w = lambda r, s, c, o: ((r * kw + s) * ci + c) * co + o + 1
    out = []
    for y in range(H - kh + 1):
        for xx in range(W - kw + 1):
            for o in range(co):
                out.append(sum(x(y + r, xx + s, c) * w(r, s, c, o) for r in range(kh) for s in range(kw) for c in range(ci)))
    return out


def main():
    p = os.path.join(HERE, "external_kat.json")
    k = json.load(open(p))
    for name, q in QUOTED.items():
        assert derive(q["in_shape"], q["filter_shape"]) == q["out"], name
        k[name] = dict(q, padding="VALID")
    # pooling_ops_test.py, PoolingTest._testMaxPoolValidPadding: 1 .. 27 as [1, 3, 3, 3], 2 x 2 windows, stride 2, VALID
    mp = {"in_shape": [1, 3, 3, 3], "out": [13, 14, 15]}
    assert [max((y * 3 + x) * 3 + c + 1 for y in (0, 1) for x in (0, 1)) for c in range(3)] == mp["out"]
    k["maxpool_tf_valid"] = mp
    k["_about_conv2d_tf"] = ("conv2d_tf_*: the expected outputs of TensorFlow's conv_ops_test.py (testConv2D1x1Filter, "
                             "testConv2D2x2Filter; inputs and filters 1, 2, 3, ... in memory order), re-derived with integer "
                             "arithmetic by tests/golden/make_external_conv.py; maxpool_tf_valid: pooling_ops_test.py "
                             "(_testMaxPoolValidPadding), likewise")
    json.dump(k, open(p, "w"), indent=1)
    print("added:", sorted(QUOTED))


if __name__ == "__main__":
    main()
