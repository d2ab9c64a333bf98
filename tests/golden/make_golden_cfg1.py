"""BASELINE config 1 in full: the float64 oracle (oracle/disn_oracle.py) on the reference's demo image
(demo/demo.py:263-276: the chair PNG as committed in oracle_kat.npz, the ground-truth camera), He-initialised weights
(seed 0), over the WHOLE 65^3 grid of sdf_res = 64 in .dist order PLUS the pad point the demo's padded splits evaluate
((0, 0, 0): demo/demo.py:292,306) -> tests/golden/cfg1_full65.npz.

    python tests/golden/make_golden_cfg1.py        (CPU, a few minutes; needs nothing outside the repo)

Stored: pred64 [274625 + 1] float64 (pred_sdf, NOT divided by SDF_WEIGHT), in chunks of 16384 points through the same
oracle functions get_model chains (encode once; get_img_points -> gather_point_feat -> both MLP streams per chunk).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import disn_oracle as O   # noqa: E402


def main():
    kat = np.load(os.path.join(HERE, "oracle_kat.npz"))
    img = kat["demo_img"].astype(np.float32) / np.float32(255.0)
    W = O.init_weights(0, "he")
    f64 = np.float64
    _, emb, maps, _ = O.encode(img, W, f64)
    grid = O.grid_points([-1, -1, -1, 1, 1, 1], 64)
    pts = np.concatenate([grid, np.zeros((1, 3), np.float32)], axis=0)           # + the pad point
    out = np.zeros(pts.shape[0], f64)
    for k0 in range(0, pts.shape[0], 16384):
        p = pts[k0:k0 + 16384][None]
        xy = O.get_img_points(p, O.DEMO_TRANS_MAT)
        feat = O.gather_point_feat(maps, xy)
        g = O.get_sdf_basic2(p, emb, W, dtype=f64)
        l = O.get_sdf_basic2_imgfeat_twostream(p, feat, W, dtype=f64)
        out[k0:k0 + p.shape[1]] = (g + l)[0, :, 0]
        print("chunk at %d done" % k0, flush=True)
    # the slice oracle_kat.npz already holds must be reproduced exactly (same functions, same order)
    k0 = int(kat["demo_k0"])
    assert np.array_equal(out[k0:k0 + 4096], kat["demo_pred64"]), np.abs(out[k0:k0 + 4096] - kat["demo_pred64"]).max()
    np.savez_compressed(os.path.join(HERE, "cfg1_full65.npz"), pred64=out, sdf_res=np.int64(64))
    print("cfg1_full65.npz", os.path.getsize(os.path.join(HERE, "cfg1_full65.npz")), "bytes; |pred| max %.3f" % np.abs(out).max())


if __name__ == "__main__":
    main()
