"""Trained-like stress set (VERDICT r3 #2c) against the float64 oracle -> tests/golden/stress_trained_like.npz.

    python tests/golden/make_golden_stress.py        (CPU, ~2 minutes; needs nothing outside the repo)

Weights: `O.trained_like_weights(STRESS_SEED)` -- heavy-tailed (Student-t, 4 d.o.f.) entries, log-normal per-channel
gains (sigma 1) with 1 % outlier channels x 10^3, compensated in the consumers so that |pred| stays O(1) (see its
docstring).  Images: the reference's demo chair (white background, oracle_kat.npz), a U[0,1) image, the chair mirrored,
and a synthetic white-background image with a dark blob -- the statistics the per-image power-of-two activation scale
of the two-term f16 split (conv_h2w.hip:176-179 and siblings) has to survive.

Stored (float64, nothing from the GPU):
  pred64_a   [4][2048]  pred_sdf of 2048 points per image (own camera): the single-step form (conv_h2 / dense_h2) runs
                        images 0..1 one at a time, the batched form (conv_h2w / dense_h2w) all four in one call
  pred64_b   [40960]    image 0, one large point set: the folded feature map + fused point MLP (mlp_fused.hip)
  emb64      [4][1024]  img_embedding
  tap64_<name>, tapidx_<name>   every TAP_STRIDE-th element (flat NHWC index) of the five taps of images 0 and 3
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import disn_oracle as O   # noqa: E402

STRESS_SEED = 11
TAP_STRIDE = 211
N_A, N_B = 2048, 40960


def stress_inputs():
    kat = np.load(os.path.join(HERE, "oracle_kat.npz"))
    demo = (kat["demo_img"].astype(np.float32) / np.float32(255.0))[0]
    rng = np.random.default_rng(STRESS_SEED)
    rnd = rng.random((137, 137, 3), dtype=np.float32)
    yy, xx = np.mgrid[0:137, 0:137].astype(np.float32)
    blob = np.exp(-(((xx - 70) / 30) ** 2 + ((yy - 60) / 22) ** 2)).astype(np.float32)
    white = (np.float32(1.0) - np.float32(0.85) * blob)[:, :, None] * np.array([1.0, 0.97, 0.93], np.float32)
    imgs = np.ascontiguousarray(np.stack([demo, rnd, demo[:, ::-1], white.astype(np.float32)]).astype(np.float32))
    tms = np.concatenate([O.DEMO_TRANS_MAT, O.synth_trans_mat(40.0, 25.0)[None], O.synth_trans_mat(150.0, 30.0)[None],
                          O.synth_trans_mat(280.0, 20.0)[None]]).astype(np.float32)
    pts_a = (rng.random((4, N_A, 3), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    pts_b = (rng.random((1, N_B, 3), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    return {"imgs": imgs, "trans_mat": tms, "pts_a": pts_a, "pts_b": pts_b}


def main():
    t0 = time.time()
    W = O.trained_like_weights(STRESS_SEED)
    s = stress_inputs()
    f64 = np.float64
    _, emb, maps, eps = O.encode(s["imgs"], W, f64)
    print("encode %.1f s; |emb| max %.3g" % (time.time() - t0, np.abs(emb).max()), flush=True)
    out = {"emb64": np.asarray(emb, f64)}
    for nm in O.TAP_NAMES:
        tap = np.asarray(eps["vgg_16/%s/%s" % (nm[:5], nm)], f64)[[0, 3]]
        idx = np.arange(0, tap[0].size, TAP_STRIDE)
        out["tap64_" + nm] = tap.reshape(2, -1)[:, idx]
        out["tapmax_" + nm] = np.abs(tap).reshape(2, -1).max(1)
        cm = np.abs(tap[0]).reshape(-1, tap.shape[-1]).max(0)
        print("%s: max %.3g, median channel max %.3g" % (nm, cm.max(), np.median(cm)))

    def pred(b, pts):
        xy = O.get_img_points(pts, s["trans_mat"][b:b + 1])
        feat = O.gather_point_feat([m[b:b + 1] for m in maps], xy)
        return (O.get_sdf_basic2(pts, emb[b:b + 1], W, dtype=f64)
                + O.get_sdf_basic2_imgfeat_twostream(pts, feat, W, dtype=f64))[0, :, 0]

    out["pred64_a"] = np.stack([pred(b, s["pts_a"][b:b + 1]) for b in range(4)])
    out["pred64_b"] = np.concatenate([pred(0, s["pts_b"][:, k:k + 8192]) for k in range(0, N_B, 8192)])
    print("|pred| max: a %.3g  b %.3g; %.1f s" % (np.abs(out["pred64_a"]).max(), np.abs(out["pred64_b"]).max(),
                                                   time.time() - t0))
    # the float32 CPU path's own distance from the truth on this set (image 0)
    p32 = O.get_model({"imgs": s["imgs"][:1], "sample_pc": s["pts_a"][:1], "sample_pc_rot": s["pts_a"][:1],
                       "trans_mat": s["trans_mat"][:1]}, W, dtype=np.float32)["pred_sdf"][0, :, 0]
    out["oracle32_minus_f64_img0"] = np.float64(np.abs(p32.astype(f64) - out["pred64_a"][0]).max())
    print("max |oracle32 - f64| image 0: %.3g" % out["oracle32_minus_f64_img0"])
    np.savez_compressed(os.path.join(HERE, "stress_trained_like.npz"), seed=np.int64(STRESS_SEED),
                        tap_stride=np.int64(TAP_STRIDE), **out)
    print("stress_trained_like.npz", os.path.getsize(os.path.join(HERE, "stress_trained_like.npz")), "bytes")


if __name__ == "__main__":
    main()
