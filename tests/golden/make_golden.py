"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container (needs /root/reference):   python tests/golden/make_golden.py

Two kinds of vectors:

1. ``reference_pins.npz`` -- outputs of the REFERENCE'S OWN CODE.  TensorFlow is not
   installable here, so the TF-executed part of the path cannot be run; but three pieces of the
   path's contract are plain numpy/struct and are executed from the reference sources
   themselves (functions are extracted by ``ast`` so that module-level ``import tensorflow`` /
   argparse never runs):
     * ``to_binary``            test/create_sdf.py:292-303         (.dist wire format)
     * ``getBlenderProj``       preprocessing/create_img_h5.py:14-63 (camera convention)
     * split arithmetic         test/create_sdf.py:69-77 (evaluated verbatim with a FLAGS stub)
     * grid construction        test/create_sdf.py:247-255 (statements evaluated verbatim)
2. ``oracle_kat.npz`` -- known-answer vectors of the oracle (oracle/disn_oracle.py) on seeded
   inputs, plus the projection KATs derived from constants in the reference
   (demo/demo.py:272-276).  These pin the oracle against accidental edits and give the GPU
   tests a fixture that does not need the oracle's slow paths.
"""
from __future__ import annotations

import ast
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def extract(path: str, names, extra_globals=None):
    """exec only the named top-level functions / assignments of a reference file."""
    src = open(path).read()
    tree = ast.parse(src)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            keep.append(node)
    mod = ast.Module(body=keep, type_ignores=[])
    g = {"np": np, "__name__": "ref_extract"}
    import struct
    g["struct"] = struct
    g.update(extra_globals or {})
    exec(compile(mod, path, "exec"), g)
    return g


def statements(path: str, first: int, last: int) -> str:
    lines = open(path).read().splitlines()[first - 1:last]
    import textwrap
    return textwrap.dedent("\n".join(lines))


def reference_pins():
    out = {}
    # ---- .dist writer ---------------------------------------------------------------------
    g = extract(os.path.join(REF, "test/create_sdf.py"), {"to_binary"})
    rng = np.random.default_rng(7)
    res = 4
    vals = rng.standard_normal((res + 1) ** 3).astype(np.float32)
    pos = [-1.0, -0.5, -0.25, 1.0, 0.75, 0.5]
    with tempfile.NamedTemporaryFile(delete=False) as f:
        name = f.name
    g["to_binary"](res, pos, vals, name)
    out["dist_res"] = np.int64(res)
    out["dist_pos"] = np.asarray(pos, np.float64)
    out["dist_vals"] = vals
    out["dist_bytes"] = np.frombuffer(open(name, "rb").read(), dtype=np.uint8)
    os.unlink(name)

    # ---- getBlenderProj ---------------------------------------------------------------------
    g = extract(os.path.join(REF, "preprocessing/create_img_h5.py"), {"getBlenderProj", "rot90y"})
    cams = np.array([[30.0, 25.0, 0.8], [201.5, 30.0, 0.65], [310.0, 12.5, 0.9], [0.0, 0.0, 1.0]])
    Ks, RTs = [], []
    for az, el, d in cams:
        K, RT = g["getBlenderProj"](az, el, d, img_w=137, img_h=137)
        Ks.append(np.asarray(K, np.float64)); RTs.append(np.asarray(RT, np.float64))
    out["cam_params"] = cams
    out["cam_K"] = np.stack(Ks); out["cam_RT"] = np.stack(RTs)
    out["rot90y"] = np.asarray(g["rot90y"], np.float32)

    # ---- split arithmetic (test/create_sdf.py:69-77 evaluated verbatim) ------------------------
    code = statements(os.path.join(REF, "test/create_sdf.py"), 69, 77)
    plans = []
    for r in (16, 32, 64, 100, 128, 256):
        FLAGS = types.SimpleNamespace(sdf_res=r, img_feat_twostream=True, threedcnn=False,
                                      num_points=1, batch_size=1)
        env = {"np": np, "FLAGS": FLAGS}
        exec(code.replace("NUM_POINTS = FLAGS.num_points", "").replace("BATCH_SIZE = FLAGS.batch_size", ""), env)
        plans.append([r, env["TOTAL_POINTS"], env["SPLIT_SIZE"], env["NUM_SAMPLE_POINTS"]])
    out["split_plans"] = np.asarray(plans, np.int64)

    # ---- grid construction (test/create_sdf.py:247-255 evaluated verbatim) ----------------------
    code = statements(os.path.join(REF, "test/create_sdf.py"), 247, 255)
    for tag, sp, r in (("a", np.array([-1, -1, -1, 1, 1, 1], np.float32), 8),
                       ("b", np.array([-0.83, -0.41, -0.27, 0.79, 0.55, 0.31], np.float32), 5)):
        # NOTE: fed as float64.  The reference ran on numpy 1.x, where np.linspace of float32
        # scalars computes in float64; under this container's numpy 2.x (NEP 50) float32 inputs
        # would make the very same statements compute in float32.  Feeding the (exactly
        # representable) values as float64 reproduces the reference environment's arithmetic.
        env = {"np": np, "sdf_params": sp.astype(np.float64), "RESOLUTION": r + 1}
        exec(code, env)
        out["grid_%s_params" % tag] = sp
        out["grid_%s_res" % tag] = np.int64(r)
        out["grid_%s_pts" % tag] = env["all_pts"].reshape(-1, 3)
    return out


def oracle_kat():
    from oracle import disn_oracle as O
    out = {}
    # projection KATs: demo GT matrix demo/demo.py:272-276
    pts = np.array([[[0, 0, 0], [1, 1, 1], [-1, -1, -1], [0.5, -0.25, 0.1], [1, -1, 1]]], np.float32)
    out["proj_pts"] = pts
    out["proj_xy"] = O.get_img_points(pts, O.DEMO_TRANS_MAT)
    out["proj_xy_survey"] = np.array([[70.69476, 70.84084], [0.0, 17.558498], [129.501, 101.92336],
                                      [49.962322, 93.89276], [1.6307365, 120.86008]], np.float32)
    rng = np.random.default_rng(11)
    # resize edge cases (Appendix A.3)
    for hin, hout in ((14, 137), (224, 137), (137, 224), (28, 137)):
        a = rng.random((1, hin, hin, 1), dtype=np.float32)
        out["resize_%d_%d_in" % (hin, hout)] = a
        out["resize_%d_%d_out" % (hin, hout)] = O.resize_bilinear_legacy(a, hout, hout)
    # resampler edge cases
    d = rng.random((1, 137, 137, 2), dtype=np.float32)
    w = (rng.random((1, 64, 2), dtype=np.float32) * 140 - 2).astype(np.float32)
    w[0, :6] = [[0, 0], [136, 136], [136, 0], [-0.5, 3.2], [136.5, 10.0], [50.25, 99.75]]
    out["resampler_data"] = d; out["resampler_warp"] = w; out["resampler_out"] = O.resampler(d, w)
    # full model, cfg2 (seed 0, 2048 pts), both weight sets: fp32 oracle and fp64 shadow
    for mode in ("xavier", "he"):
        W = O.init_weights(0, mode)
        feed = O.synth_inputs(0, 1, 2048)
        ep = O.get_model(feed, W)
        ep64 = O.get_model(feed, W, dtype=np.float64)
        out["cfg2_%s_pred" % mode] = ep["pred_sdf"].astype(np.float32)
        out["cfg2_%s_pred64" % mode] = ep64["pred_sdf"].astype(np.float64)
        out["cfg2_%s_emb" % mode] = ep["img_embedding"].astype(np.float32)
        out["cfg2_%s_emb64" % mode] = ep64["img_embedding"].astype(np.float64)
        out["cfg2_%s_xy" % mode] = ep["sample_img_points"]
        out["cfg2_%s_feat_sum" % mode] = ep["point_img_feat"].astype(np.float64).sum(axis=(0, 2, 3))
    # cfg1 fixture: demo PNG, GT trans_mat, a 4096-point slice of the 65^3 grid
    img = O.load_demo_image(os.path.join(REF, "demo/03001627_17e916fc863540ee3def89b32cef8e45_20.png"))
    out["demo_img"] = (img * 255.0 + 0.5).astype(np.uint8)           # BGR uint8, 56 KB
    W = O.init_weights(0, "he")
    grid = O.grid_points([-1, -1, -1, 1, 1, 1], 64)
    k0 = 130000
    sl = grid[k0:k0 + 4096][None]
    feed = {"imgs": out["demo_img"].astype(np.float32) / np.float32(255.0), "sample_pc": sl,
            "sample_pc_rot": sl, "trans_mat": O.DEMO_TRANS_MAT}
    out["demo_k0"] = np.int64(k0)
    out["demo_pred64"] = O.get_model(feed, W, dtype=np.float64)["pred_sdf"][0, :, 0]
    out["demo_pred"] = O.get_model(feed, W)["pred_sdf"][0, :, 0].astype(np.float32)
    return out


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("needs %s (build container only)" % REF)
    np.savez_compressed(os.path.join(HERE, "reference_pins.npz"), **reference_pins())
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **oracle_kat())
    for f in ("reference_pins.npz", "oracle_kat.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
