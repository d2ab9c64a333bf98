"""INDEPENDENT marching-cubes pin -- run with the image's conda interpreter, which carries scikit-image:

    /opt/conda/bin/python3.9 tests/golden/make_mc_fixtures.py      -> tests/golden/mc_skimage.npz

The reference meshes with Vega-FEM's closed `computeMarchingCubes` binary (test/create_sdf.py:305-322), which cannot run
here.  `skimage.measure.marching_cubes` (0.18.3, Lewiner et al.'s tables) is neither this repo's table generator
(tools/gen_mc_tables.py) nor its oracle (oracle/mc_oracle.py): a geometric pin, "independent, not reference-held".
Per volume the fixture holds the INPUT (float32 volume, box, iso level) and skimage's OUTPUT reduced to what a different
triangulation of the same surface must share: the vertex cloud (world coordinates, float32), V - E + F, the area and the
enclosed volume.  Volumes: an off-centre sphere (33^3), a two-blob smooth field with a thin neck (41^3), the float64
oracle's SDF grid of BASELINE config 1 (65^3, tests/golden/cfg1_full65.npz -- random-init weights: a rough surface), and a
low-pass noise field (25^3) whose ambiguous cubes the two tables may resolve differently (cloud + area only)."""
import os

import numpy as np
from skimage.measure import marching_cubes, mesh_surface_area

HERE = os.path.dirname(os.path.abspath(__file__))
BOX = np.array([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0])


def sphere(R, r=0.6, c=(0.05, -0.1, 0.02)):
    ax = np.linspace(-1, 1, R + 1)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    return (np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r).astype(np.float32)


def blobs(R):
    ax = np.linspace(-1, 1, R + 1)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    a = np.sqrt((x + 0.35) ** 2 + y ** 2 + z ** 2) - 0.42
    b = np.sqrt((x - 0.35) ** 2 + (y - 0.05) ** 2 + (z + 0.03) ** 2) - 0.38
    k = 0.15                                             # smooth minimum: a neck between the two
    h = np.clip(0.5 + 0.5 * (b - a) / k, 0, 1)
    return (b * (1 - h) + a * h - k * h * (1 - h)).astype(np.float32)


def smooth_noise(R, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((R + 1,) * 3)
    for ax in range(3):                                  # separable [1 2 1] smoothing, three passes
        for _ in range(3):
            v = 0.25 * np.roll(v, 1, ax) + 0.5 * v + 0.25 * np.roll(v, -1, ax)
    v = (v / np.abs(v).max()).astype(np.float32)
    v[0] = v[-1] = 1; v[:, 0] = v[:, -1] = 1; v[:, :, 0] = v[:, :, -1] = 1
    return v


def run(vol, box, iso):
    n = vol.shape[0]
    sp = tuple((box[a + 3] - box[a]) / (n - 1) for a in (2, 1, 0))          # volume is indexed [z, y, x]
    v, f, _, _ = marching_cubes(vol.astype(np.float64), level=float(iso), spacing=sp)
    world = np.stack([v[:, 2] + box[0], v[:, 1] + box[1], v[:, 0] + box[2]], 1)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = np.unique(np.sort(e, 1), axis=0)
    chi = len(v) - len(e) + len(f)
    p = world[f]
    vol6 = np.abs(np.einsum("ij,ij->i", p[:, 0], np.cross(p[:, 1], p[:, 2])).sum()) / 6.0
    return world.astype(np.float32), int(chi), float(mesh_surface_area(v, f)), float(vol6)


out = {}
cfg1 = np.load(os.path.join(HERE, "cfg1_full65.npz"))
g65 = (cfg1["pred64"][:65 ** 3].reshape(65, 65, 65) / 10.0).astype(np.float32)       # the SDF values (test/create_sdf.py:285)
cases = {"sphere33": (sphere(32), 0.0), "blobs41": (blobs(40), 0.0), "cfg1grid65": (g65, float(np.median(g65))),
         "noise25": (smooth_noise(24, 5), 0.0)}
for name, (vol, iso) in cases.items():
    cloud, chi, area, vol6 = run(vol, BOX, iso)
    out[name + "_vol"], out[name + "_iso"] = vol, np.float32(iso)
    out[name + "_cloud"], out[name + "_chi"], out[name + "_area"], out[name + "_volume"] = cloud, chi, area, vol6
    print("%-11s n %d iso %+.4f: %6d vertices, chi %d, area %.5f, volume %.5f" % (name, vol.shape[0], iso, len(cloud), chi, area, vol6))
out["box"] = BOX
np.savez_compressed(os.path.join(HERE, "mc_skimage.npz"), **out)
print("wrote tests/golden/mc_skimage.npz", os.path.getsize(os.path.join(HERE, "mc_skimage.npz")), "bytes")
