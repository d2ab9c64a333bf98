"""CPU oracle for the iso-surface stage -- TEST INFRASTRUCTURE ONLY.

The reference extracts the mesh with the closed Vega-FEM binary ``isosurface/computeMarchingCubes``
(test/create_sdf.py:305-322), which can neither be read nor run here: PARITY UNPINNED.  This is a
numpy statement of the algorithm the HIP kernels implement (indexed marching cubes with the
face-consistent case table of tools/gen_mc_tables.py), in the same vertex / face ORDER and the
same float32 operation order, so the GPU output is compared bit for bit; correctness of the
algorithm itself is established by independent properties (watertightness, orientation, volume
and area of analytic shapes, vertices lying on the iso level) in tests/test_marching_cubes.py.
"""
from __future__ import annotations

import importlib.util
import os
from typing import Tuple

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("gen_mc_tables", os.path.join(_ROOT, "tools", "gen_mc_tables.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)

F32 = np.float32


def grid_axes(sdf_params, res: int):
    """float32 grid coordinates per axis, exactly as the point grid (test/create_sdf.py:247-249)"""
    p = np.asarray(sdf_params, dtype=np.float64)
    return [np.linspace(p[a], p[a + 3], num=res + 1).astype(np.float32) for a in range(3)]


def marching_cubes(vol: np.ndarray, sdf_params, iso: float) -> Tuple[np.ndarray, np.ndarray]:
    """vol [(R+1),(R+1),(R+1)] indexed [iz,iy,ix] (the .dist order, x fastest) ->
    (verts [nv,3] float32 world coordinates, faces [nf,3] int32, 0-based)."""
    vol = np.asarray(vol, np.float32)
    n = vol.shape[0]
    R = n - 1
    iso = F32(iso)
    ax = grid_axes(sdf_params, R)
    inside = vol < iso
    # ---- vertices: one per cut grid edge, numbered in flat order 3*p + axis (p = (iz*n+iy)*n+ix)
    flags = np.zeros((n, n, n, 3), bool)
    flags[:, :, :-1, 0] = inside[:, :, :-1] != inside[:, :, 1:]
    flags[:, :-1, :, 1] = inside[:, :-1, :] != inside[:, 1:, :]
    flags[:-1, :, :, 2] = inside[:-1, :, :] != inside[1:, :, :]
    flat = flags.reshape(-1)
    eidx = np.cumsum(flat, dtype=np.int64) - flat                     # exclusive scan
    act = np.nonzero(flat)[0]
    p, axis = act // 3, act % 3
    iz, iy, ix = p // (n * n), (p // n) % n, p % n
    step = np.array([1, n, n * n])[axis]
    v0 = vol.reshape(-1)[p]
    v1 = vol.reshape(-1)[p + step]
    t = ((iso - v0) / (v1 - v0)).astype(np.float32)
    verts = np.stack([ax[0][ix], ax[1][iy], ax[2][iz]], 1).astype(np.float32)
    idx = np.stack([ix, iy, iz], 1)
    for a in range(3):
        m = axis == a
        c0 = ax[a][idx[m, a]]
        c1 = ax[a][idx[m, a] + 1]
        verts[m, a] = (c0 + (t[m] * (c1 - c0).astype(np.float32)).astype(np.float32)).astype(np.float32)
    # ---- faces: cells in flat order (iz,iy,ix), triangles in table order
    ntri, tri, maxt = gen.build_tables()
    eg = np.array(gen.edge_geometry())                               # [12,4] dx,dy,dz,axis
    corner = gen.CORNERS.astype(int)
    mask = np.zeros((R, R, R), np.int32)
    for c in range(8):
        dx, dy, dz = corner[c]
        mask |= inside[dz:dz + R, dy:dy + R, dx:dx + R].astype(np.int32) << c
    mflat = mask.reshape(-1)
    cells = np.nonzero(ntri[mflat] > 0)[0]
    cz, cy, cx = cells // (R * R), (cells // R) % R, cells % R
    faces = []
    for k in range(maxt):
        sel = ntri[mflat[cells]] > k
        if not sel.any():
            break
        tri_k = tri[mflat[cells[sel]], 3 * k:3 * k + 3].astype(int)   # [m,3] cube-edge ids
        out = np.empty_like(tri_k, dtype=np.int64)
        for j in range(3):
            e = eg[tri_k[:, j]]
            gp = ((cz[sel] + e[:, 2]) * n + (cy[sel] + e[:, 1])) * n + (cx[sel] + e[:, 0])
            out[:, j] = eidx[3 * gp + e[:, 3]]
        faces.append((cells[sel], k, out))
    if faces:
        order_cell = np.concatenate([f[0] for f in faces])
        order_k = np.concatenate([np.full(len(f[0]), f[1]) for f in faces])
        allf = np.concatenate([f[2] for f in faces])
        o = np.lexsort((order_k, order_cell))
        allf = allf[o]
    else:
        allf = np.zeros((0, 3), np.int64)
    return verts, allf.astype(np.int32)


# ---------------------------------------------------------------- mesh properties (independent checks)
def mesh_is_closed_and_oriented(faces: np.ndarray) -> Tuple[bool, int]:
    """every directed edge (a,b) must be matched by exactly one (b,a): closed 2-manifold-ish,
    consistently oriented.  Returns (ok, number of unmatched directed edges)."""
    f = np.asarray(faces, np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e[:, 0] * (f.max() + 1 if len(f) else 1) + e[:, 1]
    rkey = e[:, 1] * (f.max() + 1 if len(f) else 1) + e[:, 0]
    uk, cnt = np.unique(key, return_counts=True)
    if (cnt != 1).any():
        return False, int((cnt != 1).sum())
    missing = np.setdiff1d(rkey, uk)
    return len(missing) == 0, int(len(missing))


def mesh_volume_area(verts: np.ndarray, faces: np.ndarray) -> Tuple[float, float]:
    v = np.asarray(verts, np.float64)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    vol = float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
    area = float(np.linalg.norm(np.cross(b - a, c - a), axis=1).sum() / 2.0)
    return vol, area
