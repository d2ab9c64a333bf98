"""Mesh extraction from the predicted SDF grid: the stage behind the hot path in
``test/create_sdf.py`` (create_obj :305-317, create_one_cube_obj :319-323), without the
``.dist`` file and the ``./isosurface/computeMarchingCubes`` subprocess.

``marching_cubes`` meshes the grid where it already lies (device tensor from
``create_sdf.dense_grid_sdf``); ``create_obj`` mirrors the reference helper's name and arguments
and writes ``<dir>/<cat_id>/<cat_id>_<obj_nm>_<view_id>.obj``.  The case table is derived in
``tools/gen_mc_tables.py``; the reference's closed binary cannot be compared against
(SURVEY §2 row 11): where a cell is topologically ambiguous its triangulation may differ, the
vertices (on grid edges, linear interpolation at ``iso``) do not.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def marching_cubes(sdf: torch.Tensor, sdf_params, res: int, iso: float = 0.0,
                   ws: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """sdf: float32 device tensor with (res+1)^3 values in the flat (iz,iy,ix) order of the
    ``.dist`` format.  -> (verts [nv,3] float32, faces [nf,3] int32 0-based), on the device."""
    sdf = ops._chk(sdf.reshape(-1), "sdf")
    n = res + 1
    if sdf.numel() != n * n * n:
        raise ValueError("sdf must hold (res+1)^3 = %d values, got %d" % (n * n * n, sdf.numel()))
    dev = sdf.device
    with torch.cuda.device(dev):
        need = lib().disn_mc_workspace_bytes(res)
        if need == 0:
            raise ValueError("unsupported resolution %d" % res)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        counts = torch.zeros(2, dtype=torch.int64, device=dev)
        st = ops._stream()
        check("disn_mc_count", lib().disn_mc_count(sdf.data_ptr(), res, float(iso), counts.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), st))
        nv, nf = (int(v) for v in counts.tolist())          # the one host sync: sizes are data dependent
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        faces = torch.empty((nf, 3), dtype=torch.int32, device=dev)
        if nv and nf:
            p6 = ops._params6(sdf_params)
            check("disn_mc_emit", lib().disn_mc_emit(sdf.data_ptr(), C.byref(p6), res, float(iso),
                                                     verts.data_ptr(), faces.data_ptr(), ws.data_ptr(),
                                                     ws.numel(), st))
    return verts, faces


def write_obj(path: str, verts, faces) -> None:
    """Wavefront .obj ("v x y z" / "f a b c", 1-based)."""
    v = np.ascontiguousarray(verts.detach().cpu().numpy() if isinstance(verts, torch.Tensor) else verts, np.float32)
    f = np.ascontiguousarray(faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else faces, np.int32)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    check("disn_write_obj", lib().disn_write_obj(path.encode(), v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0]))


def read_obj(path: str):
    vs, fs = [], []
    for line in open(path):
        if line.startswith("v "):
            vs.append([float(t) for t in line.split()[1:4]])
        elif line.startswith("f "):
            fs.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    return np.asarray(vs, np.float32).reshape(-1, 3), np.asarray(fs, np.int32).reshape(-1, 3)


def create_obj(pred_sdf_val, sdf_params, dir, cat_id, obj_nm, view_id, i, res: Optional[int] = None) -> str:
    """test/create_sdf.py:305-317 -- same arguments (``i`` is the iso value), same output path;
    ``pred_sdf_val`` may be a device tensor (preferred) or a numpy array of (res+1)^3 values."""
    if not isinstance(view_id, str):
        view_id = "%02d" % view_id
    out_dir = os.path.join(dir, cat_id)
    os.makedirs(out_dir, exist_ok=True)
    cube_obj_file = os.path.join(out_dir, cat_id + "_" + obj_nm + "_" + view_id + ".obj")
    if not isinstance(pred_sdf_val, torch.Tensor):
        pred_sdf_val = torch.from_numpy(np.ascontiguousarray(pred_sdf_val, np.float32)).cuda()
    n = round(pred_sdf_val.numel() ** (1.0 / 3.0))
    res = (n - 1) if res is None else res
    verts, faces = marching_cubes(pred_sdf_val, np.asarray(sdf_params, np.float64), res, float(i))
    write_obj(cube_obj_file, verts, faces)
    return cube_obj_file
