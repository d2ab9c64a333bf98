"""Dense-grid SDF evaluation driver: the hot loop of the reference's ``test/create_sdf.py``
(and ``demo/demo.py``) on the HIP engine.

Reference (/root/reference): split arithmetic test/create_sdf.py:69-77; grid construction
:246-256; per-split ``sess.run`` loop :262-276; un-pad + ``/ SDF_WEIGHT`` :277-285; ``.dist``
writer :292-303.  Differences by design (MI355X-first):
  * the encoder runs ONCE per image, not once per split (80x at res 256);
  * grid points are generated on the device from ``sdf_params`` (float64 linspace, cast to
    float32 -- bit-identical to the numpy grid), no 204 MB host->device copy, no padding points;
  * chunks stream through ``disn_query_grid`` in the flat (iz,iy,ix) order of the reference;
  * with ``torch.distributed`` initialised the flat index range is sharded contiguously
    across ranks and collected with one all_gather (RCCL over xGMI) -- see parallel.py.
"""
from __future__ import annotations

import math
import struct
from typing import Optional, Sequence, Tuple

import numpy as np

SDF_WEIGHT = 10.0   # test/create_sdf.py:285


def split_plan(sdf_res: int, twostream: bool = True) -> Tuple[int, int, int, int]:
    """(TOTAL_POINTS, SPLIT_SIZE, NUM_SAMPLE_POINTS, pad) exactly as test/create_sdf.py:69-77.
    Kept for callers that feed fixed-shape splits through ``Session.run``."""
    resolution = sdf_res + 1
    total = resolution ** 3
    split = int(math.ceil(total / (214669.0 if twostream else 274625.0)))
    nsp = int(math.ceil(total / split))
    return total, split, nsp, split * nsp - total


def grid_points_host(sdf_params: Sequence[float], sdf_res: int) -> np.ndarray:
    """Host grid as the reference builds it (test/create_sdf.py:246-256) -- for callers that
    still feed points through placeholders.  The device path never materialises this."""
    res = sdf_res + 1
    # float64 linspace: what numpy 1.x computes for int / float64 sdf_params (demo/demo.py:278 passes ints).  For
    # FLOAT32 sdf_params numpy 1.x forms delta = stop - start and step = delta / div as float32 scalars before the
    # float64 arange multiply; with a box whose float32 difference or division is inexact (not the +-1 demo box
    # or any dyadic box) the coordinates then drift from this grid by the accumulated step rounding, up to
    # R * ulp32(step) / 2 ~ 1e-7 of the box (tests/test_oracle.py::test_grid_float32_params_caveat).  numpy >= 2 computes in float32.
    p = np.asarray(sdf_params, dtype=np.float64)
    x_ = np.linspace(p[0], p[3], num=res)
    y_ = np.linspace(p[1], p[4], num=res)
    z_ = np.linspace(p[2], p[5], num=res)
    z, y, x = np.meshgrid(z_, y_, x_, indexing='ij')
    return np.stack((x, y, z), axis=3).astype(np.float32).reshape(-1, 3)


def to_binary(res: int, pos: Sequence[float], pred_sdf_val_all: np.ndarray, sdf_file: str) -> None:
    """The ``.dist`` wire format consumed by isosurface/computeMarchingCubes
    (test/create_sdf.py:292-303): int32 -res, res, res; 6 x float64 bbox (min xyz, max xyz);
    (res+1)^3 float32 values, x fastest."""
    vals = np.ascontiguousarray(pred_sdf_val_all, dtype=np.float32).ravel()
    with open(sdf_file, 'wb') as f:
        f.write(struct.pack('i', -res))
        f.write(struct.pack('i', res))
        f.write(struct.pack('i', res))
        f.write(struct.pack('d' * len(pos), *[float(v) for v in pos]))
        f.write(vals.astype('<f4').tobytes())


def read_dist(sdf_file: str):
    """Inverse of to_binary (format as read by preprocessing/create_point_sdf_grid.py:29-51)."""
    with open(sdf_file, 'rb') as f:
        raw = f.read()
    ress = np.frombuffer(raw[:12], dtype=np.int32)
    res = int(ress[1])
    if -ress[0] != res or ress[2] != res:
        raise ValueError("inconsistent .dist header %s" % (ress,))
    pos = np.frombuffer(raw[12:12 + 48], dtype=np.float64)
    vals = np.frombuffer(raw[60:], dtype=np.float32).reshape(res + 1, res + 1, res + 1)
    return res, pos, vals


def dense_grid_sdf(engine, enc, image_index: int, trans_mat, sdf_params, sdf_res: int,
                   sdf_weight: float = SDF_WEIGHT, out=None, k_range: Optional[Tuple[int, int]] = None):
    """SDF/10 on the (res+1)^3 grid of one encoded image, flat (iz,iy,ix) order, as a device
    tensor.  ``k_range`` restricts to a contiguous flat-index slice (used by the sharded path)."""
    total = (sdf_res + 1) ** 3
    k0, k1 = (0, total) if k_range is None else k_range
    return engine.query_grid(enc, image_index, trans_mat, sdf_params, sdf_res, k0, k1, sdf_weight, out)


def create_sdf(engine, imgs, trans_mats, sdf_params, sdf_res: int, sdf_weight: float = SDF_WEIGHT):
    """``test_one_epoch`` for one batch (test/create_sdf.py:240-285): returns ``result`` --
    a float32 device tensor [B, (res+1)^3] of pred_sdf / SDF_WEIGHT."""
    import torch
    imgs = np.asarray(imgs, np.float32) if not isinstance(imgs, torch.Tensor) else imgs
    B = imgs.shape[0]
    enc = engine.encode(imgs)
    total = (sdf_res + 1) ** 3
    result = torch.empty((B, total), dtype=torch.float32, device=engine.device)
    sp = np.asarray(sdf_params, dtype=np.float64).reshape(B, 6)
    for b in range(B):
        dense_grid_sdf(engine, enc, b, trans_mats, sp[b], sdf_res, sdf_weight, out=result[b])
    return result
