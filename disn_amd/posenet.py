"""Camera head -- host mirror of models/posenet.py and the inference half of cam_est/model_cam.py.

  get_cam_mat(globalfeat) -> pred_rotation_mat [B,3,3], pred_translation [B,1,3], pred_RT [B,4,3]
                                                         (models/posenet.py:91-124)
  CameraEstimator.get_model(imgs) -> end_points{'embedding','pred_rotation','pred_translation','pred_RT',
                                                'pred_trans_mat'}          (cam_est/model_cam.py:47-109)

The estimated `pred_trans_mat` [B,4,3] is what the reference writes into its estimated-camera H5s
(cam_est/train_sdf_cam.py:568-612) and feeds to the SDF network as `trans_mat`.  All device work is
`disn_cam_head` (one launch) behind the VGG encoder of the SDF path; there is no CPU fallback.
Variables: 'cameraprediction/<scale|ortho6d|translation>/fc{1,2,3}/{weights [in,out], biases}'.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import CamWeights, check, lib

TOWERS = (("s", "scale", (1024, 64, 32, 1)), ("r", "ortho6d", (1024, 512, 256, 6)),
          ("t", "translation", (1024, 128, 64, 3)))
K_DEFAULT = np.array([[149.84375, 0.0, 68.5], [0.0, 149.84375, 68.5], [0.0, 0.0, 1.0]], np.float32)


def variable_shapes() -> Dict[str, Tuple[int, ...]]:
    s = {}
    for _, tower, dims in TOWERS:
        for i in range(3):
            s["cameraprediction/%s/fc%d/weights" % (tower, i + 1)] = (dims[i], dims[i + 1])
            s["cameraprediction/%s/fc%d/biases" % (tower, i + 1)] = (dims[i + 1],)
    return s


def random_init(seed: int = 0) -> Dict[str, np.ndarray]:
    """xavier weights / zero biases (tf_util.fully_connected defaults); translation/fc3 is
    truncated_normal(0.05) in the reference (models/posenet.py:108-110)"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in variable_shapes().items():
        if name.endswith("biases"):
            out[name] = np.zeros(shp, np.float32)
        elif name == "cameraprediction/translation/fc3/weights":
            out[name] = np.clip(rng.normal(0.0, 0.05, shp), -0.1, 0.1).astype(np.float32)
        else:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
    return out


class CameraHead:
    def __init__(self, arrays: Dict[str, np.ndarray], device="cuda:0"):
        self.device = torch.device(device)
        self._keep = []
        self.w = CamWeights()
        for name, shp in variable_shapes().items():
            if name not in arrays or tuple(np.shape(arrays[name])) != shp:
                raise ValueError("camera head variable %s missing or of the wrong shape" % name)
        for short, tower, _ in TOWERS:
            for i in (1, 2, 3):
                for kind, leaf in (("w", "weights"), ("b", "biases")):
                    t = torch.from_numpy(np.ascontiguousarray(
                        arrays["cameraprediction/%s/fc%d/%s" % (tower, i, leaf)], np.float32)).to(self.device)
                    self._keep.append(t)
                    setattr(self.w, "%s_%s%d" % (short, kind, i), t.data_ptr())

    def run(self, embedding: torch.Tensor, K: Optional[np.ndarray] = None):
        """-> rotation [B,3,3], translation [B,3], RT [B,4,3], trans_mat [B,4,3] (device tensors)"""
        if not (embedding.is_cuda and embedding.dtype == torch.float32 and embedding.shape[-1] == 1024):
            raise TypeError("embedding must be a float32 CUDA tensor [B,1024]")
        e = embedding.contiguous()
        B = e.shape[0]
        rot = torch.empty((B, 3, 3), dtype=torch.float32, device=e.device)
        tr = torch.empty((B, 3), dtype=torch.float32, device=e.device)
        RT = torch.empty((B, 4, 3), dtype=torch.float32, device=e.device)
        tm = torch.empty((B, 4, 3), dtype=torch.float32, device=e.device)
        k9 = None
        if K is not None:
            k9 = (C.c_float * 9)(*[float(v) for v in np.asarray(K, np.float32).reshape(9)])
        check("disn_cam_head", lib().disn_cam_head(C.byref(self.w), e.data_ptr(), k9, B, rot.data_ptr(),
                                                   tr.data_ptr(), RT.data_ptr(), tm.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream))
        return rot, tr, RT, tm

    def get_cam_mat(self, globalfeat: torch.Tensor):
        """models/posenet.py:91 -- (pred_rotation_mat, pred_translation [B,1,3], pred_RT)"""
        rot, tr, RT, _ = self.run(globalfeat)
        return rot, tr.view(-1, 1, 3), RT


class CameraEstimator:
    """cam_est/model_cam.py:47-109 at inference: the camera network's own VGG-16 (same architecture and
    variable names as the SDF encoder, separate weights) + the head."""

    def __init__(self, vgg_store, head_arrays: Dict[str, np.ndarray], device="cuda:0"):
        from .engine import SdfEngine
        self.engine = SdfEngine(vgg_store, torch.device(device))
        self.head = CameraHead(head_arrays, device)

    def get_model(self, imgs, K: Optional[np.ndarray] = None) -> Dict[str, torch.Tensor]:
        if not isinstance(imgs, torch.Tensor):
            imgs = torch.from_numpy(np.ascontiguousarray(imgs, np.float32)).to(self.head.device)
        enc = self.engine.encode(imgs)
        rot, tr, RT, tm = self.head.run(enc.embedding, K)
        return {"embedding": enc.embedding, "pred_rotation": rot, "pred_translation": tr.view(-1, 1, 3),
                "pred_RT": RT, "pred_trans_mat": tm, "pred_xyshift": None}
