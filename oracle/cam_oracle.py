"""CPU oracle of the camera head -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED
(TensorFlow is not available here; the restatement is checked against closed-form properties).

  fully_connected        utils/tf_util.py:328-364   out = act(x @ W + b), W [in,out], act = relu unless None
  get_cam_mat            models/posenet.py:91-124   towers on the 1024-d VGG embedding:
        scale        1024 -> 64 -> 32 -> 1      (fc3 linear)        pred_scale = s * I3
        ortho6d      1024 -> 512 -> 256 -> 6    (fc3 linear)
        translation  1024 -> 128 -> 64 -> 3     (fc3 = explicit matmul + bias_add, linear)
                     + const [-0.00193892, 0.00169222, 1.3949631]
        pred_rotation_mat = (s I3) @ R(ortho6d); pred_RT = concat([rotation_mat, translation], axis=1) [B,4,3]
  compute_rotation_matrix_from_ortho6d   models/posenet.py:22-36   Gram-Schmidt: x = n(a), z = n(x × b),
        y = z × x, columns (x, y, z); normalize_vector :13-19 clamps the norm at 1e-8
  pred_trans_mat         cam_est/model_cam.py:102-103   pred_RT @ K^T, K = [[149.84375,0,68.5],[0,149.84375,68.5],[0,0,1]]
                         (:28)
Variable names (scope 'cameraprediction', cam_est/model_cam.py:82): cameraprediction/<tower>/fc{1,2,3}/{weights,biases}.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

K_DEFAULT = np.array([[149.84375, 0.0, 68.5], [0.0, 149.84375, 68.5], [0.0, 0.0, 1.0]], np.float32)
TRANS_CONST = np.array([-0.00193892, 0.00169222, 1.3949631], np.float32)
TOWERS = (("scale", (1024, 64, 32, 1)), ("ortho6d", (1024, 512, 256, 6)), ("translation", (1024, 128, 64, 3)))


def variable_shapes() -> Dict[str, Tuple[int, ...]]:
    s = {}
    for tower, dims in TOWERS:
        for i in range(3):
            s["cameraprediction/%s/fc%d/weights" % (tower, i + 1)] = (dims[i], dims[i + 1])
            s["cameraprediction/%s/fc%d/biases" % (tower, i + 1)] = (dims[i + 1],)
    return s


def init_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in variable_shapes().items():
        if name.endswith("weights"):
            out[name] = (rng.standard_normal(shp) * np.sqrt(2.0 / shp[0])).astype(np.float32)
        else:
            out[name] = (0.1 * rng.standard_normal(shp)).astype(np.float32)
    return out


def normalize_vector(v):
    mag = np.sqrt((v * v).sum(axis=1, keepdims=True))
    return v / np.maximum(mag, v.dtype.type(1e-8))


def compute_rotation_matrix_from_ortho6d(poses):
    x = normalize_vector(poses[:, 0:3])
    z = normalize_vector(np.cross(x, poses[:, 3:6]))
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=2)  # columns x, y, z


def get_cam_mat(embedding, W, dtype=np.float32):
    """-> (pred_rotation_mat [B,3,3], pred_translation [B,1,3], pred_RT [B,4,3])"""
    e = np.asarray(embedding, dtype)
    outs = {}
    for tower, _ in TOWERS:
        h = e
        for i in range(3):
            h = h @ np.asarray(W["cameraprediction/%s/fc%d/weights" % (tower, i + 1)], dtype) \
                + np.asarray(W["cameraprediction/%s/fc%d/biases" % (tower, i + 1)], dtype)
            if i < 2:
                h = np.maximum(h, 0)
        outs[tower] = h
    B = e.shape[0]
    scale = outs["scale"].reshape(B, 1, 1) * np.eye(3, dtype=dtype)[None]
    rot = compute_rotation_matrix_from_ortho6d(outs["ortho6d"].reshape(B, 6))
    trans = (outs["translation"].reshape(B, 3) + TRANS_CONST.astype(dtype)).reshape(B, 1, 3)
    rot = scale @ rot
    return rot, trans, np.concatenate([rot, trans], axis=1)


def pred_trans_mat(pred_RT, K=K_DEFAULT):
    K = np.asarray(K, pred_RT.dtype)
    if K.ndim == 2:
        K = K[None]
    return pred_RT @ np.transpose(K, (0, 2, 1))
