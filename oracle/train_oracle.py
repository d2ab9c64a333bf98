"""CPU oracle for the TRAINING step of the path (BASELINE config 5) -- TEST
INFRASTRUCTURE ONLY.  PARITY UNPINNED (TensorFlow is not available; see disn_oracle.py).

The forward graph of oracle/disn_oracle.py restated with torch-CPU autograd, so that the
gradient of every variable, the losses of get_loss and one optimizer step can be checked:

  forward         models/model_normalization.py:47-221 (two-stream regression branch)
  loss            models/model_normalization.py:254-300: sdf_loss = mean(|10*gt - pred| * w) * 1000,
                  w = 4 where gt <= 0.01 else 1; + sum over every '/weights' variable of
                  wd * ||w||^2 / 2 (slim l2_regularizer on the VGG convs :75, tf_util 'regularizer'
                  collection utils/tf_util.py:45-47), wd = 1e-5
  optimizer       tf.train.AdamOptimizer(lr, beta1=0.5) on ALL global variables
                  (train/train_sdf.py:251,266-268): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
                  m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m/(sqrt(v)+eps)
  learning rate   max(1e-4 * 0.9^floor(step*B/200000), 1e-6)   (train/train_sdf.py:153-161)
  feed            'sdf' = sdf_val - 0.003  (train/train_sdf.py:375) -- the caller's business

The bilinear pieces are written with index arithmetic (differentiable w.r.t. the data, as TF's
ResizeBilinearGrad / ResamplerGrad are; the warp gradient is irrelevant: points and cameras are inputs).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as Fnn

from . import disn_oracle as O

WD = 1e-5


def _resize_legacy(x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """x [B,H,W,C] -> [B,out_h,out_w,C], legacy bilinear (disn_oracle.resize_bilinear_legacy)."""
    B, H, W, C = x.shape
    ylo, yhi, yl = O.resize_index_table(H, out_h)
    xlo, xhi, xl = O.resize_index_table(W, out_w)
    ylo_t, yhi_t = torch.from_numpy(ylo), torch.from_numpy(yhi)
    xlo_t, xhi_t = torch.from_numpy(xlo), torch.from_numpy(xhi)
    yl_t = torch.from_numpy(yl).to(x.dtype).view(1, -1, 1, 1)
    xl_t = torch.from_numpy(xl).to(x.dtype).view(1, 1, -1, 1)
    rlo, rhi = x[:, ylo_t], x[:, yhi_t]
    tl, tr = rlo[:, :, xlo_t], rlo[:, :, xhi_t]
    bl, br = rhi[:, :, xlo_t], rhi[:, :, xhi_t]
    top = tl + (tr - tl) * xl_t
    bot = bl + (br - bl) * xl_t
    return top + (bot - top) * yl_t


def _resampler(data: torch.Tensor, warp: np.ndarray) -> torch.Tensor:
    """data [B,H,W,C], warp [B,N,2] numpy (x,y) -> [B,N,C] (disn_oracle.resampler)."""
    B, H, W, C = data.shape
    outs = []
    for b in range(B):
        x = warp[b, :, 0].astype(np.float32); y = warp[b, :, 1].astype(np.float32)
        ok = (x > -1) & (y > -1) & (x < W) & (y < H)
        xs = np.where(ok, x, 0).astype(np.float32); ys = np.where(ok, y, 0).astype(np.float32)
        fx, fy = np.floor(xs), np.floor(ys)
        cx, cy = fx + 1, fy + 1
        dx, dy = (cx - xs).astype(np.float32), (cy - ys).astype(np.float32)

        def get(ix, iy):
            inb = (ix >= 0) & (iy >= 0) & (ix < W) & (iy < H)
            v = data[b, torch.from_numpy(np.clip(iy, 0, H - 1).astype(np.int64)),
                     torch.from_numpy(np.clip(ix, 0, W - 1).astype(np.int64))]
            return v * torch.from_numpy(inb.astype(np.float32)).to(data.dtype)[:, None]

        def wt(a):
            return torch.from_numpy((a * ok).astype(np.float32)).to(data.dtype)[:, None]

        v = wt(dx * dy) * get(fx, fy) + wt((1 - dx) * (1 - dy)) * get(cx, cy) \
            + wt(dx * (1 - dy)) * get(fx, cy) + wt((1 - dx) * dy) * get(cx, fy)
        outs.append(v)
    return torch.stack(outs)


def _conv(x, w, b, pad, relu):
    y = Fnn.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=pad).permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


def forward(feed: Dict[str, np.ndarray], Wt: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    dt = next(iter(Wt.values())).dtype
    imgs = torch.from_numpy(np.asarray(feed["imgs"], np.float32)).to(dt)
    net = _resize_legacy(imgs, 224, 224)
    taps = []
    for scope, n, _ in O.VGG_CFG:
        for j in range(1, n + 1):
            nm = "vgg_16/%s/%s_%d" % (scope, scope, j)
            net = _conv(net, Wt[nm + "/weights"], Wt[nm + "/biases"], 1, True)
        taps.append(net)
        net = Fnn.max_pool2d(net.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    net = _conv(net, Wt["vgg_16/fc6/weights"], Wt["vgg_16/fc6/biases"], 0, True)
    net = _conv(net, Wt["vgg_16/fc7/weights"], Wt["vgg_16/fc7/biases"], 0, True)
    emb = _conv(net, Wt["vgg_16/fc8/weights"], Wt["vgg_16/fc8/biases"], 0, False).reshape(imgs.shape[0], -1)
    xy = O.get_img_points(feed["sample_pc"], feed["trans_mat"])
    feat = torch.cat([_resampler(_resize_legacy(t, 137, 137), xy) for t in taps], dim=2)
    pc = torch.from_numpy(np.asarray(feed["sample_pc_rot"], np.float32)).to(dt)
    B, N, _ = pc.shape

    def mlp(scope, extra):
        h = pc
        for nm in ("fold1/conv1", "fold1/conv2", "fold1/conv3"):
            h = torch.relu(h @ Wt["%s/%s/weights" % (scope, nm)][0, 0] + Wt["%s/%s/biases" % (scope, nm)])
        h = torch.cat([h, extra], dim=2)
        for nm in ("fold2/conv1", "fold2/conv2"):
            h = torch.relu(h @ Wt["%s/%s/weights" % (scope, nm)][0, 0] + Wt["%s/%s/biases" % (scope, nm)])
        return h @ Wt["%s/fold2/conv5/weights" % scope][0, 0] + Wt["%s/fold2/conv5/biases" % scope]

    g = mlp("sdfprediction", emb[:, None, :].expand(B, N, emb.shape[1]))
    l = mlp("sdfprediction_imgfeat", feat)
    return g + l, {"embedding": emb, "feat": feat, "taps": taps}


def losses(pred: torch.Tensor, gt: torch.Tensor, Wt: Dict[str, torch.Tensor], sdf_weight=10.0, mask_weight=4.0):
    w = (gt <= 0.01).to(pred.dtype) * mask_weight + (gt > 0.01).to(pred.dtype)
    sdf_loss = ((gt * sdf_weight - pred).abs() * w).mean() * 1000
    reg = sum(WD * 0.5 * (v ** 2).sum() for k, v in Wt.items() if k.endswith("/weights"))
    out = {"accuracy": ((gt > 0) == (pred > 0)).to(pred.dtype).mean(),
           "sdf_loss_realvalue": (gt - pred / sdf_weight).abs().mean(),
           "sdf_loss": sdf_loss, "regularization": reg, "overall_loss": sdf_loss + reg}
    return out


def loss_and_grads(feed: Dict[str, np.ndarray], weights: Dict[str, np.ndarray], dtype=np.float64):
    """-> (losses dict of floats, grads dict name -> ndarray, pred ndarray)"""
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    Wt = {k: torch.tensor(np.asarray(v), dtype=tdt, requires_grad=True) for k, v in weights.items()}
    pred, _ = forward(feed, Wt)
    gt = torch.from_numpy(np.asarray(feed["sdf"], np.float32)).to(tdt)
    L = losses(pred, gt, Wt)
    L["overall_loss"].backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(v.shape)) for k, v in Wt.items()}
    return {k: float(v.detach()) for k, v in L.items()}, grads, pred.detach().numpy()


def learning_rate(step: int, batch_size: int, base=1e-4, decay_step=200000, decay_rate=0.9) -> float:
    return max(base * decay_rate ** ((step * batch_size) // decay_step), 1e-6)


def adam_step(w, g, m, v, t: int, lr: float, beta1=0.5, beta2=0.999, eps=1e-8):
    """one TF-style Adam update on numpy arrays (float64 recommended); t = 1 for the first step"""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v
