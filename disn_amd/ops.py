"""Thin torch-tensor wrappers over the C ABI (include/disn_amd.h).

PyTorch is plumbing here: it owns device memory and the stream; every
computation is a hand-written HIP kernel behind ``libdisn_amd.so``.  All
tensors are float32, contiguous, on a CUDA(=HIP) device; kernels are enqueued
on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import MlpWeights, VggWeights, check, lib

FEAT_DIM = 1472
IMG = 137
TAP_SHAPES = ((224, 64), (112, 128), (56, 256), (28, 512), (14, 512))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA tensor (the HIP path has no CPU fallback)" % name)
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def pack_kn(w_kn: torch.Tensor, kpad: Optional[int] = None) -> torch.Tensor:
    """[K,N] -> MFMA B-fragment order (disn_pack_kn)."""
    w_kn = _chk(w_kn, "w_kn")
    K, N = w_kn.shape
    kpad = kpad or ((K + 31) // 32) * 32
    out = torch.empty(kpad * N, dtype=torch.float32, device=w_kn.device)
    check("disn_pack_kn", lib().disn_pack_kn(w_kn.data_ptr(), K, N, kpad, out.data_ptr(), _stream()))
    return out


def pack_kn_x3(w_kn: torch.Tensor) -> torch.Tensor:
    """[K,N] -> the three-term bf16 image (disn_pack_kn_x3) of the fp32-accurate bf16-MFMA path"""
    w_kn = _chk(w_kn, "w_kn")
    K, N = w_kn.shape
    out = torch.empty(lib().disn_pack_kn_x3_bytes(K, N), dtype=torch.uint8, device=w_kn.device)
    check("disn_pack_kn_x3", lib().disn_pack_kn_x3(w_kn.data_ptr(), K, N, out.data_ptr(), _stream()))
    return out


def resize_bilinear(x: torch.Tensor, out_h: int, out_w: int, out: Optional[torch.Tensor] = None,
                    out_coff: int = 0) -> torch.Tensor:
    """tf.image.resize_bilinear (legacy) on NHWC."""
    x = _chk(x, "x")
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, out_h, out_w, Cc), dtype=torch.float32, device=x.device)
    cstride = out.shape[-1]
    check("disn_resize_bilinear", lib().disn_resize_bilinear(
        x.data_ptr(), B, H, W, Cc, out.data_ptr(), out_h, out_w, cstride, out_coff, _stream()))
    return out


def conv3x3(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool = True,
            plan: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
    """plan = (BM, BN, workgroups): run under an explicit GEMM plan (disn_conv3x3_planned)"""
    x = _chk(x, "x")
    B, H, W, Cin = x.shape
    out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    if plan is not None:
        bm, bn, wgs = plan
        ws = _ws(lib().disn_conv3x3_planned_workspace_bytes(B, H, W, Cin, cout, bm, bn, wgs), x.device)
        check("disn_conv3x3_planned", lib().disn_conv3x3_planned(
            x.data_ptr(), B, H, W, Cin, w_packed.data_ptr(), bias.data_ptr(), cout, int(relu), out.data_ptr(),
            ws.data_ptr(), ws.numel(), bm, bn, wgs, _stream()))
        return out
    nb = lib().disn_conv3x3_workspace_bytes(B, H, W, Cin, cout)
    ws = _ws(nb, x.device)
    check("disn_conv3x3", lib().disn_conv3x3(x.data_ptr(), B, H, W, Cin, w_packed.data_ptr(),
                                             bias.data_ptr(), cout, int(relu), out.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _stream()))
    return out


def conv3x3_x3(x: torch.Tensor, w_x3: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool = True,
               ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """disn_conv3x3 through the three-term bf16 image (pack_kn_x3): fp32-accurate, bf16 MFMA pipes"""
    x = _chk(x, "x")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    if ws is None:
        ws = _ws(lib().disn_conv3x3_x3_workspace_bytes(B, H, W, Cin, cout), x.device)
    check("disn_conv3x3_x3", lib().disn_conv3x3_x3(x.data_ptr(), B, H, W, Cin, w_x3.data_ptr(), bias.data_ptr(),
                                                   cout, int(relu), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   _stream()))
    return out


def conv1_1(x: torch.Tensor, w_hwio: torch.Tensor, bias: torch.Tensor, relu: bool = True, want_amax: bool = False):
    """disn_conv1_1: [B,H,W,3] x TF [3,3,3,64] -> [B,H,W,64] (direct fp32 FMA), optionally also max |out|"""
    x, w = _chk(x, "x"), _chk(w_hwio, "w_hwio")
    B, H, W, Cin = x.shape
    if Cin != 3 or w.numel() != 27 * 64:
        raise ValueError("conv1_1: expected 3 -> 64 channels")
    out = torch.empty((B, H, W, 64), dtype=torch.float32, device=x.device)
    amax = torch.zeros(1, dtype=torch.float32, device=x.device) if want_amax else None
    ws = _ws(lib().disn_conv1_1_workspace_bytes(), x.device)
    check("disn_conv1_1", lib().disn_conv1_1(x.data_ptr(), B, H, W, w.data_ptr(), bias.data_ptr(), int(relu),
                                             out.data_ptr(), amax.data_ptr() if want_amax else None, ws.data_ptr(),
                                             ws.numel(), _stream()))
    return (out, amax) if want_amax else out


def pack_conv_h2(w_hwio: torch.Tensor) -> torch.Tensor:
    """disn_pack_conv_h2: TF HWIO [3,3,Cin,Cout] (or [9*Cin,Cout]) -> the two-term f16 weight image of conv_h2.hip"""
    w = _chk(w_hwio, "w_hwio")
    cout = w.shape[-1]
    cin = w.numel() // (9 * cout)
    nbytes = lib().disn_pack_conv_h2_bytes(cin, cout)
    if nbytes == 0:
        raise ValueError("pack_conv_h2: Cin and Cout must be multiples of 64, got %d, %d" % (cin, cout))
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check("disn_pack_conv_h2", lib().disn_pack_conv_h2(w.data_ptr(), cin, cout, out.data_ptr(), _stream()))
    return out


def conv_h2_gain_span(image: torch.Tensor, cin: int, cout: int):
    """disn_conv_h2_gain_span -> (span_log2, warned): the channel-gain span of the variable a conv_h2 image was packed from"""
    span = C.c_float(0.0)
    rc = lib().disn_conv_h2_gain_span(image.data_ptr(), int(cin), int(cout), C.byref(span), _stream())
    if rc not in (0, 1):
        check("disn_conv_h2_gain_span", rc)
    return float(span.value), rc == 1


def conv3x3_h2(x: torch.Tensor, image: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool = True,
               pool: bool = False, want_amax: bool = False, tiling: int = 0,
               out: Optional[torch.Tensor] = None):
    """disn_conv3x3_h2 -> out, or (out, pooled, amax) with the entries asked for"""
    x = _chk(x, "x")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    pooled = torch.empty((B, H // 2, W // 2, cout), dtype=torch.float32, device=x.device) if pool else None
    amax = torch.zeros(1, dtype=torch.float32, device=x.device) if want_amax else None
    ws = _ws(lib().disn_conv3x3_h2_workspace_bytes(B), x.device)
    check("disn_conv3x3_h2", lib().disn_conv3x3_h2(
        x.data_ptr(), B, H, W, Cin, image.data_ptr(), bias.data_ptr(), cout, int(relu), out.data_ptr(),
        pooled.data_ptr() if pool else None, amax.data_ptr() if want_amax else None, int(tiling), ws.data_ptr(),
        ws.numel(), _stream()))
    if not pool and not want_amax:
        return out
    return out, pooled, amax


def maxpool2x2(x: torch.Tensor) -> torch.Tensor:
    x = _chk(x, "x")
    B, H, W, Cc = x.shape
    out = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.float32, device=x.device)
    check("disn_maxpool2x2", lib().disn_maxpool2x2(x.data_ptr(), B, H, W, Cc, out.data_ptr(), _stream()))
    return out


def fc(x: torch.Tensor, w_kn: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    x = _chk(x, "x")
    B, K = x.shape
    N = w_kn.shape[-1]
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    ws = _ws(lib().disn_fc_workspace_bytes(B, K, N), x.device)
    check("disn_fc", lib().disn_fc(x.data_ptr(), B, K, w_kn.data_ptr(), bias.data_ptr(), N, int(relu),
                                   out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def pack_dense_h2(w_kn: torch.Tensor) -> torch.Tensor:
    """disn_pack_dense_h2: W [K,N] -> the two-term f16 weight image of dense_h2.hip"""
    w = _chk(w_kn, "w_kn")
    K, N = w.shape
    nbytes = lib().disn_pack_dense_h2_bytes(K, N)
    if nbytes == 0:
        raise ValueError("pack_dense_h2: K and N must be multiples of 64, got %d, %d" % (K, N))
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check("disn_pack_dense_h2", lib().disn_pack_dense_h2(w.data_ptr(), K, N, out.data_ptr(), _stream()))
    return out


def dense_h2(a1: torch.Tensor, image: torch.Tensor, bias: torch.Tensor, n_out: int, relu: bool = True,
             a2: Optional[torch.Tensor] = None, in_bias: Optional[torch.Tensor] = None, want_amax: bool = False,
             rows_per_image: int = 0, image_k: int = 0, k_begin: int = 0, add_in: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None):
    """disn_dense_h2: act(f([a1 | a2]) @ W + b), f = relu(. + in_bias) when in_bias is given.  rows_per_image > 0:
    rows image-major, one activation scale per image, in_bias [images, K]; four images and more (of a multiple of
    128 rows each) run the batched form (dense_h2w.hip)"""
    a1 = _chk(a1, "a1")
    M, k1 = a1.shape
    k2 = 0 if a2 is None else _chk(a2, "a2").shape[1]
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float32, device=a1.device)
    amax = torch.zeros(1, dtype=torch.float32, device=a1.device) if want_amax else None
    imgs = M // rows_per_image if rows_per_image > 0 else 1
    ws = _ws(lib().disn_dense_h2_workspace_bytes(imgs), a1.device)
    check("disn_dense_h2", lib().disn_dense_h2(
        a1.data_ptr(), k1, k1, a2.data_ptr() if a2 is not None else None, k2, k2,
        in_bias.data_ptr() if in_bias is not None else None, M, int(rows_per_image), image.data_ptr(), int(image_k),
        int(k_begin), add_in.data_ptr() if add_in is not None else None, bias.data_ptr(),
        n_out, int(relu), out.data_ptr(), amax.data_ptr() if want_amax else None, ws.data_ptr(), ws.numel(), _stream()))
    return (out, amax) if want_amax else out


def get_loss(pred: torch.Tensor, gt: torch.Tensor, sdf_weight: float, mask_weight: float,
             regularization: float = 0.0) -> torch.Tensor:
    """disn_get_loss -> 5 device floats {accuracy, sdf_loss_realvalue, sdf_loss, regularization, overall_loss}"""
    pred, gt = _chk(pred, "pred"), _chk(gt, "gt")
    if pred.numel() != gt.numel():
        raise ValueError("get_loss: pred and gt differ in size")
    out = torch.zeros(5, dtype=torch.float32, device=pred.device)
    out[3] = float(regularization)
    check("disn_get_loss", lib().disn_get_loss(pred.data_ptr(), gt.data_ptr(), pred.numel(), float(sdf_weight),
                                               float(mask_weight), out.data_ptr(), _stream()))
    return out


def fc_t(x: torch.Tensor, wt_nk: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    """disn_fc_t: act(x @ wt_nk.T + bias) from the transposed matrix [N][K], one launch"""
    x, wt = _chk(x, "x"), _chk(wt_nk, "wt_nk")
    B, K = x.shape
    N = wt.shape[0]
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    check("disn_fc_t", lib().disn_fc_t(x.data_ptr(), B, K, wt.data_ptr(), bias.data_ptr(), N, int(relu),
                                       out.data_ptr(), _stream()))
    return out


def dense(a1: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, n_out: int, relu: bool = True,
          a2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act([a1 | a2] @ W + b): the tf_util.conv2d [1,1] primitive on [M,K] rows."""
    a1 = _chk(a1, "a1")
    M, k1 = a1.shape
    k2 = 0
    if a2 is not None:
        a2 = _chk(a2, "a2")
        k2 = a2.shape[1]
    out = torch.empty((M, n_out), dtype=torch.float32, device=a1.device)
    ws = _ws(lib().disn_dense_workspace_bytes(M, k1 + k2, n_out), a1.device)
    check("disn_dense", lib().disn_dense(
        a1.data_ptr(), a1.stride(0), k1, a2.data_ptr() if a2 is not None else None,
        a2.stride(0) if a2 is not None else 0, k2, M, w_packed.data_ptr(), bias.data_ptr(), n_out,
        int(relu), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def vgg16_forward(w: VggWeights, img: torch.Tensor, ws: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, List[torch.Tensor], torch.Tensor]:
    """-> (resized224 [B,224,224,3], taps[5], embedding [B,num_classes])."""
    img = _chk(img, "img")
    B = img.shape[0]
    if tuple(img.shape[1:]) != (IMG, IMG, 3):
        raise ValueError("img must be [B,137,137,3] (models/model_normalization.py:249-250 hard-codes 137)")
    dev = img.device
    resized = torch.empty((B, 224, 224, 3), dtype=torch.float32, device=dev)
    taps = [torch.empty((B, hw, hw, ch), dtype=torch.float32, device=dev) for hw, ch in TAP_SHAPES]
    emb = torch.empty((B, w.num_classes), dtype=torch.float32, device=dev)
    need = lib().disn_vgg16_workspace_bytes(B)
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    tp = (C.c_void_p * 5)(*[t.data_ptr() for t in taps])
    check("disn_vgg16_forward", lib().disn_vgg16_forward(
        C.byref(w), img.data_ptr(), B, resized.data_ptr(), C.byref(tp), emb.data_ptr(), ws.data_ptr(),
        ws.numel(), _stream()))
    return resized, taps, emb


class ConvStackRun:
    """disn_vgg16_conv_stack with every buffer allocated once (bench.py times repeated calls of .run())"""

    def __init__(self, w: VggWeights, img: torch.Tensor, want_pool5: bool = True):
        self.w, self.img = w, _chk(img, "img")
        B, dev = img.shape[0], img.device
        self.B = B
        self.resized = torch.empty((B, 224, 224, 3), dtype=torch.float32, device=dev)
        self.taps = [torch.empty((B, hw, hw, ch), dtype=torch.float32, device=dev) for hw, ch in TAP_SHAPES]
        self.pool5 = torch.empty((B, 7, 7, 512), dtype=torch.float32, device=dev) if want_pool5 else None
        self.ws = _ws(lib().disn_vgg16_workspace_bytes(B), dev)
        self.tp = (C.c_void_p * 5)(*[t.data_ptr() for t in self.taps])

    def run(self) -> None:
        check("disn_vgg16_conv_stack", lib().disn_vgg16_conv_stack(
            C.byref(self.w), self.img.data_ptr(), self.B, self.resized.data_ptr(), C.byref(self.tp),
            self.pool5.data_ptr() if self.pool5 is not None else None, self.ws.data_ptr(), self.ws.numel(), _stream()))


def stream_create() -> int:
    """a non-blocking HIP stream on the current device (disn_stream_create) -> handle for torch.cuda.ExternalStream"""
    h = C.c_void_p()
    check("disn_stream_create", lib().disn_stream_create(C.byref(h)))
    return h.value


def stream_destroy(stream: int) -> None:
    if stream:
        check("disn_stream_destroy", lib().disn_stream_destroy(stream))


def ctx_create() -> int:
    """Concurrency context (aux HIP stream + events) on the current device."""
    h = C.c_void_p()
    check("disn_ctx_create", lib().disn_ctx_create(C.byref(h)))
    return h.value


def ctx_pipeline(ctx: int, wait_event: Optional[torch.cuda.Event], record_event: Optional[torch.cuda.Event]) -> None:
    """disn_ctx_pipeline: chain the convolution stacks of consecutive steps (events already created: .cuda_event)"""
    check("disn_ctx_pipeline", lib().disn_ctx_pipeline(ctx, wait_event.cuda_event if wait_event is not None else None,
                                                       record_event.cuda_event if record_event is not None else None))


def ctx_destroy(ctx: int) -> None:
    if ctx:
        check("disn_ctx_destroy", lib().disn_ctx_destroy(ctx))


def _alloc_encoder_outputs(B: int, num_classes: int, dev, featmap: bool = True):
    resized = torch.empty((B, 224, 224, 3), dtype=torch.float32, device=dev)
    taps = [torch.empty((B, hw, hw, ch), dtype=torch.float32, device=dev) for hw, ch in TAP_SHAPES]
    emb = torch.empty((B, num_classes), dtype=torch.float32, device=dev)
    fm = torch.empty((B, IMG, IMG, FEAT_DIM), dtype=torch.float32, device=dev) if featmap else None
    return resized, taps, emb, fm


def encode(ctx: Optional[int], w: VggWeights, img: torch.Tensor, ws: Optional[torch.Tensor] = None):
    """rows A, B, C, E -> (resized224, taps[5], embedding, featmap); the tap up-samples overlap the
    convolutions on the context's auxiliary stream."""
    img = _chk(img, "img")
    B = img.shape[0]
    if tuple(img.shape[1:]) != (IMG, IMG, 3):
        raise ValueError("img must be [B,137,137,3] (models/model_normalization.py:249-250 hard-codes 137)")
    resized, taps, emb, featmap = _alloc_encoder_outputs(B, w.num_classes, img.device)
    need = lib().disn_encode_workspace_bytes(B)
    if ws is None or ws.numel() < need:
        ws = _ws(need, img.device)
    tp = (C.c_void_p * 5)(*[t.data_ptr() for t in taps])
    check("disn_encode", lib().disn_encode(ctx, C.byref(w), img.data_ptr(), B, resized.data_ptr(), C.byref(tp),
                                           emb.data_ptr(), featmap.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return resized, taps, emb, featmap


def encode_query(ctx: int, vw: VggWeights, mw: MlpWeights, img: torch.Tensor, trans_mat: torch.Tensor,
                 pts: torch.Tensor, pts_rot: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                 keep_featmap: bool = True):
    """One full evaluation of the graph for pred_sdf (what one sess.run executes), B*N <= 65536.
    -> (resized224, taps, embedding, featmap, sdf [B,N]).  keep_featmap=False: the [B,137,137,1472]
    map is never written (featmap is None); the gather up-samples the taps at the pixels it touches,
    with the same expression -- sdf is bit-identical."""
    img, pts, trans_mat = _chk(img, "img"), _chk(pts, "pts"), _chk(trans_mat, "trans_mat")
    pts_rot = pts if pts_rot is None else _chk(pts_rot, "pts_rot")
    B, N = pts.shape[0], pts.shape[1]
    resized, taps, emb, featmap = _alloc_encoder_outputs(B, vw.num_classes, img.device, keep_featmap)
    sdf = torch.empty((B, N), dtype=torch.float32, device=img.device)
    need = lib().disn_encode_query_workspace_bytes(B, N)
    if need == 0:
        raise ValueError("encode_query needs B*N <= 65536")
    if ws is None or ws.numel() < need:
        ws = _ws(need, img.device)
    tp = (C.c_void_p * 5)(*[t.data_ptr() for t in taps])
    check("disn_encode_query", lib().disn_encode_query(
        ctx, C.byref(vw), C.byref(mw), img.data_ptr(), trans_mat.data_ptr(), pts.data_ptr(), pts_rot.data_ptr(),
        B, N, resized.data_ptr(), C.byref(tp), emb.data_ptr(), featmap.data_ptr() if keep_featmap else None,
        sdf.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return resized, taps, emb, featmap, sdf


def build_featmap(taps: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B = taps[0].shape[0]
    if out is None:
        out = torch.empty((B, IMG, IMG, FEAT_DIM), dtype=torch.float32, device=taps[0].device)
    tp = (C.c_void_p * 5)(*[_chk(t, "tap").data_ptr() for t in taps])
    check("disn_build_featmap", lib().disn_build_featmap(C.byref(tp), B, out.data_ptr(), _stream()))
    return out


def project(pts: torch.Tensor, trans_mat: torch.Tensor) -> torch.Tensor:
    pts, trans_mat = _chk(pts, "pts"), _chk(trans_mat, "trans_mat")
    B, N, _ = pts.shape
    xy = torch.empty((B, N, 2), dtype=torch.float32, device=pts.device)
    check("disn_project", lib().disn_project(pts.data_ptr(), trans_mat.data_ptr(), B, N, xy.data_ptr(),
                                             _stream()))
    return xy


def gather(featmap: torch.Tensor, xy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    featmap, xy = _chk(featmap, "featmap"), _chk(xy, "xy")
    B, N, _ = xy.shape
    if out is None:
        out = torch.empty((B, N, FEAT_DIM), dtype=torch.float32, device=xy.device)
    check("disn_gather", lib().disn_gather(featmap.data_ptr(), xy.data_ptr(), B, N, out.data_ptr(),
                                           _stream()))
    return out


def scale_channels(x: torch.Tensor, scale: torch.Tensor, invert: bool = False,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., C] * scale [C] (or / scale): equalised <-> true units of taps / gathered features
    (disn_scale_channels; exact for the powers of two of disn_equalise_weights)"""
    x, scale = _chk(x, "x"), _chk(scale, "scale")
    Cc = x.shape[-1]
    if scale.numel() != Cc:
        raise ValueError("scale_channels: %d factors for %d channels" % (scale.numel(), Cc))
    if out is None:
        out = torch.empty_like(x)
    check("disn_scale_channels", lib().disn_scale_channels(x.data_ptr(), x.numel() // Cc, Cc, scale.data_ptr(),
                                                           int(bool(invert)), out.data_ptr(), _stream()))
    return out


def gather_taps(taps: Sequence[torch.Tensor], trans_mat: torch.Tensor, pts: torch.Tensor,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rows D + E + F without the feature map (disn_gather_taps): taps x pts [B,N,3] -> feat [B,N,1472]"""
    pts = _chk(pts, "pts")
    B, N, _ = pts.shape
    if out is None:
        out = torch.empty((B, N, FEAT_DIM), dtype=torch.float32, device=pts.device)
    arr = (C.c_void_p * 5)(*[_chk(t, "tap").data_ptr() for t in taps])
    check("disn_gather_taps", lib().disn_gather_taps(C.byref(arr), _chk(trans_mat, "trans_mat").data_ptr(),
                                                     pts.data_ptr(), B, N, out.data_ptr(), _stream()))
    return out


def gather_taps_split(taps: Sequence[torch.Tensor], trans_mat: torch.Tensor, pts: torch.Tensor,
                      feat_amax: torch.Tensor) -> torch.Tensor:
    """disn_gather_taps_split: the rows of gather_taps in split form -> uint8 [B, N, 6144] ([h8 | l8] per 8 channels of
    feature * the image's power-of-two scale from feat_amax [B])"""
    pts = _chk(pts, "pts")
    B, N, _ = pts.shape
    out = torch.empty((B, N, 1536 * 4), dtype=torch.uint8, device=pts.device)
    arr = (C.c_void_p * 5)(*[_chk(t, "tap").data_ptr() for t in taps])
    check("disn_gather_taps_split", lib().disn_gather_taps_split(
        C.byref(arr), _chk(trans_mat, "trans_mat").data_ptr(), pts.data_ptr(), B, N, _chk(feat_amax, "feat_amax").data_ptr(),
        out.data_ptr(), _stream()))
    return out


def gather_fold(pmap_b: torch.Tensor, trans_mat_b: torch.Tensor, pts: torch.Tensor, pre: torch.Tensor,
                bias: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(pre + resample(pmap_b)(project(pts)) + bias) for N points of one image (disn_gather_fold)"""
    pts, pre = _chk(pts, "pts"), _chk(pre, "pre")
    N = pts.shape[0]
    if out is None:
        out = torch.empty((N, 512), dtype=torch.float32, device=pts.device)
    check("disn_gather_fold", lib().disn_gather_fold(_chk(pmap_b, "pmap").data_ptr(),
                                                     _chk(trans_mat_b, "trans_mat").data_ptr(), pts.data_ptr(), N,
                                                     pre.data_ptr(), _chk(bias, "bias").data_ptr(),
                                                     out.data_ptr(), _stream()))
    return out


def mlp_fused_pack(w2: torch.Tensor, w3: torch.Tensor, w4_point: torch.Tensor, w5: torch.Tensor) -> torch.Tensor:
    """fused-kernel weight image of one MLP stream (disn_mlp_fused_pack) from its four MFMA-shaped layers,
    TF [K][N]: fold1/conv2 [64,256], fold1/conv3 [256,512], fold2/conv1 point rows [512,512], fold2/conv2 [512,256]"""
    ws = [_chk(t, "w") for t in (w2, w3, w4_point, w5)]
    for t, shp in zip(ws, ((64, 256), (256, 512), (512, 512), (512, 256))):
        if tuple(t.shape) != shp:
            raise ValueError("mlp_fused_pack: expected %s, got %s" % (shp, tuple(t.shape)))
    out = torch.empty(lib().disn_mlp_fused_image_bytes(), dtype=torch.uint8, device=ws[0].device)
    check("disn_mlp_fused_pack", lib().disn_mlp_fused_pack(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                                           ws[3].data_ptr(), out.data_ptr(), _stream()))
    return out


def mlp_fused_feat_pack(w2: torch.Tensor, w3: torch.Tensor, w4: torch.Tensor, w5: torch.Tensor) -> torch.Tensor:
    """FEAT-form weight image of the LOCAL stream (disn_mlp_fused_feat_pack): as mlp_fused_pack with w4 = the whole
    fold2/conv1 matrix [512 + 1472, 512] (the gathered features are 96 extra reduction blocks of that layer)"""
    ws = [_chk(t, "w") for t in (w2, w3, w4, w5)]
    for t, shp in zip(ws, ((64, 256), (256, 512), (1984, 512), (512, 256))):
        if tuple(t.shape) != shp:
            raise ValueError("mlp_fused_feat_pack: expected %s, got %s" % (shp, tuple(t.shape)))
    out = torch.empty(lib().disn_mlp_fused_feat_image_bytes(), dtype=torch.uint8, device=ws[0].device)
    check("disn_mlp_fused_feat_pack", lib().disn_mlp_fused_feat_pack(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                                                     ws[3].data_ptr(), out.data_ptr(), _stream()))
    return out


def query_taps_fused(w: MlpWeights, taps: Sequence[torch.Tensor], embedding: torch.Tensor, trans_mat: torch.Tensor,
                     pts: torch.Tensor, pts_rot: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """disn_query_taps_fused: rows D..H of B images x N points (any N: padded to a multiple of 128 inside the library;
    B * padded N <= 65536) from the five taps through the fused small-set kernels (split-form gather + one launch per
    MLP stream)"""
    pts = _chk(pts, "pts")
    pts_rot = pts if pts_rot is None else _chk(pts_rot, "pts_rot")
    B, N, _ = pts.shape
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
    need = lib().disn_query_taps_fused_workspace_bytes(B, N)
    if need == 0:
        raise ValueError("query_taps_fused: B * N (N rounded up to a multiple of 128) must be <= 65536, got B %d N %d" % (B, N))
    if ws is None or ws.numel() < need:
        ws = _ws(need, pts.device)
    tp = (C.c_void_p * 5)(*[_chk(t, "tap").data_ptr() for t in taps])
    check("disn_query_taps_fused", lib().disn_query_taps_fused(
        C.byref(w), C.byref(tp), _chk(embedding, "embedding").data_ptr(), _chk(trans_mat, "trans_mat").data_ptr(),
        pts.data_ptr(), pts_rot.data_ptr(), B, N, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def amax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max |x| as a 1-element device tensor (disn_amax); x.numel() % 4 == 0"""
    x = _chk(x, "x")
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    check("disn_amax", lib().disn_amax(x.data_ptr(), x.numel(), out.data_ptr(), _stream()))
    return out


def query_fused(w: MlpWeights, pmap: torch.Tensor, pmap_amax: torch.Tensor, embedding: torch.Tensor,
                trans_mat: torch.Tensor, pts: torch.Tensor, pts_rot: Optional[torch.Tensor] = None,
                ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """query_folded() through the fused point-MLP kernels; pmap [B,137*137,512], pmap_amax [B]"""
    pts = _chk(pts, "pts")
    pts_rot = pts if pts_rot is None else _chk(pts_rot, "pts_rot")
    B, N, _ = pts.shape
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
    need = lib().disn_query_fused_workspace_bytes(B, N)
    if ws is None or ws.numel() < need:
        ws = _ws(need, pts.device)
    check("disn_query_fused", lib().disn_query_fused(
        C.byref(w), _chk(pmap, "pmap").data_ptr(), _chk(pmap_amax, "pmap_amax").data_ptr(),
        _chk(embedding, "embedding").data_ptr(), _chk(trans_mat, "trans_mat").data_ptr(), pts.data_ptr(),
        pts_rot.data_ptr(), B, N, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def query_grid_fused(w: MlpWeights, pmap: torch.Tensor, pmap_amax: torch.Tensor, embedding: torch.Tensor,
                     trans_mat: torch.Tensor, sdf_params, res: int, k0: int, k1: int, sdf_weight: float = 10.0,
                     ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """disn_query_grid_fused: grid points k0..k1-1 of one image through the fused kernels, one launch per stream"""
    dev = embedding.device
    n = k1 - k0
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    need = lib().disn_query_grid_fused_workspace_bytes(n)
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    p6 = _params6(sdf_params)
    check("disn_query_grid_fused", lib().disn_query_grid_fused(
        C.byref(w), _chk(pmap, "pmap").data_ptr(), _chk(pmap_amax, "pmap_amax").data_ptr(),
        _chk(embedding, "embedding").data_ptr(), _chk(trans_mat, "trans_mat").data_ptr(), C.byref(p6), res,
        k0, k1, float(sdf_weight), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def sdf_mlp(w: MlpWeights, pts_rot: torch.Tensor, embedding: torch.Tensor, feat: torch.Tensor,
            want_streams: bool = False):
    pts_rot, embedding, feat = _chk(pts_rot, "pts_rot"), _chk(embedding, "embedding"), _chk(feat, "feat")
    B, N, _ = pts_rot.shape
    dev = pts_rot.device
    sdf = torch.empty((B, N), dtype=torch.float32, device=dev)
    g = torch.empty((B, N), dtype=torch.float32, device=dev) if want_streams else None
    l = torch.empty((B, N), dtype=torch.float32, device=dev) if want_streams else None
    ws = _ws(lib().disn_sdf_mlp_workspace_bytes(B, N), dev)
    check("disn_sdf_mlp", lib().disn_sdf_mlp(
        C.byref(w), pts_rot.data_ptr(), embedding.data_ptr(), feat.data_ptr(), B, N, sdf.data_ptr(),
        g.data_ptr() if want_streams else None, l.data_ptr() if want_streams else None, ws.data_ptr(),
        ws.numel(), _stream()))
    return (sdf, g, l) if want_streams else sdf


def query(w: MlpWeights, featmap: torch.Tensor, embedding: torch.Tensor, trans_mat: torch.Tensor,
          pts: torch.Tensor, pts_rot: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    pts = _chk(pts, "pts")
    pts_rot = pts if pts_rot is None else _chk(pts_rot, "pts_rot")
    B, N, _ = pts.shape
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
    need = lib().disn_query_workspace_bytes(B, N)
    if ws is None or ws.numel() < need:
        ws = _ws(need, pts.device)
    check("disn_query", lib().disn_query(
        C.byref(w), _chk(featmap, "featmap").data_ptr(), _chk(embedding, "embedding").data_ptr(),
        _chk(trans_mat, "trans_mat").data_ptr(), pts.data_ptr(), pts_rot.data_ptr(), B, N,
        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


MAP_PIXELS = IMG * IMG


def fold_local(w: MlpWeights, featmap_b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pmap [137*137, 512] = featmap of ONE image [137,137,1472] times the 1472 feature rows of the local
    fold2/conv1 (disn_fold_local): what the *_folded queries gather from instead of the feature map."""
    featmap_b = _chk(featmap_b, "featmap")
    if featmap_b.numel() != MAP_PIXELS * FEAT_DIM:
        raise ValueError("fold_local takes the feature map of one image")
    if out is None:
        out = torch.empty((MAP_PIXELS, 512), dtype=torch.float32, device=featmap_b.device)
    ws = _ws(lib().disn_fold_local_workspace_bytes(), featmap_b.device)
    check("disn_fold_local", lib().disn_fold_local(C.byref(w), featmap_b.data_ptr(), out.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), _stream()))
    return out


def query_folded(w: MlpWeights, pmap: torch.Tensor, embedding: torch.Tensor, trans_mat: torch.Tensor,
                 pts: torch.Tensor, pts_rot: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """query() with pmap [B,137*137,512] (fold_local per image) in place of the feature map"""
    pts = _chk(pts, "pts")
    pts_rot = pts if pts_rot is None else _chk(pts_rot, "pts_rot")
    B, N, _ = pts.shape
    if pmap.numel() != B * MAP_PIXELS * 512:
        raise ValueError("pmap must be [B,137*137,512]")
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
    need = lib().disn_query_workspace_bytes(B, N)
    if ws is None or ws.numel() < need:
        ws = _ws(need, pts.device)
    check("disn_query_folded", lib().disn_query_folded(
        C.byref(w), _chk(pmap, "pmap").data_ptr(), _chk(embedding, "embedding").data_ptr(),
        _chk(trans_mat, "trans_mat").data_ptr(), pts.data_ptr(), pts_rot.data_ptr(), B, N,
        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def _params6(sdf_params) -> C.Array:
    vals = [float(v) for v in sdf_params]
    if len(vals) != 6:
        raise ValueError("sdf_params must have 6 entries")
    return (C.c_double * 6)(*vals)


def grid_points(sdf_params, res: int, k0: int, k1: int, device) -> torch.Tensor:
    pts = torch.empty((k1 - k0, 3), dtype=torch.float32, device=device)
    p6 = _params6(sdf_params)
    with torch.cuda.device(pts.device):
        check("disn_grid_points", lib().disn_grid_points(C.byref(p6), res, k0, k1, pts.data_ptr(), _stream()))
    return pts


def query_grid(w: MlpWeights, featmap: torch.Tensor, embedding: torch.Tensor, trans_mat: torch.Tensor,
               sdf_params, res: int, k0: int, k1: int, sdf_weight: float = 10.0,
               ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               ctx: Optional[int] = None, pmap: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SDF of grid points k0..k1-1 of ONE image (featmap [137,137,1472] or [1,...]).  With a
    context the chunks are pipelined over two streams (gather of chunk i+1 under the MLP of chunk i).
    With ``pmap`` (fold_local of that image) the folded local stream runs and featmap is not read."""
    dev = pmap.device if pmap is not None else featmap.device
    if out is None:
        out = torch.empty((k1 - k0,), dtype=torch.float32, device=dev)
    p6 = _params6(sdf_params)
    if pmap is not None:
        need = lib().disn_query_grid_workspace_bytes(k1 - k0)
        if ws is None or ws.numel() < need:
            ws = _ws(need, dev)
        check("disn_query_grid_folded", lib().disn_query_grid_folded(
            C.byref(w), _chk(pmap, "pmap").data_ptr(), _chk(embedding, "embedding").data_ptr(),
            _chk(trans_mat, "trans_mat").data_ptr(), C.byref(p6), res, k0, k1, float(sdf_weight),
            out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        return out
    if ctx:
        need = lib().disn_query_grid_ctx_workspace_bytes(k1 - k0)
        if ws is None or ws.numel() < need:
            ws = _ws(need, dev)
        check("disn_query_grid_ctx", lib().disn_query_grid_ctx(
            ctx, C.byref(w), _chk(featmap, "featmap").data_ptr(), _chk(embedding, "embedding").data_ptr(),
            _chk(trans_mat, "trans_mat").data_ptr(), C.byref(p6), res, k0, k1, float(sdf_weight),
            out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        return out
    need = lib().disn_query_grid_workspace_bytes(k1 - k0)
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    check("disn_query_grid", lib().disn_query_grid(
        C.byref(w), _chk(featmap, "featmap").data_ptr(), _chk(embedding, "embedding").data_ptr(),
        _chk(trans_mat, "trans_mat").data_ptr(), C.byref(p6), res, k0, k1, float(sdf_weight),
        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


# ---------------------------------------------------------------------------
# training step
# ---------------------------------------------------------------------------
def param_layout() -> _lib.ParamLayout:
    """offsets / counts (in floats) of the 56 variables in the flat parameter buffer"""
    L = _lib.ParamLayout()
    check("disn_param_layout", lib().disn_param_layout(C.byref(L)))
    return L


def dense_backward(a: torch.Tensor, w_kn: torch.Tensor, y: Optional[torch.Tensor], dy: torch.Tensor,
                   wd: float = 0.0, need_da: bool = True, compute_bf16: bool = False):
    """backward of out = act(a @ w + b): -> (da or None, dw, db); dy is masked IN PLACE when y is given"""
    a, w_kn, dy = _chk(a, "a"), _chk(w_kn, "w_kn"), _chk(dy, "dy")
    M, K = a.shape
    N = w_kn.shape[1]
    dev = a.device
    da = torch.empty((M, K), dtype=torch.float32, device=dev) if need_da else None
    dw = torch.empty((K, N), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev)
    ws = _ws(lib().disn_dense_backward_workspace_bytes(M, K, N), dev)
    check("disn_dense_backward", lib().disn_dense_backward(
        a.data_ptr(), K, K, w_kn.data_ptr(), _chk(y, "y").data_ptr() if y is not None else None,
        dy.data_ptr(), M, N, float(wd), int(compute_bf16), da.data_ptr() if need_da else None, dw.data_ptr(),
        db.data_ptr(),
        ws.data_ptr(), ws.numel(), _stream()))
    return da, dw, db


def conv3x3_backward(x: torch.Tensor, w_hwio: torch.Tensor, y: Optional[torch.Tensor], dy: torch.Tensor,
                     wd: float = 0.0, need_dx: bool = True, compute_bf16: bool = False):
    """backward of a SAME 3x3 conv (+ReLU when y is given): -> (dx or None, dw [3,3,Cin,Cout], db)"""
    x, w_hwio, dy = _chk(x, "x"), _chk(w_hwio, "w_hwio"), _chk(dy, "dy")
    B, H, W, Cin = x.shape
    Cout = w_hwio.shape[-1]
    dev = x.device
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device=dev)
    db = torch.empty((Cout,), dtype=torch.float32, device=dev)
    ws = _ws(lib().disn_conv3x3_backward_workspace_bytes(B, H, W, Cin, Cout), dev)
    check("disn_conv3x3_backward", lib().disn_conv3x3_backward(
        x.data_ptr(), B, H, W, Cin, w_hwio.data_ptr(), _chk(y, "y").data_ptr() if y is not None else None,
        dy.data_ptr(), Cout, float(wd), int(compute_bf16), dx.data_ptr() if need_dx else None, dw.data_ptr(),
        db.data_ptr(),
        ws.data_ptr(), ws.numel(), _stream()))
    return dx, dw, db


def maxpool2x2_backward(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    x, dy = _chk(x, "x"), _chk(dy, "dy")
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    check("disn_maxpool2x2_backward", lib().disn_maxpool2x2_backward(
        x.data_ptr(), dy.data_ptr(), B, H, W, Cc, dx.data_ptr(), _stream()))
    return dx


def resize_bilinear_backward(dout: torch.Tensor, in_h: int, in_w: int, channels: Optional[int] = None,
                             out_coff: int = 0, din: Optional[torch.Tensor] = None,
                             accumulate: bool = False) -> torch.Tensor:
    dout = _chk(dout, "dout")
    B, Ho, Wo, cs = dout.shape
    Cc = channels or cs
    if din is None:
        din = torch.empty((B, in_h, in_w, Cc), dtype=torch.float32, device=dout.device)
        accumulate = False
    ws = _ws(lib().disn_resize_bilinear_backward_workspace_bytes(B, in_h, in_w, Cc, Ho, Wo), dout.device)
    check("disn_resize_bilinear_backward", lib().disn_resize_bilinear_backward(
        dout.data_ptr(), B, in_h, in_w, Cc, Ho, Wo, cs, out_coff, din.data_ptr(), int(accumulate),
        ws.data_ptr(), ws.numel(), _stream()))
    return din


def gather_backward(dfeat: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    dfeat, xy = _chk(dfeat, "dfeat"), _chk(xy, "xy")
    B, N, _ = dfeat.shape
    dmap = torch.empty((B, IMG, IMG, FEAT_DIM), dtype=torch.float32, device=dfeat.device)
    check("disn_gather_backward", lib().disn_gather_backward(dfeat.data_ptr(), xy.data_ptr(), B, N,
                                                             dmap.data_ptr(), _stream()))
    return dmap


def train_step(params: torch.Tensor, grads: torch.Tensor, img: torch.Tensor, trans_mat: torch.Tensor,
               pts: torch.Tensor, pts_rot: torch.Tensor, gt: torch.Tensor, wd: float = 1e-5,
               sdf_weight: float = 10.0, mask_weight: float = 4.0, ws: Optional[torch.Tensor] = None,
               ctx: Optional[int] = None, head_ready: Optional[torch.cuda.Event] = None,
               compute_bf16: bool = False):
    """forward + get_loss + gradients into `grads`: -> (pred [B,N], losses [5] device tensor).
    ctx: concurrency context (ctx_create); head_ready: a torch.cuda.Event (already recorded once, so
    that its handle exists) recorded when the fc/MLP part of `grads` is final.
    compute_bf16: 0 f32-input MFMA everywhere; 1 bf16 multiply (mixed precision); 2 fp32-accurate
    three-term bf16 split for the forward / data-gradient GEMMs."""
    B, N = pts.shape[0], pts.shape[1]
    dev = params.device
    need = lib().disn_train_workspace_bytes(B, N)
    if need == 0:
        raise ValueError("unsupported training shape B=%d N=%d (B*N <= 65536)" % (B, N))
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    pred = torch.empty((B, N), dtype=torch.float32, device=dev)
    losses = torch.empty((5,), dtype=torch.float32, device=dev)
    check("disn_train_step", lib().disn_train_step(
        ctx, _chk(params, "params").data_ptr(), _chk(grads, "grads").data_ptr(), _chk(img, "img").data_ptr(),
        _chk(trans_mat, "trans_mat").data_ptr(), _chk(pts, "pts").data_ptr(),
        _chk(pts_rot, "pts_rot").data_ptr(), _chk(gt, "gt").data_ptr(), B, N, float(wd),
        float(sdf_weight), float(mask_weight), int(compute_bf16), pred.data_ptr(), losses.data_ptr(),
        head_ready.cuda_event if head_ready is not None else None, ws.data_ptr(), ws.numel(), _stream()))
    return pred, losses


def adam_update(params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr_t: float,
                beta1: float = 0.5, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0) -> None:
    check("disn_adam_update", lib().disn_adam_update(
        _chk(params, "params").data_ptr(), _chk(grads, "grads").data_ptr(), _chk(m, "m").data_ptr(),
        _chk(v, "v").data_ptr(), params.numel(), float(lr_t), float(beta1), float(beta2), float(eps),
        float(grad_scale), _stream()))


def dense_bf16(a1: torch.Tensor, w_kn: torch.Tensor, bias: torch.Tensor, relu: bool = True,
               a2: Optional[torch.Tensor] = None, nsplit: int = 1) -> torch.Tensor:
    """act([a1|a2] @ w_kn + bias) with the multiply in bf16 (raw fp32 weights, packed on the fly)"""
    a1, w_kn = _chk(a1, "a1"), _chk(w_kn, "w_kn")
    M, k1 = a1.shape
    k2 = a2.shape[1] if a2 is not None else 0
    N = w_kn.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=a1.device)
    ws = _ws(lib().disn_dense_bf16_workspace_bytes(M, k1 + k2, N), a1.device)
    check("disn_dense_bf16", lib().disn_dense_bf16(
        a1.data_ptr(), k1, k1, _chk(a2, "a2").data_ptr() if a2 is not None else None, k2, k2, M,
        w_kn.data_ptr(), _chk(bias, "bias").data_ptr(), N, int(relu), int(nsplit), out.data_ptr(), ws.data_ptr(),
        ws.numel(), _stream()))
    return out


def conv3x3_bf16(x: torch.Tensor, w_hwio: torch.Tensor, bias: torch.Tensor, relu: bool = True,
                 nsplit: int = 1) -> torch.Tensor:
    x, w_hwio = _chk(x, "x"), _chk(w_hwio, "w_hwio")
    B, H, W, Cin = x.shape
    Cout = w_hwio.shape[-1]
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    ws = _ws(lib().disn_conv3x3_bf16_workspace_bytes(B, H, W, Cin, Cout), x.device)
    check("disn_conv3x3_bf16", lib().disn_conv3x3_bf16(
        x.data_ptr(), B, H, W, Cin, w_hwio.data_ptr(), _chk(bias, "bias").data_ptr(), Cout, int(relu),
        int(nsplit), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out
