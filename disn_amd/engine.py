"""Device-resident engine for the SDF query path: packed weights + encoder/decoder split.

The reference re-runs the whole graph (resize, VGG-16, five 137x137 up-samples) on every
chunk ``sess.run`` (test/create_sdf.py:262-276) although it also defines -- and never calls --
an encoder/decoder split (``get_decoder`` / ``placeholder_features``,
models/model_normalization.py:38-45,223-238).  Here that split is the primary structure:
``encode()`` runs rows A, B, C, E once per image, ``query*()`` run rows D, F, G, H per point.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import time

import numpy as np
import torch

from . import ops
from ._lib import MLP_FIELDS, MlpWeights, VggWeights
from .weights import MLP_SCOPES, VGG_CONV_NAMES, WeightStore


class DeviceWeights:
    """Uploads a WeightStore and re-packs the GEMM-shaped layers for the MFMA kernels."""

    def __init__(self, store: WeightStore, device: torch.device, conv_h2: bool = True, equalise: bool = True):
        """``equalise`` (default): upload the EQUALISED copy of the variables (WeightStore.equalised =
        disn_equalise_weights: every hidden channel times a power of two, its consumers' rows divided by it -- the same
        function, the same fp32 roundings, but channels of comparable magnitude, which the per-image operand scale of the
        two-term f16 kernels needs on trained weights: DESIGN 4k).  Taps, the feature map and gathered features of an
        engine built this way are in equalised units; ``tap_scale`` [1472] converts (SdfEngine.true_taps /
        true_features / internal_features; the model_normalization surface does it for its end_points).
        ``status``: what was done -- {'equalised', 'channel_gain_span_log2' (per layer, BEFORE equalisation),
        'max_span_log2', 'packed_gain_span_log2' (per two-term conv image, of what was packed), 'gain_span_warnings'
        (layers whose packed span exceeds 12 binades: only with equalise=False on a checkpoint that needed it)}."""
        if not store.complete():
            raise ValueError("WeightStore is incomplete")
        self.device = device
        self.num_classes = store.num_classes
        self._keep: List[torch.Tensor] = []
        dev = lambda a: self._hold(torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device))
        if equalise:
            store, tap_scale, span = store.equalised()
            self.status = {"equalised": True, "channel_gain_span_log2": [float(v) for v in span],
                           "max_span_log2": float(span.max())}
        else:
            tap_scale = np.ones(ops.FEAT_DIM, np.float32)
            self.status = {"equalised": False, "channel_gain_span_log2": None, "max_span_log2": None}
        self.equalised = bool(equalise)
        self.tap_scale = dev(tap_scale)                      # [1472]: internal = true * tap_scale

        # ---- VGG-16 --------------------------------------------------------------------
        v = VggWeights()
        for i, nm in enumerate(VGG_CONV_NAMES):
            w = store[nm + "/weights"]                     # [3,3,Cin,Cout] HWIO
            kh, kw, ci, co = w.shape
            wd = dev(w.reshape(kh * kw * ci, co))
            packed = self._hold(ops.pack_kn(wd))
            v.conv_w[i] = packed.data_ptr()
            if ci != 3:   # three-term bf16 image: same fp32 accuracy on the 16x faster bf16 MFMA pipes
                v.conv_w_x3[i] = self._hold(ops.pack_kn_x3(wd)).data_ptr()
            if conv_h2:   # single-image kernels (conv_h2.hip): two-term f16 image; conv1_1: the TF tensor as is
                img_h2 = wd if ci == 3 else self._hold(ops.pack_conv_h2(wd))
                v.conv_w_h2[i] = img_h2.data_ptr()
                if ci != 3:   # the accuracy contract's guard (disn_conv_h2_gain_span): the span of what was ACTUALLY packed
                    sp, warned = ops.conv_h2_gain_span(img_h2, ci, co)
                    self.status.setdefault("packed_gain_span_log2", []).append(sp)
                    if warned:
                        self.status.setdefault("gain_span_warnings", []).append(nm)
            v.conv_b[i] = dev(store[nm + "/biases"]).data_ptr()
        for i, nm in enumerate(("fc6", "fc7", "fc8")):
            w = store["vgg_16/%s/weights" % nm]
            wkn = w.reshape(-1, w.shape[3])
            v.fc_w[i] = dev(wkn).data_ptr()   # [K][N], K=(h,w,c) = NHWC flatten
            # [N][K]: one launch per layer, no split-K partials (gemv_rows_kernel) -- fc7 / fc8 only: fc6 stays on the
            # split-K stream kernel (api.hip vgg_head passes no transposed matrix for it), so its 411 MB transpose is
            # neither built nor uploaded
            if i > 0:
                v.fc_w_t[i] = dev(wkn.T).data_ptr()
            v.fc_b[i] = dev(store["vgg_16/%s/biases" % nm]).data_ptr()
        v.num_classes = store.num_classes
        self.vgg = v

        # ---- point MLPs ------------------------------------------------------------------
        m = MlpWeights()
        g, l = MLP_SCOPES

        def W(scope, layer):
            return store["%s/%s/weights" % (scope, layer)][0, 0]   # [Cin,Cout]

        def Bv(scope, layer):
            return store["%s/%s/biases" % (scope, layer)]

        def pk(a, x3_field=None):
            d = dev(a)
            if x3_field:
                setattr(m, x3_field, self._hold(ops.pack_kn_x3(d)).data_ptr())
            return self._hold(ops.pack_kn(d)).data_ptr()

        for pre, scope in (("g", g), ("l", l)):
            setattr(m, pre + "_w1", dev(W(scope, "fold1/conv1")).data_ptr())       # [3,64] as is
            setattr(m, pre + "_b1", dev(Bv(scope, "fold1/conv1")).data_ptr())
            setattr(m, pre + "_w2", pk(W(scope, "fold1/conv2"), pre + "_x2"))
            setattr(m, pre + "_b2", dev(Bv(scope, "fold1/conv2")).data_ptr())
            setattr(m, pre + "_w3", pk(W(scope, "fold1/conv3"), pre + "_x3"))
            setattr(m, pre + "_b3", dev(Bv(scope, "fold1/conv3")).data_ptr())
            setattr(m, pre + "_b4", dev(Bv(scope, "fold2/conv1")).data_ptr())
            setattr(m, pre + "_w5", pk(W(scope, "fold2/conv2"), pre + "_x5"))
            setattr(m, pre + "_b5", dev(Bv(scope, "fold2/conv2")).data_ptr())
            setattr(m, pre + "_w6", dev(W(scope, "fold2/conv5").reshape(-1)).data_ptr())   # [256]
            setattr(m, pre + "_b6", dev(Bv(scope, "fold2/conv5")).data_ptr())
        w4g = W(g, "fold2/conv1")                          # [512+1024, 512]: rows 0-511 point, rest global
        m.g_w4_point = pk(w4g[:512], "g_x4_point")
        m.g_w4_global = dev(w4g[512:]).data_ptr()          # folded into a per-image bias by the library
        m.g_w4_global_t = dev(w4g[512:].T).data_ptr()      # [512][1024]: that fold as one launch
        w4l = W(l, "fold2/conv1")                          # [512+1472, 512]
        m.l_w4 = pk(w4l, "l_x4")
        m.l_w4_point = pk(w4l[:512], "l_x4_point")         # the two halves on their own: the folded
        m.l_w4_feat = pk(w4l[512:], "l_x4_feat")           # local stream (disn_fold_local)
        # fused point-MLP kernels (mlp_fused.hip): one weight image per stream from the raw TF matrices
        for pre, scope, w4 in (("g", g, w4g), ("l", l, w4l)):
            img = ops.mlp_fused_pack(dev(W(scope, "fold1/conv2")), dev(W(scope, "fold1/conv3")), dev(w4[:512]),
                                     dev(W(scope, "fold2/conv2")))
            setattr(m, pre + "_fused", self._hold(img).data_ptr())
        # ... and the FEAT form of the local stream (small point sets: the gathered features as reduction blocks)
        m.l_feat = self._hold(ops.mlp_fused_feat_pack(dev(W(l, "fold1/conv2")), dev(W(l, "fold1/conv3")), dev(w4l),
                                                      dev(W(l, "fold2/conv2")))).data_ptr()
        # dense_h2.hip images (two-term f16): the layers of a small point set (the 2048-point step), one launch each
        def d2(a):
            return self._hold(ops.pack_dense_h2(dev(a))).data_ptr()

        m.g_d2, m.g_d3, m.g_d4_point, m.g_d5 = (d2(W(g, "fold1/conv2")), d2(W(g, "fold1/conv3")), d2(w4g[:512]),
                                                d2(W(g, "fold2/conv2")))
        w4l_pad = np.zeros((2048, 512), np.float32)        # [1984][512] + 64 zero rows: 256-column chunks
        w4l_pad[:w4l.shape[0]] = w4l
        m.l_d2, m.l_d3, m.l_d4, m.l_d5 = (d2(W(l, "fold1/conv2")), d2(W(l, "fold1/conv3")), d2(w4l_pad),
                                          d2(W(l, "fold2/conv2")))
        for f in MLP_FIELDS:
            assert getattr(m, f), f
        self.mlp = m
        torch.cuda.current_stream(device).synchronize()

    def _hold(self, t: torch.Tensor) -> torch.Tensor:
        self._keep.append(t)
        return t


# Points per image from which folding the local fold2/conv1 into the feature map pays: the fold is a
# 28 GFLOP GEMM (~0.25 ms), it saves 1.5 MFLOP and 15 KB of gather traffic per point (~12 ns).
FOLD_MIN_POINTS = 32768


@dataclass
class Encoded:
    """Per-image state produced once by the encoder."""
    resized: torch.Tensor          # [B,224,224,3]   'resized_ref_img'
    taps: List[torch.Tensor]       # conv1_2, conv2_2, conv3_3, conv4_3, conv5_3 (native resolution)
    embedding: torch.Tensor        # [B,1024]        'img_embedding'
    featmap: Optional[torch.Tensor]  # [B,137,137,1472] the five resized taps, channel-concatenated;
                                     # None until something needs it (SdfEngine.featmap_of builds it)
    pred: Optional[torch.Tensor] = None   # pred_sdf of the run that produced this state (encode_query)
    pmap: Optional[dict] = None           # image index -> [137*137,512] folded feature map (SdfEngine.pmap_of)
    pmap_amax: Optional[dict] = None      # image index -> max |pmap| (1-element tensor; the fused kernels' bound)
    units: str = "true"                   # "equalised": taps / featmap carry the engine's power-of-two channel factors
                                          # (DeviceWeights.tap_scale; SdfEngine.true_taps / true_features convert);
                                          # "true": the reference's end_points magnitudes


class SdfEngine:
    """``fused``: run the folded queries (large point sets, the dense grid) through the fused point-MLP
    kernels (mlp_fused.hip: activations in registers, fp32-accurate two-term fp16 products) instead of the
    layer-by-layer GEMM chain (three-term bf16 products); same math, fp32-rounding-level difference."""

    def __init__(self, store: Optional[WeightStore], device: Optional[torch.device] = None, fused: bool = True,
                 conv_h2: bool = True, weights: Optional[DeviceWeights] = None, equalise: bool = True,
                 strict: bool = False):
        """``weights``: share the device weights of another engine (``store`` is then ignored): several engine
        contexts -- each with its own workspaces and auxiliary stream -- over one copy of the ~0.85 GB of device weights
        (fc6..fc8 as [K][N] 495 MB, fc7 / fc8 transposed 84 MB, the convolutions in three packed forms 206 MB, the
        point MLPs and their images ~50 MB)"""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if not torch.cuda.is_available():
            raise RuntimeError("disn_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(device)
        self.fused = bool(fused)
        self.strict = bool(strict)
        with torch.cuda.device(self.device):
            self.weights = weights if weights is not None else DeviceWeights(store, self.device, conv_h2=conv_h2,
                                                                             equalise=equalise)
            self._ctx = ops.ctx_create()      # aux HIP stream + events for the overlapped encoder
        # ``strict``: disn_vgg_weights_t.strict_forms = 1 -- the single-image convolution kernels (k-wave tree, chains of 108
        # MFMAs per accumulator) for calls of ANY size instead of the batched form (chains of up to 432) from four images
        # on, the fc head and the point-MLP layers likewise: a request's taps, embedding and pred_sdf in a batched call are
        # bit for bit those of the request alone; about 40 % of a batched call's throughput
        # (include/disn_amd.h).  The struct is this engine's own copy: the packed weights stay shared.
        self._vgg = VggWeights.from_buffer_copy(self.weights.vgg)
        self._vgg.strict_forms = 1 if self.strict else 0
        self._ws: Dict[str, torch.Tensor] = {}

    def __del__(self):
        try:
            ctx, self._ctx = getattr(self, "_ctx", None), None
            if ctx:
                with torch.cuda.device(self.device):
                    ops.ctx_destroy(ctx)
        except Exception:  # interpreter shutdown
            pass

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = t
        return t

    def _dev(self, a) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            # (the common case first: a resident fp32 tensor costs one comparison chain, not two dispatcher calls -- a
            # 16-step call assembles 48 of them on the host before its first launch)
            if a.dtype is torch.float32 and a.device == self.device and a.is_contiguous():
                return a
            return a.to(self.device, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    # equalised <-> true units (DeviceWeights(equalise=True): Encoded.taps / .featmap and everything gathered from them
    # carry the power-of-two channel factors weights.tap_scale; identity copies for an engine built with equalise=False)
    def _units(self) -> str:
        return "equalised" if self.weights.status.get("equalised") else "true"

    def true_taps(self, enc: "Encoded") -> List[torch.Tensor]:
        """the five taps in the reference's units (end_points of slim vgg_16: models/model_normalization.py:76-78)"""
        with torch.cuda.device(self.device):
            o, out = 0, []
            for t in enc.taps:
                c = t.shape[-1]
                out.append(ops.scale_channels(t, self.weights.tap_scale[o:o + c], invert=True))
                o += c
        return out

    def true_features(self, feat: torch.Tensor) -> torch.Tensor:
        """[..., 1472] gathered features / feature map: internal -> the reference's units ('point_img_feat')"""
        with torch.cuda.device(self.device):
            return ops.scale_channels(feat, self.weights.tap_scale, invert=True)

    def internal_features(self, feat: torch.Tensor) -> torch.Tensor:
        """[..., 1472] features in the reference's units (fed 'point_img_feat') -> what this engine's MLP weights expect"""
        with torch.cuda.device(self.device):
            return ops.scale_channels(self._dev(feat), self.weights.tap_scale)

    # rows A, B, C, E
    def encode(self, imgs) -> Encoded:
        imgs = self._dev(imgs)
        with torch.cuda.device(self.device):
            from ._lib import lib
            ws = self._workspace("vgg", lib().disn_encode_workspace_bytes(imgs.shape[0]))
            resized, taps, emb, featmap = ops.encode(self._ctx, self._vgg, imgs, ws)
        return Encoded(resized, taps, emb, featmap, units=self._units())

    # rows A..H in one call: what ONE sess.run([pred_sdf]) of the reference executes
    def featmap_of(self, enc: Encoded) -> torch.Tensor:
        """enc.featmap, built from the taps on first use (row E; encode_query does not write it)."""
        if enc.featmap is None:
            with torch.cuda.device(self.device):
                enc.featmap = ops.build_featmap(enc.taps)
        return enc.featmap

    def pmap_of(self, enc: Encoded, image_index: int) -> torch.Tensor:
        """Feature map of one image times the feature rows of the local fold2/conv1 (disn_fold_local),
        built on first use: what the folded queries gather from (4 x 512 instead of 4 x 1472 floats per
        point, and a 512- instead of a 1984-deep layer)."""
        if enc.pmap is None:
            enc.pmap = {}
        if image_index not in enc.pmap:
            with torch.cuda.device(self.device):
                enc.pmap[image_index] = ops.fold_local(self.weights.mlp, self.featmap_of(enc)[image_index])
        return enc.pmap[image_index]

    def pmap_amax_of(self, enc: Encoded, image_index: int) -> torch.Tensor:
        """max |pmap| of one image (1-element device tensor): bounds the additive term of fold2/conv1 when the
        fused kernels pick the activation scale of that layer"""
        if enc.pmap_amax is None:
            enc.pmap_amax = {}
        if image_index not in enc.pmap_amax:
            with torch.cuda.device(self.device):
                enc.pmap_amax[image_index] = ops.amax(self.pmap_of(enc, image_index))
        return enc.pmap_amax[image_index]

    def encode_query(self, imgs, pts, trans_mat, pts_rot=None, keep_featmap: bool = False):
        """-> (Encoded, pred_sdf [B,N]).  Nothing cached; B*N <= 65536.  The fc weight stream (HBM
        bound) overlaps the gather + local MLP on a second stream.  Without ``keep_featmap`` the
        110 MB/image feature map is not materialised: the gather up-samples the taps at the <= 4
        pixels a point touches (bit-identical result); a later query()/query_grid() on the returned
        state builds the map once from the taps."""
        imgs, pts, trans_mat = self._dev(imgs), self._dev(pts), self._dev(trans_mat)
        pts_rot = pts if pts_rot is None else self._dev(pts_rot)
        with torch.cuda.device(self.device):
            from ._lib import lib
            ws = self._workspace("encq", lib().disn_encode_query_workspace_bytes(pts.shape[0], pts.shape[1]))
            resized, taps, emb, featmap, sdf = ops.encode_query(self._ctx, self._vgg, self.weights.mlp,
                                                                imgs, trans_mat, pts, pts_rot, ws, keep_featmap)
        return Encoded(resized, taps, emb, featmap, units=self._units()), sdf

    # rows D, F, G, H
    def query(self, enc: Encoded, pts, trans_mat, pts_rot=None, fold: Optional[bool] = None,
              fused: Optional[bool] = None) -> torch.Tensor:
        """pts [B,N,3] -> pred_sdf [B,N] (un-divided, as models/model_normalization.py:204).
        ``fold``: use the folded local stream (pmap_of; same math re-associated, fp32-rounding-level
        difference); default: from FOLD_MIN_POINTS points per image on.  ``fused``: run the folded form
        through the fused kernels (default: the engine's setting)."""
        pts = self._dev(pts)
        pts_rot = pts if pts_rot is None else self._dev(pts_rot)
        trans_mat = self._dev(trans_mat)
        with torch.cuda.device(self.device):
            from ._lib import lib
            ws = self._workspace("query", lib().disn_query_workspace_bytes(pts.shape[0], pts.shape[1]))
            if fold is None:
                fold = pts.shape[1] >= FOLD_MIN_POINTS
            if fold:
                B = pts.shape[0]
                pm = self.pmap_of(enc, 0) if B == 1 else torch.stack([self.pmap_of(enc, b) for b in range(B)])
                if self.fused if fused is None else fused:
                    am = torch.cat([self.pmap_amax_of(enc, b) for b in range(B)])
                    wsf = self._workspace("fused", lib().disn_query_fused_workspace_bytes(B, pts.shape[1]))
                    return ops.query_fused(self.weights.mlp, pm, am, enc.embedding, trans_mat, pts, pts_rot, wsf)
                return ops.query_folded(self.weights.mlp, pm, enc.embedding, trans_mat, pts, pts_rot, ws)
            return ops.query(self.weights.mlp, self.featmap_of(enc), enc.embedding, trans_mat, pts, pts_rot, ws)

    def query_grid(self, enc: Encoded, image_index: int, trans_mat, sdf_params, res: int,
                   k0: int = 0, k1: Optional[int] = None, sdf_weight: float = 10.0,
                   out: Optional[torch.Tensor] = None, pipelined: bool = False,
                   fold: Optional[bool] = None, fused: Optional[bool] = None) -> torch.Tensor:
        """rows J + D..H + '/SDF_WEIGHT' for grid points k0..k1-1 of one image.  ``fold``: folded local
        stream (see query(); the default unless ``pipelined`` -- for every range size, so that any slice of
        a grid equals the same slice of the whole grid up to the GEMM plan).  ``pipelined``:
        chunk i+1's gather on the auxiliary stream under chunk i's MLP (same result; measured
        0.564 s vs 0.561 s sequential for 257^3 -- the gather's traffic slows the GEMMs as much as
        it hides, so it is off by default)."""
        total = (res + 1) ** 3
        k1 = total if k1 is None else k1
        tm = self._dev(trans_mat).reshape(-1, 4, 3)
        tm = tm[image_index if tm.shape[0] > 1 else 0]
        with torch.cuda.device(self.device):
            from ._lib import lib
            ctx = self._ctx if pipelined else None
            if fold is None:
                fold = not pipelined
            if fold and (self.fused if fused is None else fused):
                # one launch per MLP stream over the whole range (no chunks, no per-point activations in HBM)
                ws = self._workspace("fused", lib().disn_query_grid_fused_workspace_bytes(k1 - k0))
                return ops.query_grid_fused(self.weights.mlp, self.pmap_of(enc, image_index),
                                            self.pmap_amax_of(enc, image_index),
                                            enc.embedding[image_index:image_index + 1], tm.contiguous(), sdf_params,
                                            res, k0, k1, sdf_weight, ws, out)
            if fold:
                ws = self._workspace("grid", lib().disn_query_grid_workspace_bytes(k1 - k0))
                return ops.query_grid(self.weights.mlp, None, enc.embedding[image_index:image_index + 1],
                                      tm.contiguous(), sdf_params, res, k0, k1, sdf_weight, ws, out, None,
                                      self.pmap_of(enc, image_index))
            need = (lib().disn_query_grid_ctx_workspace_bytes(k1 - k0) if ctx
                    else lib().disn_query_grid_workspace_bytes(k1 - k0))
            ws = self._workspace("grid", need)
            return ops.query_grid(self.weights.mlp, self.featmap_of(enc)[image_index], enc.embedding[image_index:image_index + 1],
                                  tm.contiguous(), sdf_params, res, k0, k1, sdf_weight, ws, out, ctx)


class StepPipeline:
    """``in_flight`` independent encode + query steps at a time on one GPU.

    A step (disn_encode_query) is a chain of ~35 dependent launches; between two dependent launches the GPU idles
    for several microseconds (measured on MI355X: a convolution layer takes 18 us alone and 26 us as a link of the
    chain), which at a 0.5 ms step is a third of the time.  Steps on different images are independent, so the
    chains of several steps -- each on its own HIP stream, fed by its own host thread (the ctypes calls release
    the GIL), each with its own workspaces and auxiliary stream, all over ONE copy of the weights -- fill each
    other's gaps: 0.49 -> 0.35 ms per step at three in flight (tools/multi_stream_try.py).  Every step still does
    all of its work; nothing is shared or cached between steps, and each result equals the single-stream one bit
    for bit (the kernels and their launch order within a step are the same)."""

    def __init__(self, store: WeightStore, device: Optional[torch.device] = None, in_flight: int = 3,
                 batch: int = 1, strict: bool = False):
        """``batch``: consecutive jobs of equal shape are submitted ``batch`` at a time as ONE disn_encode_query call
        (images and point sets concatenated): the 495 MB of fc weights are read once per call instead of once per
        image, and every launch has ``batch`` times the work against the same fixed cost.  Every image keeps its own
        activation scales (one slot set per image, picked per row tile inside the one launch of a layer), so its result
        is bit for bit what it gets alone."""
        import queue
        import threading
        self.batch = max(1, int(batch))
        self.trace = None      # a list: run() appends (context, call index, time.perf_counter() after enqueueing)
        self._threading = threading
        # Stream / context creation ORDER matters: ROCm hands out its hardware queues (GPU_MAX_HW_QUEUES, default
        # 4 including the null stream's) to streams in creation order and shares them afterwards, and two streams
        # on one queue serialise.  So: for every step context its launch stream, then its disn_ctx_t (auxiliary
        # stream), created back to back through the library -- not torch.cuda.Stream(), whose first use creates a
        # pool of 64 streams and makes the assignment a lottery (0.35 ... 0.75 ms per step measured for one and
        # the same program).  Measured with this order (tools/pipeline_try.py, one MI355X): default 4 queues
        # 0.505 / 0.41 / 0.35 / 0.38 ms per step at 1 / 2 / 3 / 4 in flight; GPU_MAX_HW_QUEUES=8 is WORSE
        # (0.50 / 0.45 / 0.45 at 2 / 3 / 4) -- leave the runtime's default.
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._handles, self.streams, self.engines = [], [], []
        weights = None
        with torch.cuda.device(self.device):
            for _ in range(in_flight):
                h = ops.stream_create()
                self._handles.append(h)
                self.streams.append(torch.cuda.ExternalStream(h, device=self.device))
                eng = SdfEngine(store if weights is None else None, self.device, weights=weights, strict=strict)
                weights = eng.weights
                self.engines.append(eng)
        # one host thread per context, alive for the pipeline's life (a thread started per run() costs ~0.5 ms
        # before its first launch: visible when a run is only a few calls long); it keeps its stream current
        self._queues = [queue.SimpleQueue() for _ in range(in_flight)]
        self._workers = [threading.Thread(target=StepPipeline._worker, daemon=True,
                                          args=(i, self._queues[i], self.device, self.streams[i]))
                         for i in range(in_flight)]      # (no reference to self: the pipeline stays collectable)
        for t in self._workers:
            t.start()

    @staticmethod
    def _worker(i, q, device, stream):
        with torch.cuda.device(device), torch.cuda.stream(stream):
            while True:
                item = q.get()
                if item is None:
                    return
                fn, errs, done = item
                try:
                    fn(i)
                except BaseException as e:  # noqa: BLE001 -- handed to the caller of run()
                    errs.append(e)
                finally:
                    del fn, item
                    done.release()

    def _dispatch(self, fn, n):
        """fn(i) on the host thread of context i for i < n; returns when all have returned, re-raises their errors"""
        errs, done = [], self._threading.Semaphore(0)
        for i in range(n):
            self._queues[i].put((fn, errs, done))
        for _ in range(n):
            done.acquire()
        if errs:
            raise errs[0]

    def close(self):
        """stop the host threads, free the contexts and streams (also done when the pipeline is collected)"""
        self.__del__()

    def __del__(self):
        try:
            with torch.cuda.device(self.device):
                for q in getattr(self, "_queues", []):
                    q.put(None)
                for t in getattr(self, "_workers", []):
                    t.join(timeout=5)
                self._queues, self._workers = [], []
                torch.cuda.synchronize(self.device)
                self.engines = []
                for h in self._handles:
                    ops.stream_destroy(h)
                self._handles = []
        except Exception:  # interpreter shutdown
            pass

    def run(self, jobs, keep_encoded: bool = False, chained: Optional[bool] = None, balance: bool = False):
        """jobs: sequence of (imgs, pts, trans_mat[, pts_rot]) -> list of pred_sdf (or (Encoded, pred_sdf)) in
        job order.  Job k runs on engine context k % in_flight.  Returns with the work enqueued, not finished:
        synchronise the device (or the returned tensors' use on the current stream) as usual.

        ``chained`` (default False): ONE host thread enqueues the steps in order and the convolution stack of step
        k+1 starts behind the one of step k (disn_ctx_pipeline): a deterministic two-stage pipeline -- convolutions
        of one step beside the fc head and point MLPs of the previous one.  Measured 0.457 ms per step, every run
        the same: the single host thread (~40 launches and event calls per step through Python) is the pacer, not
        the GPU.  Default: every context is fed by its own host thread and the steps overlap as the hardware
        schedules them: 0.33 ms per step on average at three in flight, 0.33 ... 0.40 ms from run to run (when two
        steps' convolution stacks happen to run in phase they halve each other)."""
        S = len(self.engines)
        if chained is None:
            chained = False
        if self.batch > 1:
            return self._run_batched(jobs, keep_encoded, balance, chained)
        out = [None] * len(jobs)
        cur = torch.cuda.current_stream(self.device)
        if chained:
            with torch.cuda.device(self.device):
                if not hasattr(self, "_conv_done"):
                    self._conv_done = [torch.cuda.Event() for _ in range(S + 1)]
                    for e in self._conv_done:
                        e.record()                       # creates the hipEvent_t handles
                for st in self.streams:
                    st.wait_stream(cur)
                prev = None
                try:
                    for k, job in enumerate(jobs):
                        i = k % S
                        rec = self._conv_done[k % (S + 1)]
                        ops.ctx_pipeline(self.engines[i]._ctx, prev, rec)
                        with torch.cuda.stream(self.streams[i]):
                            enc, sdf = self.engines[i].encode_query(*job)
                        out[k] = (enc, sdf) if keep_encoded else sdf
                        prev = rec
                finally:                                 # a call that raised must not leave stale event handles behind
                    for eng in self.engines:
                        ops.ctx_pipeline(eng._ctx, None, None)
            for st in self.streams:
                cur.wait_stream(st)
            return out

        def work(i):                                      # on context i's host thread: its stream is current
            self.streams[i].wait_stream(cur)              # inputs produced on the caller's stream
            for k in range(i, len(jobs), S):
                enc, sdf = self.engines[i].encode_query(*jobs[k])
                out[k] = (enc, sdf) if keep_encoded else sdf

        self._dispatch(work, min(S, len(jobs)))
        for st in self.streams:
            cur.wait_stream(st)                            # results are ordered before later work on the caller's stream
        return out

    def run_calls(self, calls):
        """calls: sequence of ALREADY batched requests (imgs [B,137,137,3], pts [B,N,3], trans_mat [B,4,3][, pts_rot]) --
        one disn_encode_query call each, call g on context g % in_flight (own stream, own host thread) -> list of
        pred_sdf [B,N].  What run() does after grouping and concatenating single requests, without the per-request host
        work: for a client that keeps its request batches in device buffers."""
        S = len(self.engines)
        out = [None] * len(calls)
        cur = torch.cuda.current_stream(self.device)

        def work(i):
            if self.trace is not None:
                self.trace.append((i, "start", time.perf_counter()))
            self.streams[i].wait_stream(cur)
            for g in range(i, len(calls), S):
                out[g] = self.engines[i].encode_query(*calls[g])[1]
                if self.trace is not None:
                    self.trace.append((i, g, time.perf_counter()))

        self._dispatch(work, min(S, len(calls)))
        for st in self.streams:
            cur.wait_stream(st)
        return out

    @staticmethod
    def call_sizes(njobs: int, batch: int, in_flight: int, balance: bool):
        """how many consecutive jobs go into each call.  Default: full calls of ``batch`` + a shorter last one.
        ``balance``: the same number of calls rounded up to a multiple of ``in_flight`` (every context gets the same
        number), all of (nearly) equal size -- never below four jobs per call while full calls would have had four (the
        batched kernels' threshold), never above ``batch``: a short run (bench.py --steps 20) then ends with every
        context busy instead of one draining a short last call alone."""
        if njobs <= 0:
            return []
        n = -(-njobs // batch)
        if not balance:
            return [batch] * (njobs // batch) + ([njobs % batch] if njobs % batch else [])
        n = -(-n // in_flight) * in_flight
        while n > 1 and njobs // n < min(4, batch):
            n -= 1
        n = max(n, -(-njobs // batch))
        base, extra = divmod(njobs, n)
        return [base + (1 if i < extra else 0) for i in range(n)]

    def _run_batched(self, jobs, keep_encoded, balance=False, chained=False):
        """groups of up to ``batch`` consecutive jobs -> one call each; group g runs on context g % in_flight.
        ``chained``: ONE host thread enqueues the calls in order and call g + 1's convolution stack starts behind call g's
        (disn_ctx_pipeline): a two-stage pipeline -- the MFMA-bound convolutions of one call beside the tail (gather,
        fused point-MLP kernels, HBM-bound fc head) of the previous one -- instead of calls that start together and stay
        in phase (convolutions beside convolutions, tails beside tails)."""
        S, Bn = len(self.engines), self.batch
        groups, o = [], 0
        for n in self.call_sizes(len(jobs), Bn, S, balance):
            groups.append(list(range(o, o + n)))
            o += n
        out = [None] * len(jobs)
        cur = torch.cuda.current_stream(self.device)

        def cat(idx, pos):
            ts = [self.engines[0]._dev(jobs[k][pos]) for k in idx]
            if len(ts) == 1:
                return ts[0]
            # requests that already lie back to back in one allocation (a client that keeps its request batch in one
            # buffer; bench.py's job pool) are submitted as they are: no copy, no allocation, ~0.1 ms less host time per call
            t0 = ts[0]
            if t0.is_contiguous():
                step = t0.numel() * t0.element_size()
                if all(t.is_contiguous() and t.dtype == t0.dtype and t.shape == t0.shape and
                       t.data_ptr() == t0.data_ptr() + i * step and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr()
                       for i, t in enumerate(ts)):
                    off = (t0.data_ptr() - t0.untyped_storage().data_ptr()) // t0.element_size()
                    flat = torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage(), off,
                                                                                   (len(ts) * t0.shape[0],) + tuple(t0.shape[1:]))
                    return flat
            return torch.cat(ts, dim=0)

        # call g runs on context g % in_flight.  (Tried for short runs -- bench.py --steps 20 = 8 + 8 + 4: the full calls back
        # to back on one context and the remainder beside them, because two calls started together stay in phase and gain
        # little from each other; no better, 4.13 vs 4.05-4.13 ms: profiles/r03o_bench_pack20.txt.)
        mine = [[g for g in range(len(groups)) if g % S == i] for i in range(S)]

        def check_one_shape(idx):
            if len({tuple(jobs[k][1].shape) for k in idx}) != 1:
                raise ValueError("jobs of one batch must have point sets of one shape")

        def work(i):
            if self.trace is not None:           # host-side pacing: this context's host thread is running
                self.trace.append((i, "start", time.perf_counter()))
            self.streams[i].wait_stream(cur)
            for g in mine[i]:
                idx = groups[g]
                check_one_shape(idx)
                args = [cat(idx, p) for p in range(len(jobs[idx[0]]))]
                if self.trace is not None:       # ... the call's inputs are assembled
                    self.trace.append((i, "call %d" % g, time.perf_counter()))
                enc, sdf = self.engines[i].encode_query(*args)
                if self.trace is not None:       # ... its launches are all enqueued
                    self.trace.append((i, g, time.perf_counter()))
                o = 0
                for k in idx:
                    b = jobs[k][0].shape[0]
                    if keep_encoded:
                        e = Encoded(enc.resized[o:o + b], [t[o:o + b] for t in enc.taps], enc.embedding[o:o + b], None, units=enc.units)
                        out[k] = (e, sdf[o:o + b])
                    else:
                        out[k] = sdf[o:o + b]
                    o += b

        if chained and len(groups) > 1:
            with torch.cuda.device(self.device):
                if not hasattr(self, "_conv_done"):
                    self._conv_done = [torch.cuda.Event() for _ in range(S + 1)]
                    for e in self._conv_done:
                        e.record()                       # creates the hipEvent_t handles
                for st in self.streams:
                    st.wait_stream(cur)
                prev = None
                try:
                    for g, idx in enumerate(groups):
                        i = g % S
                        check_one_shape(idx)
                        rec = self._conv_done[g % (S + 1)]
                        ops.ctx_pipeline(self.engines[i]._ctx, prev, rec)
                        # the call's inputs are assembled WITH stream i current: a real copy (torch.cat of separately
                        # allocated requests, the H2D copy of a numpy feed) is then ordered before the call that reads
                        # it and its temporaries belong to stream i's allocator pool
                        with torch.cuda.stream(self.streams[i]):
                            args = [cat(idx, p) for p in range(len(jobs[idx[0]]))]
                            enc, sdf = self.engines[i].encode_query(*args)
                        o = 0
                        for k in idx:
                            b = jobs[k][0].shape[0]
                            out[k] = (Encoded(enc.resized[o:o + b], [t[o:o + b] for t in enc.taps], enc.embedding[o:o + b], None, units=enc.units),
                                      sdf[o:o + b]) if keep_encoded else sdf[o:o + b]
                            o += b
                        prev = rec
                finally:                                 # a call that raised must not leave stale event handles behind
                    for eng in self.engines:
                        ops.ctx_pipeline(eng._ctx, None, None)
            for st in self.streams:
                cur.wait_stream(st)
            return out
        self._dispatch(work, min(S, len(groups)))
        for st in self.streams:
            cur.wait_stream(st)
        return out
