"""ctypes binding of libdisn_amd.so (include/disn_amd.h).

The HIP library is the product: there is NO CPU fallback.  ``lib()`` raises
``DisnLibraryError`` when the shared object is missing or does not export the
ABI this package was written against; every wrapper raises ``DisnError`` on a
non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# DISN_AMD_LIB: tools/ only -- points the binding at a tuning build (csrc/build.py --tuning), never set by the product
LIB_PATH = os.environ.get("DISN_AMD_LIB") or os.path.join(HERE, "csrc", "libdisn_amd.so")
ABI_VERSION = 10

c_float_p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class DisnLibraryError(RuntimeError):
    pass


class DisnError(RuntimeError):
    def __init__(self, fn: str, status: int):
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "workspace too small"}.get(
            status, "hipError_t %d" % status if status > 0 else "error")
        super().__init__("%s failed: %s (status %d)" % (fn, kind, status))
        self.status = status


class VggWeights(C.Structure):  # disn_vgg_weights_t
    _fields_ = [("conv_w", C.c_void_p * 13), ("conv_b", C.c_void_p * 13),
                ("fc_w", C.c_void_p * 3), ("fc_b", C.c_void_p * 3), ("num_classes", C.c_int),
                ("conv_w_x3", C.c_void_p * 13),   # optional three-term bf16 images (disn_pack_kn_x3)
                ("conv_w_h2", C.c_void_p * 13),   # optional two-term f16 images (disn_pack_conv_h2)
                ("fc_w_t", C.c_void_p * 3),       # optional transposed fc matrices [N][K]
                ("strict_forms", C.c_int)]           # 1: single-image convolution kernels for every call size ("strict")


MLP_FIELDS = ("g_w1", "g_b1", "g_w2", "g_b2", "g_w3", "g_b3", "g_w4_point", "g_w4_global", "g_b4",
              "g_w5", "g_b5", "g_w6", "g_b6", "l_w1", "l_b1", "l_w2", "l_b2", "l_w3", "l_b3",
              "l_w4", "l_b4", "l_w5", "l_b5", "l_w6", "l_b6")


MLP_X3_FIELDS = ("g_x2", "g_x3", "g_x4_point", "g_x5", "l_x2", "l_x3", "l_x4", "l_x5")   # optional
MLP_FOLD_FIELDS = ("l_w4_point", "l_w4_feat", "l_x4_point", "l_x4_feat")   # optional: *_folded entry points
MLP_FUSED_FIELDS = ("g_fused", "l_fused")   # optional: *_fused entry points (disn_mlp_fused_pack images)
MLP_T_FIELDS = ("g_w4_global_t",)           # optional: g_w4_global transposed [512][1024]
MLP_D_FIELDS = ("g_d2", "g_d3", "g_d4_point", "g_d5", "l_d2", "l_d3", "l_d4", "l_d5")   # optional: dense_h2 images
MLP_FEAT_FIELDS = ("l_feat",)               # optional: disn_mlp_fused_feat_pack image (fused small-set local stream)


class MlpWeights(C.Structure):  # disn_mlp_weights_t
    _fields_ = [(n, C.c_void_p) for n in MLP_FIELDS + MLP_X3_FIELDS + MLP_FOLD_FIELDS + MLP_FUSED_FIELDS + MLP_T_FIELDS + MLP_D_FIELDS + MLP_FEAT_FIELDS]


CAM_FIELDS = tuple("%s_%s%d" % (t, k, i) for t in "srt" for i in (1, 2, 3) for k in "wb")


class CamWeights(C.Structure):  # disn_cam_weights_t: s_w1, s_b1, s_w2, ... t_b3
    _fields_ = [(n, C.c_void_p) for n in CAM_FIELDS]


class EqWeights(C.Structure):  # disn_eq_weights_t (HOST pointers, modified in place)
    _fields_ = [("conv_w", C.c_void_p * 13), ("conv_b", C.c_void_p * 13), ("fc6_w", C.c_void_p),
                ("mlp_w", (C.c_void_p * 6) * 2), ("mlp_b", (C.c_void_p * 6) * 2), ("num_classes", C.c_int)]


NUM_VARS = 56


class ParamLayout(C.Structure):  # disn_param_layout_t
    _fields_ = [("offset", C.c_int64 * NUM_VARS), ("count", C.c_int64 * NUM_VARS), ("total", C.c_int64)]


# name -> (restype, argtypes); every symbol declared in include/disn_amd.h
I, Z, P, F, L = C.c_int, C.c_size_t, C.c_void_p, C.c_float, C.c_int64
SIGNATURES = {
    "disn_abi_version": (I, []),
    "disn_pack_kn": (I, [P, I, I, I, P, P]),
    "disn_pack_kn_x3_bytes": (Z, [I, I]),
    "disn_pack_kn_x3": (I, [P, I, I, P, P]),
    "disn_conv3x3_x3_workspace_bytes": (Z, [I, I, I, I, I]),
    "disn_conv3x3_x3": (I, [P, I, I, I, I, P, P, I, I, P, P, Z, P]),
    "disn_pack_conv_h2_bytes": (Z, [I, I]),
    "disn_pack_conv_h2": (I, [P, I, I, P, P]),
    "disn_conv_h2_gain_span": (I, [P, I, I, P, P]),
    "disn_conv1_1_workspace_bytes": (Z, []),
    "disn_conv1_1": (I, [P, I, I, I, P, P, I, P, P, P, Z, P]),
    "disn_conv3x3_h2_workspace_bytes": (Z, [I]),
    "disn_conv3x3_h2": (I, [P, I, I, I, I, P, P, I, I, P, P, P, I, P, Z, P]),
    "disn_resize_bilinear": (I, [P, I, I, I, I, P, I, I, I, I, P]),
    "disn_vgg16_workspace_bytes": (Z, [I]),
    "disn_vgg16_forward": (I, [C.POINTER(VggWeights), P, I, P, C.POINTER(C.c_void_p * 5), P, P, Z, P]),
    "disn_vgg16_conv_stack": (I, [C.POINTER(VggWeights), P, I, P, C.POINTER(C.c_void_p * 5), P, P, Z, P]),
    "disn_conv3x3_workspace_bytes": (Z, [I, I, I, I, I]),
    "disn_conv3x3": (I, [P, I, I, I, I, P, P, I, I, P, P, Z, P]),
    "disn_conv3x3_planned_workspace_bytes": (Z, [I, I, I, I, I, I, I, I]),
    "disn_conv3x3_planned": (I, [P, I, I, I, I, P, P, I, I, P, P, Z, I, I, I, P]),
    "disn_maxpool2x2": (I, [P, I, I, I, I, P, P]),
    "disn_fc_workspace_bytes": (Z, [I, I, I]),
    "disn_fc": (I, [P, I, I, P, P, I, I, P, P, Z, P]),
    "disn_pack_dense_h2_bytes": (Z, [I, I]),
    "disn_pack_dense_h2": (I, [P, I, I, P, P]),
    "disn_dense_h2_workspace_bytes": (Z, [I]),
    "disn_dense_h2": (I, [P, I, I, P, I, I, P, I, I, P, I, I, P, P, I, I, P, P, P, Z, P]),
    "disn_get_loss": (I, [P, P, L, F, F, P, P]),
    "disn_fc_t": (I, [P, I, I, P, P, I, I, P, P]),
    "disn_dense_workspace_bytes": (Z, [I, I, I]),
    "disn_dense": (I, [P, I, I, P, I, I, I, P, P, I, I, P, P, Z, P]),
    "disn_build_featmap": (I, [C.POINTER(C.c_void_p * 5), I, P, P]),
    "disn_project": (I, [P, P, I, I, P, P]),
    "disn_gather": (I, [P, P, I, I, P, P]),
    "disn_gather_taps": (I, [C.POINTER(C.c_void_p * 5), P, P, I, I, P, P]),
    "disn_gather_taps_split": (I, [C.POINTER(C.c_void_p * 5), P, P, I, I, P, P, P]),
    "disn_gather_fold": (I, [P, P, P, I, P, P, P, P]),
    "disn_mlp_fused_image_bytes": (Z, []),
    "disn_mlp_fused_pack": (I, [P, P, P, P, P, P]),
    "disn_mlp_fused_feat_image_bytes": (Z, []),
    "disn_mlp_fused_feat_pack": (I, [P, P, P, P, P, P]),
    "disn_query_taps_fused_workspace_bytes": (Z, [I, I]),
    "disn_query_taps_fused": (I, [C.POINTER(MlpWeights), C.POINTER(C.c_void_p * 5), P, P, P, P, I, I, P, P, Z, P]),
    "disn_amax": (I, [P, L, P, P]),
    "disn_query_fused_workspace_bytes": (Z, [I, L]),
    "disn_query_fused": (I, [C.POINTER(MlpWeights), P, P, P, P, P, P, I, L, P, P, Z, P]),
    "disn_query_grid_fused_workspace_bytes": (Z, [L]),
    "disn_query_grid_fused": (I, [C.POINTER(MlpWeights), P, P, P, P, C.POINTER(C.c_double * 6), I, L, L, F, P,
                                  P, Z, P]),
    "disn_sdf_mlp_workspace_bytes": (Z, [I, I]),
    "disn_sdf_mlp": (I, [C.POINTER(MlpWeights), P, P, P, I, I, P, P, P, P, Z, P]),
    "disn_query_workspace_bytes": (Z, [I, I]),
    "disn_query": (I, [C.POINTER(MlpWeights), P, P, P, P, P, I, I, P, P, Z, P]),
    "disn_stream_create": (I, [C.POINTER(C.c_void_p)]),
    "disn_stream_destroy": (I, [P]),
    "disn_ctx_create": (I, [C.POINTER(C.c_void_p)]),
    "disn_ctx_destroy": (I, [P]),
    "disn_ctx_pipeline": (I, [P, P, P]),
    "disn_encode_workspace_bytes": (Z, [I]),
    "disn_encode": (I, [P, C.POINTER(VggWeights), P, I, P, C.POINTER(C.c_void_p * 5), P, P, P, Z, P]),
    "disn_encode_query_workspace_bytes": (Z, [I, I]),
    "disn_encode_query": (I, [P, C.POINTER(VggWeights), C.POINTER(MlpWeights), P, P, P, P, I, I, P,
                              C.POINTER(C.c_void_p * 5), P, P, P, P, Z, P]),
    "disn_query_grid_ctx_workspace_bytes": (Z, [L]),
    "disn_query_grid_ctx": (I, [P, C.POINTER(MlpWeights), P, P, P, C.POINTER(C.c_double * 6), I, L, L, F, P,
                                P, Z, P]),
    "disn_crc32c": (C.c_uint32, [P, Z, C.c_uint32]),
    "disn_equalise_weights": (I, [C.POINTER(EqWeights), P, P]),
    "disn_scale_channels": (I, [P, L, I, P, I, P, P]),
    "disn_dense_bf16_workspace_bytes": (Z, [I, I, I]),
    "disn_dense_bf16": (I, [P, I, I, P, I, I, I, P, P, I, I, I, P, P, Z, P]),
    "disn_conv3x3_bf16_workspace_bytes": (Z, [I, I, I, I, I]),
    "disn_conv3x3_bf16": (I, [P, I, I, I, I, P, P, I, I, I, P, P, Z, P]),
    "disn_cam_head": (I, [C.POINTER(CamWeights), P, C.POINTER(C.c_float * 9), I, P, P, P, P, P]),
    "disn_param_layout": (I, [C.POINTER(ParamLayout)]),
    "disn_train_workspace_bytes": (Z, [I, I]),
    "disn_train_step": (I, [P, P, P, P, P, P, P, P, I, I, F, F, F, I, P, P, P, P, Z, P]),
    "disn_adam_update": (I, [P, P, P, P, L, F, F, F, F, F, P]),
    "disn_dense_backward_workspace_bytes": (Z, [I, I, I]),
    "disn_dense_backward": (I, [P, I, I, P, P, P, I, I, F, I, P, P, P, P, Z, P]),
    "disn_conv3x3_backward_workspace_bytes": (Z, [I, I, I, I, I]),
    "disn_conv3x3_backward": (I, [P, I, I, I, I, P, P, P, I, F, I, P, P, P, P, Z, P]),
    "disn_maxpool2x2_backward": (I, [P, P, I, I, I, I, P, P]),
    "disn_resize_bilinear_backward_workspace_bytes": (Z, [I, I, I, I, I, I]),
    "disn_resize_bilinear_backward": (I, [P, I, I, I, I, I, I, I, I, P, I, P, Z, P]),
    "disn_gather_backward": (I, [P, P, I, I, P, P]),
    "disn_mc_workspace_bytes": (Z, [I]),
    "disn_mc_count": (I, [P, I, F, P, P, Z, P]),
    "disn_mc_emit": (I, [P, C.POINTER(C.c_double * 6), I, F, P, P, P, Z, P]),
    "disn_write_obj": (I, [C.c_char_p, P, L, P, L]),
    "disn_grid_points": (I, [C.POINTER(C.c_double * 6), I, L, L, P, P]),
    "disn_query_grid_workspace_bytes": (Z, [L]),
    "disn_query_grid": (I, [C.POINTER(MlpWeights), P, P, P, C.POINTER(C.c_double * 6), I, L, L, F, P,
                            P, Z, P]),
    "disn_fold_local_workspace_bytes": (Z, []),
    "disn_fold_local": (I, [C.POINTER(MlpWeights), P, P, P, Z, P]),
    "disn_query_folded": (I, [C.POINTER(MlpWeights), P, P, P, P, P, I, I, P, P, Z, P]),
    "disn_query_grid_folded": (I, [C.POINTER(MlpWeights), P, P, P, C.POINTER(C.c_double * 6), I, L, L, F, P,
                                   P, Z, P]),
}

_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and type the shared library.  Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise DisnLibraryError(
            "HIP extension %s is missing: build it with `python -m disn_amd.csrc.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise DisnLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise DisnLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    v = h.disn_abi_version()
    if v != ABI_VERSION:
        raise DisnLibraryError("ABI mismatch: library %d, binding %d" % (v, ABI_VERSION))
    _LIB = h
    return h


def check(fn: str, status: int) -> None:
    if status != 0:
        raise DisnError(fn, status)
