"""Drop-in surface of the reference's ``models/sdfnet.py`` for the two functions on the path:

    get_sdf_basic2                      /root/reference/models/sdfnet.py:69-92
    get_sdf_basic2_imgfeat_twostream    /root/reference/models/sdfnet.py:171-190

Same names and argument orders.  Like the reference they must be called under the variable
scope that owns their weights (``sdfprediction`` / ``sdfprediction_imgfeat``,
models/model_normalization.py:194,199); the other decoder variants of that file (binary,
onestream, 3-D deconvolution) are not selected by ``--img_feat_twostream`` regression mode
and are out of scope.

Both streams are computed by one library entry (``disn_sdf_mlp``); a call that needs only one
stream feeds zeros to the other and discards it.
"""
from __future__ import annotations

from . import graph, ops
from .graph import SymTensor

FEAT_DIM = 1472


def _scope_or(default: str) -> str:
    s = graph.current_scope()
    return s if s else default


def get_sdf_basic2(src_pc, globalfeats, is_training, batch_size, num_point, bn, bn_decay, wd=None):
    """Global stream: 3->64->256->512, concat [point512, global1024], ->512->256->1."""
    if bn:
        raise NotImplementedError("bn=False on this path")
    scope = _scope_or("sdfprediction")
    if scope != "sdfprediction":
        raise ValueError("get_sdf_basic2 weights live under scope 'sdfprediction', not %r" % scope)
    shp = src_pc.get_shape()

    def fn(sess, pc, emb):
        import torch
        eng = sess.engine
        pc, emb = eng._dev(pc), eng._dev(emb)
        feat = torch.zeros((pc.shape[0], pc.shape[1], FEAT_DIM), dtype=torch.float32, device=pc.device)
        _, g, _ = ops.sdf_mlp(eng.weights.mlp, pc, emb.reshape(emb.shape[0], -1), feat, want_streams=True)
        return g.reshape(g.shape[0], -1, 1)

    return SymTensor('pred_sdf_value_global', (shp[0], shp[1], 1), fn, (src_pc, globalfeats))


def get_sdf_basic2_imgfeat_twostream(src_pc, point_feat, is_training, batch_size, num_point, bn, bn_decay,
                                     wd=None):
    """Local stream: 3->64->256->512, concat [point512, feat1472], ->512->256->1."""
    if bn:
        raise NotImplementedError("bn=False on this path")
    scope = _scope_or("sdfprediction_imgfeat")
    if scope != "sdfprediction_imgfeat":
        raise ValueError("twostream weights live under scope 'sdfprediction_imgfeat', not %r" % scope)
    shp = src_pc.get_shape()

    def fn(sess, pc, feat):
        import torch
        eng = sess.engine
        pc, feat = eng._dev(pc), eng._dev(feat)
        emb = torch.zeros((pc.shape[0], 1024), dtype=torch.float32, device=pc.device)
        feat = eng.internal_features(feat.reshape(pc.shape[0], pc.shape[1], FEAT_DIM))   # end-point units -> the engine's
        _, _, l = ops.sdf_mlp(eng.weights.mlp, pc, emb, feat, want_streams=True)
        return l.reshape(l.shape[0], -1, 1)

    return SymTensor('pred_sdf_value_local', (shp[0], shp[1], 1), fn, (src_pc, point_feat))
