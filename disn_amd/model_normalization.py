"""Drop-in surface of the reference's ``models/model_normalization.py`` on the HIP engine.

Same function names, argument orders and ``end_points`` keys as the reference
(/root/reference/models/model_normalization.py):

    placeholder_inputs  :14-35      placeholder_features :38-45
    get_model           :47-221     (regression ``--img_feat_twostream`` branch :169-206)
    get_decoder         :223-238    get_img_points       :241-251
    get_loss            :254-300

so that ``import disn_amd.model_normalization as model`` plus ``disn_amd.graph.Session``
runs the loops of test/create_sdf.py:262-276 and train/train_sdf.py:371-387 unchanged.
Only the mode the hot path names is built (two-stream regression); the other ``FLAGS``
branches (binary, threedcnn, img_feat_onestream, multi_view, alpha) raise
NotImplementedError -- they are out of scope.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import numpy as np

from . import graph, ops, sdfnet
from .graph import Placeholder, SymTensor

FEAT_DIM = 1472


def _flags(FLAGS):
    d = dict(alpha=False, num_classes=1024, binary=False, threedcnn=False, img_feat_onestream=False,
             img_feat_twostream=True, multi_view=False, img_h=137, img_w=137, tanh=False,
             num_sample_points=2048, batch_size=1)
    if FLAGS is not None:
        for k in d:
            if hasattr(FLAGS, k):
                d[k] = getattr(FLAGS, k)
    return SimpleNamespace(**d)


def _check_supported(F):
    if F.alpha or F.binary or F.threedcnn or F.img_feat_onestream or F.multi_view or not F.img_feat_twostream:
        raise NotImplementedError(
            "only the regression --img_feat_twostream mode (the SDF query hot path) is built; "
            "alpha/binary/threedcnn/img_feat_onestream/multi_view are out of scope")
    if (F.img_h, F.img_w) != (137, 137):
        raise NotImplementedError("the reference hard-codes 137x137 (clamp [0,136], model_normalization.py:249-250)")
    if F.num_classes != 1024:
        raise NotImplementedError("num_classes must be 1024 (sdfprediction/fold2/conv1 expects 512+1024)")


def placeholder_inputs(batch_size, num_points, img_size, num_sample_pc=256, scope='', FLAGS=None):
    """models/model_normalization.py:14-35 -> dict of placeholders."""
    F = _flags(FLAGS)
    sdf = {}
    sdf['pc'] = Placeholder('pc', (batch_size, num_points, 3))
    sdf['sample_pc'] = Placeholder('sample_pc', (batch_size, num_sample_pc, 3))
    sdf['sample_pc_rot'] = Placeholder('sample_pc_rot', (batch_size, num_sample_pc, 3))
    sdf['imgs'] = Placeholder('imgs', (batch_size, img_size[0], img_size[1], 4 if F.alpha else 3))
    sdf['sdf'] = Placeholder('sdf', (batch_size, num_sample_pc, 1))
    sdf['sdf_params'] = Placeholder('sdf_params', (batch_size, 6))
    sdf['trans_mat'] = Placeholder('trans_mat', (batch_size, 4, 3))
    return sdf


def placeholder_features(batch_size, num_sample_pc=256, scope=''):
    """models/model_normalization.py:38-45."""
    return {'ref_feats_embedding_cnn': Placeholder('ref_feats_embedding_cnn', (batch_size, 1, 1, 1024)),
            'point_img_feat': Placeholder('point_img_feat', (batch_size, num_sample_pc, 1, FEAT_DIM))}


def get_img_points(sample_pc, trans_mat_right):
    """models/model_normalization.py:241-251: homogeneous right-multiply by the 4x3 matrix,
    divide by depth, clamp to [0,136] -> [B,N,2] (x=column, y=row)."""
    shp = sample_pc.get_shape()

    def fn(sess, pc, tm):
        return ops.project(sess.engine._dev(pc), sess.engine._dev(tm))

    return SymTensor('sample_img_points', (shp[0], shp[1], 2), fn, (sample_pc, trans_mat_right))


def get_model(ref_dict, num_point, is_training, bn=False, bn_decay=None, img_size=224, wd=1e-5, FLAGS=None):
    """models/model_normalization.py:47-221 (two-stream regression branch)."""
    F = _flags(FLAGS)
    _check_supported(F)
    if bn:
        raise NotImplementedError("bn=False on this path (train/train_sdf.py:239, test/create_sdf.py:164)")
    if img_size != 224:
        raise NotImplementedError("encoder input is 224 (callers never override img_size)")
    ref_img = ref_dict['imgs']
    ref_sample_pc = ref_dict['sample_pc']
    ref_sample_pc_rot = ref_dict['sample_pc_rot']
    ref_trans_mat = ref_dict['trans_mat']
    B = ref_img.get_shape()[0]
    N = ref_sample_pc.get_shape()[1]

    end_points = {}
    end_points['wd'] = wd                                   # get_loss's weight decay (the reference threads it through slim scopes)
    end_points['ref_pc'] = ref_dict['pc']
    end_points['ref_sdf'] = ref_dict['sdf']
    end_points['ref_img'] = ref_img                       # :62 -- the UN-resized input

    # rows A, B, C, E: resize 137->224 (:65-72), slim vgg_16 (:74-78), 5 up-sampled taps (:171-183)
    # On an encoder-cache miss with a batch that fits one launch sequence the whole graph
    # (encode + query) goes through the overlapped single entry disn_encode_query.
    enc = SymTensor('encoder', (), lambda sess, imgs, pc, pc_rot, tm: sess.encoded(imgs, pc, pc_rot, tm),
                    (ref_img, ref_sample_pc, ref_sample_pc_rot, ref_trans_mat))
    end_points['resized_ref_img'] = SymTensor('resized_ref_img', (B, 224, 224, 3),
                                              lambda sess, e: e.resized, (enc,))
    emb = SymTensor('img_embedding', (B, 1024), lambda sess, e: e.embedding, (enc,))
    end_points['img_embedding'] = emb

    # row D (:170)
    sample_img_points = get_img_points(ref_sample_pc, ref_trans_mat)

    # row F (:172-190): point_img_feat [B,N,1,1472]
    def feat_fn(sess, e, xy):   # (the engine's feature map is in equalised units: the end point is not)
        f = ops.gather(sess.engine.featmap_of(e), xy)
        return sess.engine.true_features(f).reshape(xy.shape[0], xy.shape[1], 1, FEAT_DIM)

    point_img_feat = SymTensor('point_img_feat', (B, N, 1, FEAT_DIM), feat_fn, (enc, sample_img_points))

    # rows G1, G2 (:194-202).  Evaluated on their own only if a caller fetches them.
    with graph.variable_scope("sdfprediction"):
        pred_sdf_value_global = sdfnet.get_sdf_basic2(ref_sample_pc_rot, emb, is_training, B, num_point,
                                                      bn, bn_decay, wd=wd)
    with graph.variable_scope("sdfprediction_imgfeat"):
        pred_sdf_value_local = sdfnet.get_sdf_basic2_imgfeat_twostream(ref_sample_pc_rot, point_img_feat,
                                                                       is_training, B, num_point, bn,
                                                                       bn_decay, wd=wd)

    # row H (:204): pred_sdf = global + local.  The fetch every caller uses; it runs the fused
    # project -> gather -> two MLPs -> sum entry (disn_query) instead of the three separate nodes.
    def pred_fn(sess, e, pc, pc_rot, tm):
        out = e.pred if getattr(e, "pred", None) is not None else sess.engine.query(e, pc, tm, pc_rot)
        if F.tanh:                                         # :214-215 (off by default)
            import torch
            out = torch.tanh(out)
        return out.reshape(out.shape[0], out.shape[1], 1)

    pred_sdf = SymTensor('pred_sdf', (B, N, 1), pred_fn, (enc, ref_sample_pc, ref_sample_pc_rot, ref_trans_mat))

    end_points["pred_sdf_value_global"] = pred_sdf_value_global
    end_points["pred_sdf_value_local"] = pred_sdf_value_local
    end_points['pred_sdf'] = pred_sdf
    end_points["sample_img_points"] = sample_img_points
    end_points["ref_feats_embedding_cnn"] = emb
    end_points["point_img_feat"] = point_img_feat
    return end_points


def get_decoder(num_point, input_pls, feature_pls, bn=False, bn_decay=None, wd=None):
    """models/model_normalization.py:223-238: both MLP streams from FED features."""
    emb = feature_pls["ref_feats_embedding_cnn"]
    feat = feature_pls["point_img_feat"]
    pc_rot = input_pls['sample_pc_rot']
    shp = pc_rot.get_shape()

    def fn(sess, e, f, p):
        eng = sess.engine
        e, f, p = eng._dev(e), eng._dev(f), eng._dev(p)
        f = eng.internal_features(f.reshape(p.shape[0], p.shape[1], FEAT_DIM))
        out = ops.sdf_mlp(eng.weights.mlp, p, e.reshape(e.shape[0], -1), f)
        return out.reshape(out.shape[0], out.shape[1], 1)

    return SymTensor('multi_pred_sdf', (shp[0], shp[1], 1), fn, (emb, feat, pc_rot))


def get_loss(end_points, sdf_weight=10., regularization=True, mask_weight=4.,
             num_sample_points=2048, FLAGS=None, batch_size=None):
    """models/model_normalization.py:254-300, regression branch.  The scalars come from ONE launch of the
    library's loss kernel (disn_get_loss = loss_reduce_kernel, the one disn_train_step uses); the weight decay is
    the `wd` given to get_model (slim l2_regularizer(wd) on the VGG conv weights, :75, + the tf_util
    'regularizer' collection, utils/tf_util.py:45-47: wd * sum(w^2) / 2 over every '/weights' variable).  This
    shim serves the INFERENCE / evaluation loops (test/create_sdf.py); there is no train_op here -- training is
    disn_amd.train_sdf.Trainer."""
    F = _flags(FLAGS)
    _check_supported(F)
    pred_sdf = end_points['pred_sdf']
    gt_sdf = end_points['ref_sdf']
    wd = float(end_points.get('wd', 1e-5))
    end_points['losses'] = {}

    def reg_fn(sess):
        if not regularization:
            return 0.0
        # a function of the weights alone (140 M squares in float64 on the host: ~0.5 s): computed once per weight
        # set, not at every fetch.  The cache lives ON the weight store (an id()-keyed cache on the session
        # can outlive its store -- CPython reuses ids) and is keyed by the identities of the arrays it summed, so a
        # replaced array invalidates it; an in-place edit of an array is the caller's to announce
        # (`store._regularization_cache = None`), as with any cached function of mutable data.
        store = sess.weights
        names = [k for k in store.keys() if k.endswith('/weights')]
        key = (wd, tuple(id(store[k]) for k in names))
        cache = getattr(store, '_regularization_cache', None)
        if cache is None or cache[0] != key:
            val = float(sum(wd * 0.5 * float(np.sum(np.asarray(store[k], np.float64) ** 2)) for k in names))
            cache = (key, val, [store[k] for k in names])      # (the arrays are kept alive: their ids stay theirs)
            try:
                store._regularization_cache = cache
            except AttributeError:      # a plain dict of weights: no cache
                pass
        return cache[1]

    reg = SymTensor('regularization', (), reg_fn, ())

    def five_fn(sess, pred, gt, r):
        import torch
        from . import ops
        gt_d = torch.from_numpy(np.ascontiguousarray(gt, np.float32)).to(pred.device)
        return ops.get_loss(pred.reshape(-1), gt_d.reshape(-1), sdf_weight, mask_weight, r)

    five = SymTensor('loss_scalars', (5,), five_fn, (pred_sdf, gt_sdf, reg))

    def mask_fn(sess, gt):
        gt = np.asarray(gt, np.float32)
        return (gt <= 0.01).astype(np.float32) * np.float32(mask_weight) + (gt > 0.01).astype(np.float32)

    end_points['weighed_mask'] = SymTensor('weighed_mask', gt_sdf.get_shape(), mask_fn, (gt_sdf,))
    for i, name in enumerate(('accuracy', 'sdf_loss_realvalue', 'sdf_loss')):
        end_points['losses'][name] = SymTensor(name, (), lambda sess, v, i=i: v[i], (five,))
    if regularization:
        end_points['losses']['regularization'] = reg
    loss = SymTensor('overall_loss', (), lambda sess, v: v[4] if regularization else v[2], (five,))
    end_points['losses']['overall_loss'] = loss
    return loss, end_points
