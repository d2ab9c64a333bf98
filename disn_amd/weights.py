"""Name-keyed weight store using the reference's TF variable namespace.

Checkpoint layout kept from the reference:
  vgg_16/conv{1..5}/conv{i}_{j}/{weights [3,3,Cin,Cout], biases [Cout]}
  vgg_16/fc6/weights [7,7,512,4096], fc7 [1,1,4096,4096], fc8 [1,1,4096,num_classes]
      (slim vgg_16 under scope 'vgg_16': models/model_normalization.py:76, models/CNN/vgg.py:187-214)
  sdfprediction/fold1/conv{1,2,3}, fold2/conv{1,2,5}            (models/sdfnet.py:71-88)
  sdfprediction_imgfeat/fold1/conv{1,2,3}, fold2/conv{1,2,5}    (models/sdfnet.py:173-186)
      leaf names 'weights' [1,1,Cin,Cout] / 'biases' [Cout]      (utils/tf_util.py:163,173)
Storage order is TF's HWIO; kernels re-pack at load (disn_pack_kn).  Until the
TF Saver-V2 bundle reader lands checkpoints are .npz files keyed
by exactly these names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

FEAT_DIM = 1472
VGG_CFG = (("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512))
VGG_CONV_NAMES = tuple("vgg_16/%s/%s_%d" % (s, s, j) for s, n, _ in VGG_CFG for j in range(1, n + 1))
MLP_SCOPES = ("sdfprediction", "sdfprediction_imgfeat")
MLP_LAYERS = ("fold1/conv1", "fold1/conv2", "fold1/conv3", "fold2/conv1", "fold2/conv2", "fold2/conv5")


def variable_shapes(num_classes: int = 1024) -> Dict[str, Tuple[int, ...]]:
    shapes: Dict[str, Tuple[int, ...]] = {}
    cin = 3
    for scope, n, cout in VGG_CFG:
        for j in range(1, n + 1):
            nm = "vgg_16/%s/%s_%d" % (scope, scope, j)
            shapes[nm + "/weights"] = (3, 3, cin, cout)
            shapes[nm + "/biases"] = (cout,)
            cin = cout
    for nm, shp in (("fc6", (7, 7, 512, 4096)), ("fc7", (1, 1, 4096, 4096)),
                    ("fc8", (1, 1, 4096, num_classes))):
        shapes["vgg_16/%s/weights" % nm] = shp
        shapes["vgg_16/%s/biases" % nm] = (shp[3],)
    for scope, kc in ((MLP_SCOPES[0], 512 + num_classes), (MLP_SCOPES[1], 512 + FEAT_DIM)):
        for nm, ci, co in (("fold1/conv1", 3, 64), ("fold1/conv2", 64, 256), ("fold1/conv3", 256, 512),
                           ("fold2/conv1", kc, 512), ("fold2/conv2", 512, 256), ("fold2/conv5", 256, 1)):
            shapes["%s/%s/weights" % (scope, nm)] = (1, 1, ci, co)
            shapes["%s/%s/biases" % (scope, nm)] = (co,)
    return shapes


class WeightStore:
    """dict-like {tf_variable_name: float32 ndarray} with the reference's loader semantics."""

    def __init__(self, arrays: Optional[Dict[str, np.ndarray]] = None, num_classes: int = 1024):
        self.num_classes = num_classes
        self.shapes = variable_shapes(num_classes)
        self.arrays: Dict[str, np.ndarray] = {}
        if arrays is not None:
            self.assign(arrays, strict=True)

    # -- initialisation (what tf.global_variables_initializer gives the reference) ----------
    @classmethod
    def random_init(cls, seed: int = 0, num_classes: int = 1024, mode: str = "xavier") -> "WeightStore":
        """xavier-uniform weights / zero biases = tf.contrib.layers.xavier_initializer
        (utils/tf_util.py:41) and the slim default; this is what the reference runs on when no
        checkpoint restores (test/create_sdf.py:184-192).  mode 'he' = N(0,2/fan_in) weights and
        N(0,0.1) biases for numerically meaningful parity tests."""
        rng = np.random.default_rng(seed)
        ws = cls(num_classes=num_classes)
        for name, shp in ws.shapes.items():
            if name.endswith("/weights"):
                kh, kw, ci, co = shp
                fan_in, fan_out = kh * kw * ci, kh * kw * co
                if mode == "xavier":
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    a = rng.uniform(-lim, lim, size=shp)
                elif mode == "he":
                    a = rng.normal(0.0, math.sqrt(2.0 / fan_in), size=shp)
                else:
                    raise ValueError("mode must be 'xavier' or 'he'")
            else:
                a = np.zeros(shp) if mode == "xavier" else rng.normal(0.0, 0.1, size=shp)
            ws.arrays[name] = np.ascontiguousarray(a, dtype=np.float32)
        return ws

    # -- restore: prefix + exact-shape match, skip on mismatch (train/train_sdf.py:196-205) --
    def assign(self, arrays: Dict[str, np.ndarray], strict: bool = False, prefix: str = "") -> int:
        n = 0
        for name, a in arrays.items():
            if prefix and not name.startswith(prefix):
                continue
            if name not in self.shapes:
                if strict:
                    raise KeyError("unknown variable %r" % name)
                continue
            a = np.asarray(a)
            if tuple(a.shape) != self.shapes[name]:
                if strict:
                    raise ValueError("%s: shape %s != %s" % (name, a.shape, self.shapes[name]))
                continue
            self.arrays[name] = np.ascontiguousarray(a, dtype=np.float32)
            n += 1
        return n

    def complete(self) -> bool:
        return all(k in self.arrays for k in self.shapes)

    def save(self, path: str) -> None:
        np.savez(path, **self.arrays)

    @classmethod
    def load(cls, path: str, num_classes: int = 1024, strict: bool = True) -> "WeightStore":
        with np.load(path) as z:
            ws = cls(num_classes=num_classes)
            ws.assign({k: z[k] for k in z.files}, strict=strict)
        if strict and not ws.complete():
            missing = [k for k in ws.shapes if k not in ws.arrays]
            raise KeyError("checkpoint %s lacks %d variables, e.g. %s" % (path, len(missing), missing[:3]))
        return ws

    # -- TensorFlow Saver-V2 bundles (the reference's checkpoint format) -----------------------------
    def save_tf(self, prefix: str, write_state: bool = True) -> None:
        """``saver.save(sess, prefix)`` for the model variables (train/train_sdf.py:322-328):
        <prefix>.index + <prefix>.data-00000-of-00001 (+ the 'checkpoint' state file)."""
        import os
        from . import tf_checkpoint as tfc
        tfc.save_checkpoint(prefix, self.arrays)
        if write_state:
            tfc.write_checkpoint_state(os.path.dirname(os.path.abspath(prefix)), os.path.basename(prefix))

    @classmethod
    def load_tf(cls, prefix: str, num_classes: int = 1024, strict: bool = False, name_prefix: str = "",
                verify: bool = True) -> "WeightStore":
        """``saver.restore`` semantics of the reference: variables are matched by name (optionally
        only those under ``name_prefix``, e.g. 'vgg_16' for vgg_16.ckpt -- train/train_sdf.py:276-278)
        and exact shape; optimizer slots ('.../Adam', 'beta1_power', ...) and anything else unknown
        are ignored unless ``strict``."""
        from . import tf_checkpoint as tfc
        ws = cls(num_classes=num_classes)
        entries = tfc.list_variables(prefix, verify)
        want = [n for n in entries if n in ws.shapes and (not name_prefix or n.startswith(name_prefix))]
        ws.assign(tfc.load_checkpoint(prefix, want, verify), strict=strict, prefix=name_prefix)
        if strict and not ws.complete():
            raise KeyError("checkpoint %s lacks model variables" % prefix)
        return ws

    @classmethod
    def restore_latest(cls, directory: str, num_classes: int = 1024) -> Optional["WeightStore"]:
        """``tf.train.get_checkpoint_state(dir)`` + restore (test/create_sdf.py:180-192); None when the
        directory holds no checkpoint (the caller decides about random init -- never silently)."""
        from . import tf_checkpoint as tfc
        prefix = tfc.get_checkpoint_state(directory)
        return None if prefix is None else cls.load_tf(prefix, num_classes)

    # -- equalised inference copy (include/disn_amd.h, disn_equalise_weights) -----------------------------------
    def equalised(self) -> Tuple["WeightStore", np.ndarray, np.ndarray]:
        """-> (store', tap_scale [1472], span_log2 [23]): a COPY of the variables with every hidden channel multiplied
        by a power of two and its consumers' rows divided by it (the library's host routine; exact: the network
        function and every fp32 rounding are unchanged, models/model_normalization.py:74-78,171-204).  What the
        inference engine uploads: the channels of every hidden tensor then have comparable magnitudes, which the
        per-image operand scale of the two-term f16 kernels needs on trained weights.  Taps / gathered features
        computed with store' are in equalised units: true = value / tap_scale[channel]."""
        import ctypes as C
        from ._lib import EqWeights, lib
        if not self.complete():
            raise ValueError("WeightStore is incomplete")
        out = WeightStore(num_classes=self.num_classes)
        # fc7 / fc8 (and every bias the routine does not touch) are shared, not copied: it never writes them
        touched = set(n + sfx for n in VGG_CONV_NAMES for sfx in ("/weights", "/biases"))
        touched.add("vgg_16/fc6/weights")
        touched.update("%s/%s/%s" % (s, l, leaf) for s in MLP_SCOPES for l in MLP_LAYERS for leaf in ("weights", "biases"))
        for k, a in self.arrays.items():
            out.arrays[k] = np.array(a, dtype=np.float32, order="C", copy=True) if k in touched else a
        w = EqWeights()
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        for i, nm in enumerate(VGG_CONV_NAMES):
            w.conv_w[i] = ptr(out.arrays[nm + "/weights"])
            w.conv_b[i] = ptr(out.arrays[nm + "/biases"])
        w.fc6_w = ptr(out.arrays["vgg_16/fc6/weights"])
        for s, scope in enumerate(MLP_SCOPES):
            for l, layer in enumerate(MLP_LAYERS):
                w.mlp_w[s][l] = ptr(out.arrays["%s/%s/weights" % (scope, layer)])
                w.mlp_b[s][l] = ptr(out.arrays["%s/%s/biases" % (scope, layer)])
        w.num_classes = self.num_classes
        tap_scale = np.ones(FEAT_DIM, np.float32)
        span = np.zeros(23, np.float32)
        rc = lib().disn_equalise_weights(C.byref(w), ptr(tap_scale), ptr(span))
        if rc:
            raise RuntimeError("disn_equalise_weights failed (status %d)" % rc)
        return out, tap_scale, span

    def __getitem__(self, k: str) -> np.ndarray:
        return self.arrays[k]

    def __contains__(self, k: str) -> bool:
        return k in self.arrays

    def keys(self):
        return self.arrays.keys()

    def items(self):
        return self.arrays.items()

    def n_params(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.shapes.values()))
