"""CPU tests of the host logic and the C-ABI surface (no GPU compute).  (-m "not gpu")"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from disn_amd.csrc import build
    return build.build()


def test_library_exports_every_declared_symbol():
    """the C-ABI library loads and exports exactly what include/disn_amd.h declares"""
    path = _build()
    hdr = open(os.path.join(ROOT, "include", "disn_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(disn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    h = ctypes.CDLL(path)
    for name in sorted(declared):
        assert hasattr(h, name), "library does not export %s" % name
    from disn_amd import _lib
    assert set(_lib.SIGNATURES) == declared, (set(_lib.SIGNATURES) ^ declared)
    assert _lib.lib().disn_abi_version() == _lib.ABI_VERSION == 10


def test_code_object_targets_gfx950():
    path = _build()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]   # single-arch build


def test_argument_validation_without_gpu():
    """error convention: <0 for invalid arguments, never a crash; checked before any launch"""
    from disn_amd import _lib
    h = _lib.lib()
    assert h.disn_pack_kn(None, 32, 64, 32, None, None) == -1
    assert h.disn_pack_kn(1, 32, 48, 32, 1, None) == -2            # N % 32
    assert h.disn_resize_bilinear(None, 1, 2, 2, 3, None, 4, 4, 3, 0, None) == -1
    assert h.disn_resize_bilinear(1, 1, 2, 2, 3, 1, 4, 4, 2, 0, None) == -2   # coff + C > cstride
    assert h.disn_conv3x3(1, 1, 8, 8, 5, 1, 1, 64, 1, 1, None, 0, None) == -2  # Cin not 3 / %32
    assert h.disn_conv3x3(1, 1, 8, 8, 32, 1, 1, 48, 1, 1, None, 0, None) == -2  # Cout % 64
    assert h.disn_fc(1, 1, 64, 1, 1, 100, 0, 1, 1, 1 << 20, None) == -2         # N % 256
    assert h.disn_dense(1, 64, 48, None, 0, 0, 8, 1, 1, 64, 1, 1, None, 0, None) == -2
    assert h.disn_project(None, None, 1, 1, None, None) == -1
    assert h.disn_gather(1, 1, 0, 4, 1, None) == -1
    assert h.disn_vgg16_forward(None, None, 1, None, None, None, None, 0, None) == -1
    p6 = (ctypes.c_double * 6)(-1, -1, -1, 1, 1, 1)
    assert h.disn_grid_points(ctypes.byref(p6), 4, 0, 126, 1, None) == -1      # k1 > 5^3
    assert h.disn_grid_points(ctypes.byref(p6), 0, 0, 1, 1, None) == -1
    # workspace queries are pure host arithmetic
    assert h.disn_vgg16_workspace_bytes(1) > 12 * 2**20
    assert h.disn_vgg16_workspace_bytes(8) > h.disn_vgg16_workspace_bytes(1)
    assert h.disn_query_workspace_bytes(1, 2048) > 2048 * 1472 * 4
    assert h.disn_query_workspace_bytes(0, 5) == 0
    # folded local stream: NULL weights / maps, and weights without the split fold2/conv1 halves
    assert h.disn_fold_local_workspace_bytes() > 2048
    assert h.disn_fold_local(None, 1, 1, 1, 1 << 30, None) == -1
    w = _lib.MlpWeights()
    for f in _lib.MLP_FIELDS:
        setattr(w, f, 1)
    assert h.disn_fold_local(ctypes.byref(w), 1, 1, 1, 1 << 30, None) == -1        # l_w4_point / l_w4_feat unset
    assert h.disn_query_folded(ctypes.byref(w), 1, 1, 1, 1, 1, 1, 8, 1, 1, 1 << 30, None) == -1
    assert h.disn_query_grid_folded(ctypes.byref(w), 1, 1, 1, ctypes.byref(p6), 4, 0, 10, 10.0, 1, 1, 1 << 30,
                                    None) == -1
    w.l_w4_point = 1
    w.l_w4_feat = 1
    assert h.disn_fold_local(ctypes.byref(w), 1, 1, 1, 16, None) == -3              # workspace too small
    assert h.disn_query_grid_folded(ctypes.byref(w), 1, 1, 1, ctypes.byref(p6), 4, 0, 126, 10.0, 1, 1, 1 << 30,
                                    None) == -1                                     # k1 > 5^3
    # disn_encode_query takes featmap = NULL (map not materialised) but still validates the rest
    assert h.disn_encode_query(None, None, None, 1, 1, 1, 1, 1, 8, None, None, 1, None, 1, 1, 0, None) == -1
    with pytest.raises(_lib.DisnError):
        _lib.check("x", -3)


def test_product_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import torch
    from disn_amd import _lib, ops
    with pytest.raises(TypeError):          # CPU tensors are refused: no CPU fallback
        ops.project(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
    if not torch.cuda.is_available():
        from disn_amd.engine import SdfEngine
        from disn_amd.weights import WeightStore
        with pytest.raises((RuntimeError, AssertionError)):
            SdfEngine(WeightStore(num_classes=1024))
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.DisnLibraryError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "disn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+(\.+)?oracle\b", src, flags=re.M), (dp, f)
                    assert not re.search(r"import_module\(|__import__\(|exec\(|dlopen|CDLL\([^)]*oracle", src), (dp, f)
                    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
                    assert "disn_oracle" not in re.sub(r'""".*?"""', "", code, flags=re.S), (dp, f)
                else:
                    assert not re.search(r"#include\s+[\"<][^\">]*oracle", src), (dp, f)


def test_weight_store_namespace_and_restore_semantics(tmp_path):
    from disn_amd.weights import WeightStore, variable_shapes
    from oracle import disn_oracle as O
    assert variable_shapes() == O.variable_shapes()
    ws = WeightStore.random_init(0)
    assert ws.complete() and ws.n_params() == 140819266
    ref = O.init_weights(0, "xavier")
    for k in ref:
        assert np.array_equal(ws[k], ref[k]), k           # same generator order as the oracle's init
    he = WeightStore.random_init(0, mode="he")
    assert np.array_equal(he["vgg_16/fc8/biases"], O.init_weights(0, "he")["vgg_16/fc8/biases"])
    # biases zero (utils/tf_util.py:173 constant_initializer(0.0))
    assert not ws["sdfprediction/fold1/conv1/biases"].any()
    # restore: prefix + exact shape match, others skipped (train/train_sdf.py:196-205)
    other = WeightStore(num_classes=1024)
    n = other.assign({"vgg_16/conv1/conv1_1/weights": ws["vgg_16/conv1/conv1_1/weights"],
                      "vgg_16/fc8/weights": np.zeros((1, 1, 4096, 1000), np.float32),   # shape mismatch
                      "unrelated/var": np.zeros(3, np.float32)}, prefix="vgg_16")
    assert n == 1 and not other.complete()
    with pytest.raises(ValueError):
        other.assign({"vgg_16/fc8/weights": np.zeros((1, 1, 4096, 1000), np.float32)}, strict=True)
    small = {k: v for k, v in ws.items() if k.startswith("sdfprediction")}
    p = str(tmp_path / "dec.npz")
    np.savez(p, **small)
    with pytest.raises(KeyError):
        WeightStore.load(p)                                 # incomplete checkpoint is an error when strict
    part = WeightStore.load(p, strict=False)
    assert len(list(part.keys())) == len(small)


def test_graph_surface_matches_reference_signatures():
    import inspect
    import disn_amd.model_normalization as model
    import disn_amd.sdfnet as sdfnet
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(model.placeholder_inputs) == ["batch_size", "num_points", "img_size", "num_sample_pc", "scope", "FLAGS"]
    assert sig(model.get_model) == ["ref_dict", "num_point", "is_training", "bn", "bn_decay", "img_size", "wd", "FLAGS"]
    assert sig(model.get_loss) == ["end_points", "sdf_weight", "regularization", "mask_weight",
                                   "num_sample_points", "FLAGS", "batch_size"]
    assert sig(model.get_decoder) == ["num_point", "input_pls", "feature_pls", "bn", "bn_decay", "wd"]
    assert sig(model.get_img_points) == ["sample_pc", "trans_mat_right"]
    assert sig(model.placeholder_features) == ["batch_size", "num_sample_pc", "scope"]
    assert sig(sdfnet.get_sdf_basic2) == ["src_pc", "globalfeats", "is_training", "batch_size", "num_point",
                                          "bn", "bn_decay", "wd"]
    assert sig(sdfnet.get_sdf_basic2_imgfeat_twostream) == ["src_pc", "point_feat", "is_training", "batch_size",
                                                             "num_point", "bn", "bn_decay", "wd"]
    pls = model.placeholder_inputs(2, 1, (137, 137), num_sample_pc=100)
    assert set(pls) == {"pc", "sample_pc", "sample_pc_rot", "imgs", "sdf", "sdf_params", "trans_mat"}
    assert pls["imgs"].get_shape() == (2, 137, 137, 3) and pls["trans_mat"].get_shape() == (2, 4, 3)
    ep = model.get_model(pls, 1, None, bn=False)
    loss, ep = model.get_loss(ep, num_sample_points=100, batch_size=2)
    for k in ("pred_sdf", "ref_img", "sample_img_points", "ref_sdf", "ref_pc", "resized_ref_img", "img_embedding",
              "pred_sdf_value_global", "pred_sdf_value_local", "ref_feats_embedding_cnn", "point_img_feat",
              "weighed_mask"):
        assert k in ep, k
    assert set(ep["losses"]) == {"accuracy", "sdf_loss_realvalue", "sdf_loss", "regularization", "overall_loss"}
    assert ep["ref_img"] is pls["imgs"]                     # the UN-resized input (model_normalization.py:62)
    assert ep["pred_sdf"].get_shape() == (2, 100, 1)
    assert ep["point_img_feat"].get_shape() == (2, 100, 1, 1472)

    class F:
        binary = True
    with pytest.raises(NotImplementedError):
        model.get_model(pls, 1, None, FLAGS=F)
    with pytest.raises(NotImplementedError):
        model.get_model(pls, 1, None, bn=True)


def test_create_sdf_host_helpers(pins, tmp_path):
    from disn_amd import create_sdf as cs
    for r, total, split, nsp in pins["split_plans"]:
        assert cs.split_plan(int(r))[:3] == (total, split, nsp)
    for tag in ("a", "b"):
        assert np.array_equal(cs.grid_points_host(pins["grid_%s_params" % tag], int(pins["grid_%s_res" % tag])),
                              pins["grid_%s_pts" % tag])
    p = str(tmp_path / "x.dist")
    cs.to_binary(int(pins["dist_res"]), pins["dist_pos"], pins["dist_vals"], p)
    assert np.array_equal(np.frombuffer(open(p, "rb").read(), np.uint8), pins["dist_bytes"])
    res, pos, vals = cs.read_dist(p)
    assert res == 4 and np.array_equal(pos, pins["dist_pos"]) and np.array_equal(vals.ravel(), pins["dist_vals"])


def test_bench_and_entry_contract_static():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("--gpus", "--steps", "--warmup", '"roofline"', '"cpu_baseline"', '"vs_baseline"', "ms_per_step"):
        assert key in src, key
    ent = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "def build(" in ent and "def smoke(" in ent


def test_product_library_has_no_tuning_knobs():
    """no environment variable is read and no knob is exported by the product library:
    the knobs of csrc/tuning.hpp exist only in `build.py --tuning` builds"""
    path = _build()
    h = ctypes.CDLL(path)
    assert not hasattr(h, "disn_tuning_set")
    blob = open(path, "rb").read()
    assert b"getenv" not in blob
    for name in (b"DISN_X3", b"DISN_OVERLAP", b"DISN_BF_SPLITS", b"DISN_BF16_SKIP_PACK", b"DISN_GEMM_FORCE"):
        assert name not in blob, name
    csrc = os.path.join(ROOT, "disn_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".hpp")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_equalised_weights_are_the_same_network():
    """disn_equalise_weights (WeightStore.equalised; host routine, no GPU): every hidden channel times a power of two,
    its consumers' rows divided by it (models/model_normalization.py:74-78,171-204, models/sdfnet.py:71-88,173-186).
    (1) the float32 oracle returns the SAME BITS for pred_sdf / the embedding on the equalised copy -- powers of two
    commute with every fp32 rounding on the way; (2) taps / gathered features differ by exactly tap_scale; (3) on
    trained-like statistics (sigma = 2 channel gains, 1 % outliers x 10^4) every tap's channel maxima collapse from a
    span of > 2^20 to a few binades; (4) a second pass finds nothing left to do; (5) the original store is untouched
    and the tensors the routine never writes are shared, not copied."""
    from disn_amd.weights import WeightStore
    from oracle import disn_oracle as O
    for label, W in (("he", O.init_weights(3, "he")),
                     ("trained-like", O.trained_like_weights(3, sigma=2.0, outlier_gain=1e4))):
        st = WeightStore(W)
        before = {k: v.copy() for k, v in st.arrays.items() if "conv1_2" in k or "fold2/conv1" in k}
        eq, ts, span = st.equalised()
        assert all(np.array_equal(before[k], st.arrays[k]) for k in before)
        assert st.arrays["vgg_16/fc7/weights"] is eq.arrays["vgg_16/fc7/weights"]
        assert not np.shares_memory(st.arrays["vgg_16/fc6/weights"], eq.arrays["vgg_16/fc6/weights"])
        assert ts.shape == (1472,) and span.shape == (23,)
        m, e = np.frexp(ts)
        assert np.all(m == 0.5), "tap factors must be powers of two"
        feed = O.synth_inputs(3, 1, 64)
        a, b = O.get_model(feed, st.arrays), O.get_model(feed, eq.arrays)
        assert np.array_equal(a["pred_sdf"], b["pred_sdf"]) and np.array_equal(a["img_embedding"], b["img_embedding"])
        fa, fb = a["point_img_feat"][0, :, 0, :], b["point_img_feat"][0, :, 0, :]
        assert np.array_equal(fa, fb / ts)
        if label == "he":
            assert span.max() <= 6
        else:
            assert span.max() >= 20
            ca, cb = np.abs(fa).max(0), np.abs(fb).max(0)
            for o, c in ((0, 64), (64, 128), (192, 256), (448, 512), (960, 512)):
                live = ca[o:o + c] > 0
                sa = np.log2(ca[o:o + c][live].max() / ca[o:o + c][live].min())
                sb = np.log2(cb[o:o + c][live].max() / cb[o:o + c][live].min())
                print("tap channels %4d..%4d: log2 span of the channel maxima %.1f -> %.1f" % (o, o + c, sa, sb))
                assert sb <= sa
            assert np.log2(cb[cb > 0].max() / np.median(cb[cb > 0])) <= 8
        eq2, ts2, span2 = eq.equalised()
        assert span2.max() <= 1 and np.all((ts2 == 1.0) | (ts2 == 0.5) | (ts2 == 2.0))


def test_equalisation_does_not_blow_up_a_dead_column():
    """a hidden channel whose weight column is 2^-40 of its neighbours' (a dead feature with an ordinary bias) is scaled
    up only as far as its bias allows: b c stays within the layer's largest bias (ADVICE r5 -- scaled by the full 2^16
    it would set the image's activation scale and cost every other channel of the two-term kernels ~13 bits); with a
    zero bias the cap is 2^16"""
    from disn_amd.weights import WeightStore
    st = WeightStore.random_init(1, mode="he")
    w = st.arrays["vgg_16/conv3/conv3_2/weights"]
    w[:, :, :, 7] *= np.float32(2.0 ** -40)
    eq, ts, span = st.equalised()
    b = st.arrays["vgg_16/conv3/conv3_2/biases"]
    c7 = eq.arrays["vgg_16/conv3/conv3_2/biases"][7] / b[7]
    assert 1.0 <= c7 <= 2.0 ** 16 and np.frexp(c7)[0] == 0.5
    assert abs(b[7]) * c7 <= np.abs(b).max() and (c7 == 2.0 ** 16 or abs(b[7]) * 2 * c7 > np.abs(b).max())
    st = WeightStore.random_init(1, mode="he")
    st.arrays["vgg_16/conv3/conv3_2/weights"][:, :, :, 7] *= np.float32(2.0 ** -40)
    st.arrays["vgg_16/conv3/conv3_2/biases"][7] = 0.0
    eq, ts, span = st.equalised()
    w0, w1 = st.arrays["vgg_16/conv3/conv3_2/weights"][..., 7], eq.arrays["vgg_16/conv3/conv3_2/weights"][..., 7]
    # (the column also carries the inverse factors of conv3_1's channels on its rows: compare one row's ratio with a neighbour column's)
    r7 = (w1 / w0)[0, 0, :] / (eq.arrays["vgg_16/conv3/conv3_2/weights"][..., 8] / st.arrays["vgg_16/conv3/conv3_2/weights"][..., 8])[0, 0, :]
    assert np.all(np.isfinite(r7)) and np.log2(r7.max()) >= 10
    nxt = eq.arrays["vgg_16/conv3/conv3_3/weights"][:, :, 7, :] * np.float32(2.0 ** 16)
    ref = st.arrays["vgg_16/conv3/conv3_3/weights"][:, :, 7, :]
    c_out = eq.arrays["vgg_16/conv3/conv3_3/biases"] / st.arrays["vgg_16/conv3/conv3_3/biases"]
    assert np.array_equal(nxt, ref * c_out[None, None, :])
    assert h_status(eq) == 0


def h_status(store):
    """disn_equalise_weights argument validation"""
    import ctypes as C
    from disn_amd import _lib
    w = _lib.EqWeights()
    assert _lib.lib().disn_equalise_weights(C.byref(w), None, None) == -1
    ts = np.ones(1472, np.float32)
    assert _lib.lib().disn_equalise_weights(C.byref(w), ts.ctypes.data_as(C.c_void_p), None) == -1   # null tensors
    assert _lib.lib().disn_scale_channels(None, 1, 64, None, 0, None, None) == -1
    assert _lib.lib().disn_scale_channels(1, 1, 6, 1, 0, 1, None) == -2
    return 0


def test_header_is_c99_and_struct_layouts_match_the_ctypes_binding(tmp_path):
    """include/disn_amd.h is the boundary a C caller compiles against: it must be valid C99 (gcc -std=c99 -pedantic), a C
    program must link against the library and reach its host-only entry points, and the struct layouts the C compiler
    sees must be the ones disn_amd/_lib.py declares to ctypes (a field added on one side only would shift every later
    member: disn_vgg_weights_t.strict_forms, disn_eq_weights_t were added in ABI 9)"""
    import ctypes as C
    from disn_amd import _lib
    lib_path = _build()
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#include "disn_amd.h"
int main(void) {
  disn_vgg_weights_t v; disn_mlp_weights_t m; disn_eq_weights_t e; disn_param_layout_t p; disn_cam_weights_t c;
  memset(&v, 0, sizeof v); memset(&m, 0, sizeof m); memset(&e, 0, sizeof e); (void)p; (void)c;
  printf("abi %d %d\n", DISN_ABI_VERSION, disn_abi_version());
  printf("vgg %zu %zu %zu %zu\n", sizeof v, offsetof(disn_vgg_weights_t, num_classes), offsetof(disn_vgg_weights_t, fc_w_t),
         offsetof(disn_vgg_weights_t, strict_forms));
  printf("mlp %zu %zu\n", sizeof m, offsetof(disn_mlp_weights_t, l_feat));
  printf("eq %zu %zu %zu\n", sizeof e, offsetof(disn_eq_weights_t, mlp_w), offsetof(disn_eq_weights_t, num_classes));
  printf("layout %zu cam %zu\n", sizeof p, sizeof c);
  printf("crc %u\n", disn_crc32c("123456789", 9, 0));
  printf("args %d %d %zu\n", disn_equalise_weights(&e, NULL, NULL), disn_scale_channels(NULL, 1, 64, NULL, 0, NULL, NULL),
         disn_encode_query_workspace_bytes(1, 2048) > 0 ? (size_t)1 : (size_t)0);
  return 0;
}
''')
    exe = tmp_path / "abi"
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe),
                        lib_path, "-Wl,-rpath," + os.path.dirname(lib_path), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr[-2000:])
    f = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.strip()}
    assert f["abi"] == [str(_lib.ABI_VERSION)] * 2
    V, M, E = _lib.VggWeights, _lib.MlpWeights, _lib.EqWeights
    assert [int(x) for x in f["vgg"]] == [C.sizeof(V), V.num_classes.offset, V.fc_w_t.offset, V.strict_forms.offset]
    assert [int(x) for x in f["mlp"]] == [C.sizeof(M), M.l_feat.offset]
    assert [int(x) for x in f["eq"]] == [C.sizeof(E), E.mlp_w.offset, E.num_classes.offset]
    assert [int(f["layout"][0]), int(f["layout"][2])] == [C.sizeof(_lib.ParamLayout), C.sizeof(_lib.CamWeights)]
    assert int(f["crc"][0]) == 0xE3069283                 # CRC-32C check value of "123456789"
    assert f["args"] == ["-1", "-1", "1"]
