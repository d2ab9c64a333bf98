"""Trained-like statistics through the two-term f16 split (VERDICT r3 #2c): heavy-tailed weights, log-normal channel
gains with 1 % outlier channels x 10^3, white-background images -- against the float64 oracle run stored in
tests/golden/stress_trained_like.npz (tests/golden/make_golden_stress.py; nothing in it comes from the GPU).  The
per-image power-of-two activation scale (largest magnitude -> 2^14) is what such inputs stress: channels a thousand
times below the tensor's maximum must keep their ~22 bits (they do as long as the scaled value stays above 2^-3: the
l term absorbs the residual down to f16's subnormal floor, tests/test_split_scale_bound.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import disn_oracle as O

pytestmark = pytest.mark.gpu
PRED_ATOL = 1e-5

sys.path.insert(0, GOLDEN)
import make_golden_stress as S   # noqa: E402


@pytest.fixture(scope="module")
def stress():
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    gold = dict(np.load(os.path.join(GOLDEN, "stress_trained_like.npz")))
    inp = S.stress_inputs()
    eng = SdfEngine(WeightStore(O.trained_like_weights(S.STRESS_SEED)))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return dict(gold=gold, inp=inp, eng=eng, dev=dev)


def _report(label, got, ref):
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    print("\n[parity stress %s] max |gpu - f64| %.3g (|ref| max %.3g)" % (label, err, float(np.abs(ref).max())))
    return err


def test_single_step_form_on_trained_like_statistics(stress):
    """one image per call: conv_h2.hip / dense_h2.hip (K parallel inside the workgroup)"""
    s, g = stress, stress["gold"]
    for b in (0, 1):
        enc, pred = s["eng"].encode_query(s["dev"](s["inp"]["imgs"][b:b + 1]), s["dev"](s["inp"]["pts_a"][b:b + 1]),
                                          s["dev"](s["inp"]["trans_mat"][b:b + 1]))
        e = _report("single step, image %d" % b, pred.cpu().numpy().reshape(-1), g["pred64_a"][b])
        ee = _report("single step, image %d, embedding" % b, enc.embedding.cpu().numpy()[0], g["emb64"][b])
        assert e <= PRED_ATOL
        assert ee <= 1e-5 + 2e-6 * float(np.abs(g["emb64"][b]).max())


def test_batched_form_on_trained_like_statistics(stress):
    """four images per call: conv_h2w.hip / dense_h2w.hip (K sequential); taps against the oracle's at their scale"""
    s, g = stress, stress["gold"]
    enc, pred = s["eng"].encode_query(s["dev"](s["inp"]["imgs"]), s["dev"](s["inp"]["pts_a"]), s["dev"](s["inp"]["trans_mat"]))
    worst = 0.0
    for b in range(4):
        worst = max(worst, _report("batched call, image %d" % b, pred[b].cpu().numpy().reshape(-1), g["pred64_a"][b]))
    assert worst <= PRED_ATOL
    stride = int(g["tap_stride"])
    for t, nm in zip(s["eng"].true_taps(enc), O.TAP_NAMES):     # (the engine computes in equalised units)
        got = t.cpu().numpy()[[0, 3]].reshape(2, -1)[:, ::stride].astype(np.float64)
        for i in range(2):
            scale = float(g["tapmax_" + nm][i])
            err = float(np.abs(got[i] - g["tap64_" + nm][i]).max())
            print("[parity stress tap %s image %d] max err %.3g = %.3g of the tap's maximum %.3g" % (
                nm, (0, 3)[i], err, err / scale, scale))
            assert err <= 2e-6 * scale + 1e-5


def test_fused_point_mlp_on_trained_like_statistics(stress):
    """40 960 points of one image: folded feature map + mlp_fused.hip (per-point activation scales)"""
    s, g = stress, stress["gold"]
    enc = s["eng"].encode(s["dev"](s["inp"]["imgs"][:1]))
    pred = s["eng"].query(enc, s["dev"](s["inp"]["pts_b"]), s["dev"](s["inp"]["trans_mat"][:1]))
    e = _report("fused point MLP, 40960 points", pred.cpu().numpy().reshape(-1), g["pred64_b"])
    assert e <= PRED_ATOL
