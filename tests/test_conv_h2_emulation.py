"""CPU checks of the conv_h2.hip layout through its lane-level restatement (tests/conv_h2_emulation.py):
every tiling reproduces a float64 3x3 SAME convolution, stores every output exactly once, pools correctly,
its A-fragment reads are bank-conflict-free, and the two-term f16 split is fp32-accurate."""
import numpy as np
import pytest

import conv_h2_emulation as E


def ref_conv(x, w, b, relu=True):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin))
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[-1]))
    for dy in range(3):
        for dx in range(3):
            out += xp[dy:dy + H, dx:dx + W] @ w[dy, dx].astype(np.float64)
    out += b
    return np.maximum(out, 0) if relu else out


def case(H, W, cin, cout, seed=0):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((H, W, cin)), 0).astype(np.float32) * 3.0
    x[rng.random((H, W, cin)) < 0.3] *= 1e-3          # a wide dynamic range inside one tensor
    w = (rng.standard_normal((3, 3, cin, cout)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("tiling,H,W", [(1, 4, 14), (1, 6, 10), (2, 4, 28), (3, 2, 30), (4, 8, 16), (4, 10, 20), (5, 4, 16),
                                        (5, 6, 20)])
def test_every_tiling_reproduces_the_convolution(tiling, H, W):
    x, w, b = case(H, W, 64, 64, seed=tiling)
    out, pooled, vmax = E.conv(x, w, b, tiling)
    ref = ref_conv(x, w, b)
    assert not np.isnan(out).any(), "an output element was never stored"
    scale = np.abs(ref).max()
    err = np.abs(out - ref).max()
    assert err <= 3e-7 * scale, (err, scale)         # two-term split: ~2^-22 per operand
    pref = ref.reshape(H // 2, 2, W // 2, 2, -1).max(axis=(1, 3))
    assert not np.isnan(pooled).any()
    assert np.abs(pooled - pref).max() <= 3e-7 * scale
    assert abs(vmax - scale) <= 3e-7 * scale


@pytest.mark.parametrize("tiling,cin", [(1, 128), (1, 256), (3, 128), (2, 128)])
def test_several_chunks_and_both_k_wave_counts(tiling, cin):
    """Cin = 128: one 128-channel chunk of eight k-waves (tilings 1, 2) or two 64-channel chunks of four"""
    x, w, b = case(2, 14 if tiling == 1 else 28, cin, 64, seed=7)
    out, _, _ = E.conv(x, w, b, tiling)
    ref = ref_conv(x, w, b)
    assert np.abs(out - ref).max() <= 3e-7 * np.abs(ref).max()


def test_exact_operands_give_the_exact_convolution():
    """with the activation split switched off only the weights' two-term split remains"""
    x, w, b = case(4, 14, 64, 64, seed=3)
    out, _, _ = E.conv(x, w, b, 1, exact_operands=True)
    ref = ref_conv(x, w, b)
    assert np.abs(out - ref).max() <= 2e-7 * np.abs(ref).max()


@pytest.mark.parametrize("tiling,wk", [(1, 4), (2, 4), (3, 4), (4, 4), (1, 8), (2, 8), (5, 4)])
def test_a_fragment_reads_are_bank_conflict_free(tiling, wk):
    assert E.lds_read_conflicts(tiling, wk) == 0


def test_pow2_scale_and_sigma():
    assert sorted(E.sigma(i) for i in range(32)) == list(range(32))
    for a in (1e-20, 3e-5, 0.7, 1.0, 1.5, 123.0, 6e4):
        s = float(E.pow2_scale(a, 14))
        assert 2.0 ** 14 <= a * s < 2.0 ** 15 and np.log2(s) == int(np.log2(s))
    assert float(E.pow2_scale(0.0, 14)) == 1.0
