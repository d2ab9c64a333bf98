"""CPU checks of the conv_h2w.hip layout through its lane-level restatement (tests/conv_h2w_emulation.py): every
variant reproduces a float64 3x3 SAME convolution on ragged shapes, stores every output exactly once, pools
correctly, reads only LDS bytes that were written, and its A-fragment reads are (nearly) bank-conflict-free."""
import numpy as np
import pytest

import conv_h2w_emulation as E
from test_conv_h2_emulation import case, ref_conv


@pytest.mark.parametrize("variant,H,W,cin,cout", [(1, 10, 34, 64, 64), (2, 8, 28, 64, 128), (2, 12, 30, 64, 128),
                                                  (3, 6, 28, 64, 64), (4, 10, 28, 64, 128), (5, 8, 30, 128, 64),
                                                  (6, 8, 28, 128, 128), (7, 10, 30, 128, 64)])
def test_every_variant_reproduces_the_convolution(variant, H, W, cin, cout):
    x, w, b = case(H, W, cin, cout, seed=variant)
    out, pooled, vmax = E.conv(x, w, b, variant)
    ref = ref_conv(x, w, b)
    assert not np.isnan(out).any(), "an output element was never stored"
    scale = np.abs(ref).max()
    err = np.abs(out - ref).max()
    assert err <= 3e-7 * scale, (err, scale)         # two-term split: ~2^-22 per operand
    pref = ref.reshape(H // 2, 2, W // 2, 2, -1).max(axis=(1, 3))
    assert not np.isnan(pooled).any()
    assert np.abs(pooled - pref).max() <= 3e-7 * scale
    assert abs(vmax - scale) <= 3e-7 * scale


@pytest.mark.parametrize("variant", sorted(E.VARIANTS))
def test_a_fragment_reads_are_nearly_bank_conflict_free(variant):
    worst, total, reads = E.lds_read_conflicts(variant)
    assert worst <= 1                      # never more than one extra LDS cycle on a 4-cycle read
    if E.VARIANTS[variant][5] == 32:
        assert total == 0                  # 32-pixel patches: four windows never straddle a row pair
    else:
        assert total <= 0.15 * reads       # 28-pixel patches: the groups that straddle a row pair (1 in 7)


def test_geometry_fits_the_hardware():
    for v, (MB, MWV, NWV, WK, TH, TW) in E.VARIANTS.items():
        G = E.geometry(v)
        assert TH * TW == 32 * MB * MWV
        assert G["ROWB"] % 256 == 128 and G["ROWB"] >= G["RP"] * G["KPIX"]
        assert (G["KPIX"] // 16) % 2 == 1
        lds = max(2 * G["BUF"], MWV * NWV * MB * 4096 if WK == 2 else 0)
        occ = 2 if WK == 1 else 1
        if E.SEGMENTED.get(v) == "park":   # p0 of every wave parked BEHIND the halo buffers, one workgroup per CU
            lds, occ = 2 * G["BUF"] + MWV * NWV * WK * MB * 4096, 1
        elif v in E.SEGMENTED:
            occ = 1
        assert lds * occ <= 160 * 1024
        # the largest tap offset is an immediate of ds_read (16 bits)
        assert 2 * G["ROWB"] + 2 * G["KPIX"] + G["CK"] * 2 < 65536
        # a thread's halo units fit the sub-steps that carry them
        assert -(-G["UNITS"] // G["NT"]) <= 5 * MB


@pytest.mark.parametrize("cin", [128, 256, 512])
def test_the_two_segmented_variants_sum_the_same_things_in_the_same_order(cin):
    """variant 6 (one k-wave, p0 parked at the midpoint) and variant 7 (k-wave w = K half w): the same two halves, the
    same two-chunk segments, every k16 block exactly once -- what makes tilings 12 and 13 bit-identical on the GPU"""
    p6, p7 = E.segment_plan(6, cin), E.segment_plan(7, cin)
    assert p6 == p7
    blocks = [b for half in p6 for seg in half for b in seg]
    assert sorted(blocks) == list(range(cin // 16)) and all(len(seg) == 2 for half in p6 for seg in half)
    # the walks: PARK visits chunk c = block c; HALVES stages blocks c and cin / 32 + c in chunk c (k-wave 0 / 1)
    assert [E.chunk_blocks(6, cin, c)[0] for c in range(cin // 16)] == blocks
    assert [E.chunk_blocks(7, cin, c)[0] for c in range(cin // 32)] == [b for seg in p7[0] for b in seg]
    assert [E.chunk_blocks(7, cin, c)[1] for c in range(cin // 32)] == [b for seg in p7[1] for b in seg]
