"""CPU checks of the lane-level restatement of dense_h2.hip (tests/dense_h2_emulation.py): every tile shape
reproduces a float64 product, stores every output exactly once, gives the same value whatever the tile shape
(the summation tree is the tiling's only numerical degree of freedom and it is fixed), handles two sources, the
deferred bias + ReLU and per-image scales; A-fragment reads are bank-conflict free.  And the executable
specification of the producer-side split: split-form operands with bound-derived scales,
two sources with different scales, give the same accuracy with a loader that does no arithmetic."""
import numpy as np
import pytest

import dense_h2_emulation as E


def case(M, K, N, seed, k2=0):
    rng = np.random.default_rng(seed)
    a = np.maximum(rng.standard_normal((M, K + k2)), 0).astype(np.float32) * np.float32(2.0)
    w = (rng.standard_normal((K + k2, N)) * np.sqrt(2.0 / (K + k2))).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    return a, w, b


@pytest.mark.parametrize("MB,NW,KPW,M,K,N", [(1, 2, 1, 40, 64, 64), (2, 2, 1, 70, 128, 64), (2, 4, 2, 64, 128, 128),
                                            (4, 2, 2, 128, 256, 64), (2, 2, 4, 64, 256, 64)])
def test_every_tile_shape_reproduces_the_product(MB, NW, KPW, M, K, N):
    a, w, b = case(M, K, N, MB + NW + KPW)
    out = E.dense(a, None, w, b, MB, NW, KPW)
    ref = np.maximum(a.astype(np.float64) @ w.astype(np.float64) + b, 0)
    assert not np.isnan(out).any()                         # every element stored (exactly once: asserted inside)
    assert np.abs(out - ref).max() <= 2.0 ** -21 * np.abs(ref).max()


def test_tile_shapes_agree_to_the_last_bit_of_the_model():
    """the emulation accumulates in float64, so equal values here mean: same operand bits, same summation tree"""
    a, w, b = case(64, 256, 128, 3)
    outs = [E.dense(a, None, w, b, MB, NW, KPW) for MB, NW, KPW in ((1, 2, 1), (2, 2, 4), (2, 4, 2), (2, 4, 1))]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_two_sources_and_deferred_bias():
    a, w, b = case(64, 128, 64, 5, k2=64)
    a1, a2 = np.ascontiguousarray(a[:, :128]), np.ascontiguousarray(a[:, 128:])
    out = E.dense(a1, a2, w, b, 2, 2, 1)
    ref = np.maximum(a.astype(np.float64) @ w.astype(np.float64) + b, 0)
    assert np.abs(out - ref).max() <= 2.0 ** -21 * np.abs(ref).max()
    rng = np.random.default_rng(9)
    pre = rng.standard_normal((64, 128)).astype(np.float32)
    ib = rng.standard_normal(128).astype(np.float32)
    w2, b2 = w[:128], b
    out = E.dense(pre, None, w2, b2, 2, 2, 2, in_bias=ib)
    ref = np.maximum(np.maximum(pre + ib, 0).astype(np.float64) @ w2.astype(np.float64) + b2, 0)
    assert np.abs(out - ref).max() <= 2.0 ** -20 * np.abs(ref).max()


def test_per_image_scales_make_an_image_independent_of_its_batch():
    a, w, b = case(128, 64, 64, 11)
    a[64:] *= np.float32(37.0)                               # second image much brighter
    both = E.dense(a, None, w, b, 2, 2, 1, rows_per_image=64)
    first = E.dense(a[:64], None, w, b, 2, 2, 1)
    assert np.array_equal(both[:64], first)
    shared = E.dense(a, None, w, b, 2, 2, 1)                  # one scale for the batch: the dim image changes
    assert not np.array_equal(shared[:64], first)


@pytest.mark.parametrize("MB,KPW", [(1, 1), (2, 1), (2, 2), (2, 4), (4, 2)])
def test_a_fragment_reads_are_bank_conflict_free(MB, KPW):
    reads, writes = E.lds_conflicts(MB, KPW)
    assert reads == 0
    assert writes <= 1          # the loader's ds_write_b64 pairs: at most two-way (two units of one row per 32 lanes)


def test_presplit_operands_with_bound_scales_keep_the_accuracy():
    """the specification of the next kernel step: producer-side split, scales from bounds 2^5 / 2^7 above the true
    maxima and DIFFERENT for the two sources; the loader copies, the accumulators are rescaled once"""
    a, w, b = case(64, 128, 64, 21, k2=64)
    a1, a2 = np.ascontiguousarray(a[:, :128]), np.ascontiguousarray(a[:, 128:]) * np.float32(0.01)
    full = np.concatenate([a1, a2], axis=1)
    ref = np.maximum(full.astype(np.float64) @ w.astype(np.float64) + b, 0)
    s1 = E.pow2_scale(np.abs(a1).max() * 2.0 ** 5, 14)       # what a producer would pick from its bound
    s2 = E.pow2_scale(np.abs(a2).max() * 2.0 ** 7, 14)
    assert s1 != s2
    out = E.dense(E.to_split_form(a1, s1), E.to_split_form(a2, s2), w, b, 2, 2, 1, presplit=(s1, s2))
    assert np.abs(out - ref).max() <= 2.0 ** -20 * np.abs(ref).max()
    today = E.dense(a1, a2, w, b, 2, 2, 1)
    assert np.abs(today - ref).max() <= 2.0 ** -20 * np.abs(ref).max()
