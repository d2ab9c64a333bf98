import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test on a box without a GPU is an error of the invocation, not a pass
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pins():
    with np.load(os.path.join(GOLDEN, "reference_pins.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def kat():
    with np.load(os.path.join(GOLDEN, "oracle_kat.npz")) as z:
        return {k: z[k] for k in z.files}


_INTERNAL = {}


def internal_arrays(store):
    """the variables in the units an SdfEngine built from `store` computes in (WeightStore.equalised: hidden channels
    times powers of two, consumers' rows divided -- the same function): what an oracle continuation has to use when it
    starts from the engine's OWN taps / feature map (cached per store)"""
    key = id(store)
    if key not in _INTERNAL:
        _INTERNAL.clear()                      # (one store at a time: the copies are ~0.5 GB)
        _INTERNAL[key] = (store, store.equalised()[0].arrays)
    return _INTERNAL[key][1]


def report_close(name, got, ref, atol, rtol=0.0):
    """assert |got-ref| <= atol + rtol*|ref| with a diagnostic that localises GEMM bugs."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = ~(err <= tol)           # NaN counts as bad
    if bad.any():
        idx = np.argwhere(bad)
        worst = np.unravel_index(np.nanargmax(np.where(np.isnan(err), np.inf, err)), err.shape)
        msg = ["%s: %d/%d elements out of tolerance (atol %g rtol %g)" % (name, bad.sum(), bad.size, atol, rtol),
               "  max abs err %g at %s: got %r ref %r" % (np.nanmax(err), worst, got[worst], ref[worst]),
               "  first bad idx %s; nan in got: %d" % (idx[:4].tolist(), int(np.isnan(got).sum()))]
        if got.ndim >= 2:
            g2 = bad.reshape(-1, bad.shape[-1])
            rows = np.nonzero(g2.any(1))[0]
            cols = np.nonzero(g2.any(0))[0]
            msg.append("  bad rows: %d (first %s) bad cols: %d (first %s)" % (
                len(rows), rows[:8].tolist(), len(cols), cols[:8].tolist()))
        pytest.fail("\n".join(msg))
    return float(np.nanmax(err)) if err.size else 0.0
