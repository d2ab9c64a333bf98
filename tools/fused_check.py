"""Counted LDS-DMA waits vs wait-for-everything in the fused point-MLP kernels (mlp_fused.hip).

Needs the tuning build:  python -m disn_amd.csrc.build --tuning
                         DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so python tools/fused_check.py
Runs the same queries with tune::fused_safe = 0 (product behaviour) and 1 (s_waitcnt vmcnt(0) at every
sync) and demands bit-identical results, repeatedly and under a concurrent HBM stream (uneven load)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from disn_amd import _lib
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    from oracle import disn_oracle as O
    import ctypes as C
    h = _lib.lib()
    setk = h.disn_tuning_set
    setk.restype, setk.argtypes = C.c_int, [C.c_int, C.c_int]
    eng = SdfEngine(WeightStore.random_init(2, mode="he"))
    rng = np.random.default_rng(0)
    enc = eng.encode(rng.random((1, 137, 137, 3), dtype=np.float32))
    eng.featmap_of(enc)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    bad = 0
    for n in (128, 5000, 65536, 400003):
        pts = torch.from_numpy(rng.uniform(-1, 1, (1, n, 3)).astype(np.float32)).cuda()
        assert setk(4, 1) == 0
        ref = eng.query(enc, pts, O.DEMO_TRANS_MAT, fold=True, fused=True).clone()
        assert setk(4, 0) == 0
        for rep in range(6):
            if rep >= 3:    # uneven load: an HBM stream on another HIP stream while the kernel runs
                with torch.cuda.stream(side):
                    noise.add_(1)
            got = eng.query(enc, pts, O.DEMO_TRANS_MAT, fold=True, fused=True)
            torch.cuda.synchronize()
            nd = int((got != ref).sum())
            if nd:
                bad += 1
                print("n=%d rep=%d: %d of %d values differ, max |d| %.3g" % (n, rep, nd, n, float((got - ref).abs().max())))
        print("n=%d ok" % n, flush=True)
    # the FEAT form (small point sets from the taps: disn_query_taps_fused), several images per launch
    imgs = rng.random((4, 137, 137, 3), dtype=np.float32)
    enc4 = eng.encode(imgs)
    tms = torch.from_numpy(np.repeat(O.DEMO_TRANS_MAT, 4, axis=0)).cuda()
    from disn_amd import ops
    for n in (128, 2048, 8192):
        pts = torch.from_numpy(rng.uniform(-1, 1, (4, n, 3)).astype(np.float32)).cuda()
        assert setk(4, 1) == 0
        ref = ops.query_taps_fused(eng.weights.mlp, enc4.taps, enc4.embedding, tms, pts).clone()
        assert setk(4, 0) == 0
        for rep in range(6):
            if rep >= 3:
                with torch.cuda.stream(side):
                    noise.add_(1)
            got = ops.query_taps_fused(eng.weights.mlp, enc4.taps, enc4.embedding, tms, pts)
            torch.cuda.synchronize()
            nd = int((got != ref).sum())
            if nd:
                bad += 1
                print("feat form n=%d rep=%d: %d values differ, max |d| %.3g" % (n, rep, nd, float((got - ref).abs().max())))
        print("feat form n=4x%d ok" % n, flush=True)
    assert torch.isfinite(ref).all()
    if bad:
        print("FUSED_CHECK_FAILED")
        sys.exit(1)
    print("FUSED_CHECK_OK")


if __name__ == "__main__":
    main()
