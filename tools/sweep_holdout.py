"""HOLD-OUT check of the parity sweep (round 6): weight sets the kernels were never tuned or asserted against -- seeds 41..44 x
sigma {1, 2} x outlier gain {1e3, 1e4} of oracle.trained_like_weights, the sweep's own images / cameras / point sets -- through
the same forms as tests/test_gpu_sweep.py, against float64 goldens computed here.

    python tools/sweep_holdout.py make      CPU, ~1 minute per set: writes tools/_holdout_sweep.npz (git-ignored; it travels
                                            to the GPU box with the snapshot)
    python tools/sweep_holdout.py           GPU: the distribution of max |gpu - f64| per request and form (profiles/r06y_holdout.txt)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden_sweep as MS   # noqa: E402
from oracle import disn_oracle as O   # noqa: E402

SETS = [(s, sg, og) for s in (41, 42, 43, 44) for sg in (1.0, 2.0) for og in (1.0e3, 1.0e4)]
PATH = os.path.join(ROOT, "tools", "_holdout_sweep.npz")

if len(sys.argv) > 1 and sys.argv[1] == "make":
    s = MS.sweep_inputs()
    out = {}
    for i, (seed, sigma, og) in enumerate(SETS):
        p, g, e, o32 = MS.one_set(seed, sigma, og, s)
        out["pred64_%02d" % i], out["grid64_%02d" % i], out["emb64_%02d" % i], out["o32_%02d" % i] = p, g, e, np.float64(o32)
        print("set %2d seed %d sigma %.1f outliers %.0e: |pred| max %.3g, float32 oracle off by %.3g" % (i, seed, sigma, og, np.abs(p).max(), o32), flush=True)
        np.savez_compressed(PATH, **out)
    sys.exit(0)

import torch   # noqa: E402
import test_gpu_sweep as T   # noqa: E402  (the sweep test's own _forms / _errors)
from disn_amd.engine import SdfEngine   # noqa: E402
from disn_amd.weights import WeightStore   # noqa: E402

gold = np.load(PATH)
s = MS.sweep_inputs()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
per_form = {}
for i, (seed, sigma, og) in enumerate(SETS):
    if "pred64_%02d" % i not in gold.files:
        break
    eng = SdfEngine(WeightStore(O.trained_like_weights(seed, sigma=sigma, outlier_gain=og)))
    strict_eng = SdfEngine(None, weights=eng.weights, strict=True)
    forms, grid = T._forms(eng, s, dev, strict_eng)
    e = T._errors(forms, grid, gold, i)
    for f, v in e.items():
        per_form.setdefault(f, []).extend(v)
    print("hold-out set %2d seed %d sigma %.1f outliers %.0e: %s   (the fp32 CPU oracle itself: %.2e)" % (
        i, seed, sigma, og, "  ".join("%s %.2e" % (f, max(v)) for f, v in e.items()), float(gold["o32_%02d" % i])), flush=True)
    del eng, strict_eng
    torch.cuda.empty_cache()
print("distribution of max |gpu - f64| per (hold-out weight set, request):")
for f, v in per_form.items():
    v = np.sort(np.asarray(v))
    print("   %-9s n %4d  median %.2e  p90 %.2e  max %.2e   above 1e-5: %d" % (f, v.size, np.median(v), v[int(0.9 * (v.size - 1))], v[-1], int((v > 1e-5).sum())))
