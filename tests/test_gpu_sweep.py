"""Trained-like stress SWEEP: 8 seeds x sigma {1, 1.5, 2} x outlier gain {1e3, 1e4} of
`oracle.trained_like_weights`, 8 images (four U[0,1) at different brightness, four white-background) per weight set,
through every form of the path -- one request per call, calls of 4 and of 16 requests, the dense-grid form -- against
the float64 oracle run committed in tests/golden/stress_sweep.npz (tests/golden/make_golden_sweep.py; nothing in it
comes from the GPU).  north_star's bar: 1e-5 absolute on pred_sdf (|pred| up to 2-3 here).

What makes this pass at sigma = 2 (channel gains spanning > 2^20 inside one tensor): the engine uploads the EQUALISED
copy of the variables (disn_equalise_weights, include/disn_amd.h) -- the same function with every hidden channel brought
to a common binade by an exact power of two; `test_sweep_without_equalisation_fails_where_the_model_says` shows the same
kernels on the raw variables miss the bar on the hardest set, i.e. the sweep does exercise what it claims to.

Two modes are run on every weight set: the DEFAULT (kernel forms by call size: calls of >= 4 requests take the batched
convolutions -- from round 6 in SEGMENTED accumulation, conv_h2w.hip's SEG form: chains of 54 MFMAs per accumulator, the
segments summed in fp32 VALU adds -- and the fused small-set MLP) and STRICT (disn_vgg_weights_t.strict_forms = 1,
SdfEngine(strict=True): the single-image forms for every call size; a request's taps, embedding and pred_sdf bit for bit
those of the request alone).  EVERY request of EVERY form in EITHER mode: <= 1e-5.

Default: 18 of the 48 weight sets (all of sigma = 2 for four seeds + one of every other (sigma, outlier) pair + the seven
sets that hold the full run's worst requests);
DISN_SWEEP=full runs all 48 (profiles/r05*_sweep_full.json is that run).  The distribution is printed and, when
gpurun_out/ exists, written there as JSON.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import disn_oracle as O

pytestmark = pytest.mark.gpu
BAR = 1e-5
# What is asserted: every case of every form <= BAR (1e-5 on the UN-divided network output; the SDF value is pred_sdf / 10,
# test/create_sdf.py:285).  Measured over all 48 sets (profiles/r06w_sweep_full.json: 1104 default-mode + 960 strict-mode
# requests): single <= 6.2e-6, grid <= 8.4e-6, calls of 4 / 16 requests <= 7.9e-6 / 9.5e-6 (median 2.7e-6, p90 5.1e-6),
# strict <= 8.8e-6; the float32 CPU oracle itself: median 8.6e-6, worst 1.87e-5.  Until round 5 the batched convolutions
# summed K in chains of up to 432 MFMAs per accumulator and 2.7 % of the batched requests sat at 1.0-1.46e-5
# (profiles/r05k_sweep_full.json); tools/ubench/mfma_round.hip measures what a chain costs (rms 6.2 ulp at 432 MFMAs,
# 0.9-1.0 ulp with a restart every 27-54) and conv_h2w.hip's SEG form is that restart.

sys.path.insert(0, GOLDEN)
import make_golden_sweep as MS   # noqa: E402


def _chosen():
    if os.environ.get("DISN_SWEEP", "") == "full":
        return list(range(len(MS.SETS)))
    pick = [i for i, (s, sg, og) in enumerate(MS.SETS) if sg == 2.0 and s in MS.SEEDS[:4]]
    pick += [MS.SETS.index((MS.SEEDS[4], 1.0, 1e3)), MS.SETS.index((MS.SEEDS[5], 1.0, 1e4)),
             MS.SETS.index((MS.SEEDS[6], 1.5, 1e3)), MS.SETS.index((MS.SEEDS[7], 1.5, 1e4))]
    # round 6: + every set that holds a request above 7.5e-6 in the committed full run (profiles/r06w_sweep_full.json), so
    # that the run the driver makes sees the whole tail of the distribution, not a sample of it
    pick += [20, 24, 25, 28, 29, 33, 35]
    return sorted(set(pick))


def _forms(eng, s, dev, strict_eng=None):
    """-> {form: [(image, point set, pred numpy [256])]} + the grid.  strict_eng (same weights, strict_forms = 1): the same
    calls of 4 and 16 requests through the single-image convolution kernels -> 'strict4' / 'strict16'"""
    out = {"single": [], "batch4": [], "batch16": []}
    for b in (0, 4):
        p = eng.encode_query(dev(s["imgs"][b:b + 1]), dev(s["pts"][b, 0][None]), dev(s["trans_mat"][b:b + 1]))[1]
        out["single"].append((b, 0, p[0].cpu().numpy()))
    idx = [0, 1, 4, 5]
    p = eng.encode_query(dev(s["imgs"][idx]), dev(s["pts"][idx, 0]), dev(s["trans_mat"][idx]))[1].cpu().numpy()
    out["batch4"] = [(b, 0, p[i]) for i, b in enumerate(idx)]
    imgs16 = np.concatenate([s["imgs"], s["imgs"]])
    pts16 = np.concatenate([s["pts"][:, 0], s["pts"][:, 1]])
    tms16 = np.concatenate([s["trans_mat"], s["trans_mat"]])
    p = eng.encode_query(dev(imgs16), dev(pts16), dev(tms16))[1].cpu().numpy()
    out["batch16"] = [(i % 8, i // 8, p[i]) for i in range(16)]
    if strict_eng is not None:
        p = strict_eng.encode_query(dev(s["imgs"][idx]), dev(s["pts"][idx, 0]), dev(s["trans_mat"][idx]))[1].cpu().numpy()
        out["strict4"] = [(b, 0, p[i]) for i, b in enumerate(idx)]
        enc16, p16 = strict_eng.encode_query(dev(imgs16), dev(pts16), dev(tms16))
        p16 = p16.cpu().numpy()
        out["strict16"] = [(i % 8, i // 8, p16[i]) for i in range(16)]
        # strict = the single-image forms of every kernel for every call size: image 4's taps, embedding and pred_sdf in the
        # 16-request call are bit for bit those of the request alone
        one = eng.encode(dev(s["imgs"][4:5]))
        assert all(torch.equal(a[4], b[0]) for a, b in zip(enc16.taps, one.taps)), "strict call: taps differ from the request alone"
        assert torch.equal(enc16.embedding[4], one.embedding[0]), "strict call: embedding differs from the request alone"
        alone = [p for b, j, p in out["single"] if b == 4][0]
        assert np.array_equal(p16[4], alone), float(np.abs(p16[4] - alone).max())
    enc = eng.encode(dev(s["imgs"][:1]))
    grid = eng.query_grid(enc, 0, dev(s["trans_mat"][:1]), MS.GRID_PARAMS, MS.GRID_RES, sdf_weight=1.0).cpu().numpy()
    return out, grid


def _errors(forms, grid, gold, i):
    e = {f: [float(np.abs(p.astype(np.float64) - gold["pred64_%02d" % i][b, j]).max()) for b, j, p in lst]
         for f, lst in forms.items()}
    e["grid"] = [float(np.abs(grid.astype(np.float64) - gold["grid64_%02d" % i]).max())]
    return e


def _dist(v):
    v = np.sort(np.asarray(v, np.float64))
    return {"n": int(v.size), "min": float(v[0]), "median": float(np.median(v)), "p90": float(v[int(0.9 * (v.size - 1))]),
            "max": float(v[-1])}


def test_sweep_within_the_bar_on_every_form():
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    gold = np.load(os.path.join(GOLDEN, "stress_sweep.npz"))
    s = MS.sweep_inputs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rows = []
    for i in _chosen():
        if "pred64_%02d" % i not in gold.files:
            pytest.fail("tests/golden/stress_sweep.npz lacks set %d: run tests/golden/make_golden_sweep.py" % i)
        seed, sigma, outlier = MS.SETS[i]
        eng = SdfEngine(WeightStore(O.trained_like_weights(seed, sigma=sigma, outlier_gain=outlier)))
        strict_eng = SdfEngine(None, weights=eng.weights, strict=True)
        forms, grid = _forms(eng, s, dev, strict_eng)
        del strict_eng
        e = _errors(forms, grid, gold, i)
        row = {"set": i, "seed": seed, "sigma": sigma, "outlier_gain": outlier,
               "max_span_log2": eng.weights.status["max_span_log2"],
               "oracle32_minus_f64": float(gold["o32_%02d" % i]), **{f: max(v) for f, v in e.items()},
               "per_case": e}
        rows.append(row)
        print("[parity sweep] set %2d seed %d sigma %.1f outliers %.0e (channel gains span 2^%.0f): single %.2e  batch4 %.2e  "
              "batch16 %.2e  grid %.2e  strict4 %.2e  strict16 %.2e   (the fp32 CPU oracle itself: %.2e)" % (
                  i, seed, sigma, outlier, row["max_span_log2"], row["single"], row["batch4"], row["batch16"], row["grid"],
                  row["strict4"], row["strict16"], row["oracle32_minus_f64"]), flush=True)
        del eng
        torch.cuda.empty_cache()
    summary = {"bar": BAR, "sets": len(rows), "by_form": {}, "by_sigma": {}}
    for f in ("single", "batch4", "batch16", "grid", "strict4", "strict16"):
        summary["by_form"][f] = _dist([x for r in rows for x in r["per_case"][f]])
    for sg in sorted({r["sigma"] for r in rows}):
        summary["by_sigma"]["%.1f" % sg] = _dist([x for r in rows if r["sigma"] == sg for f in r["per_case"]
                                                  if not f.startswith("strict") for x in r["per_case"][f]])
    summary["oracle32_minus_f64"] = _dist([r["oracle32_minus_f64"] for r in rows])
    worst = max(d["max"] for f, d in summary["by_form"].items())
    summary["worst"] = worst
    summary["headroom"] = 1.0 - worst / BAR
    print("[parity sweep] distribution of max |gpu - f64| per (weight set, request): " + json.dumps(summary))
    bf = summary["by_form"]
    assert max(bf[f]["max"] for f in ("single", "grid")) <= BAR, json.dumps(bf)
    # strict mode (disn_vgg_weights_t.strict_forms = 1: single-image convolution kernels in calls of any size): EVERY request
    strict_worst = max(bf["strict4"]["max"], bf["strict16"]["max"])
    summary["strict_worst"], summary["strict_headroom"] = strict_worst, 1.0 - strict_worst / BAR
    print("[parity sweep] strict mode: worst %.3g over %d requests (headroom %.0f %%)" % (
        strict_worst, bf["strict4"]["n"] + bf["strict16"]["n"], 100.0 * (1.0 - strict_worst / BAR)))
    assert strict_worst <= BAR, json.dumps({f: bf[f] for f in ("strict4", "strict16")})   # (measured: 8.0e-6, median 2.2e-6)
    batched = np.array([x for r in rows for f in ("batch4", "batch16") for x in r["per_case"][f]])
    summary["batched_fraction_above_bar"] = float((batched > BAR).mean())
    print("[parity sweep] batched forms: %d of %d requests above %.0e (%.1f %%), worst %.3g; SDF values (pred / 10): worst %.3g" % (
        int((batched > BAR).sum()), batched.size, BAR, 100.0 * (batched > BAR).mean(), batched.max(), worst / 10.0))
    od = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(od):
        with open(os.path.join(od, "sweep_%s.json" % ("full" if len(rows) == len(MS.SETS) else "default")), "w") as f:
            json.dump({"summary": summary, "rows": rows}, f, indent=1)
    for f in ("batch4", "batch16"):
        assert bf[f]["max"] <= BAR, json.dumps(bf[f])
    assert worst / 10.0 <= 2e-6          # the SDF values themselves (test/create_sdf.py:285 divides by SDF_WEIGHT = 10)
    # the GPU path is closer to the float64 truth than the reference's own fp32 arithmetic (the CPU oracle in float32)
    assert np.median([r["batch16"] for r in rows]) <= np.median([r["oracle32_minus_f64"] for r in rows])


def test_sweep_without_equalisation_fails_where_the_model_says():
    """the hardest weight set (sigma = 2, outliers x 1e4) through the SAME kernels on the RAW variables
    (SdfEngine(equalise=False)): the operand-split model (tools/split_model.py) predicts ~1e-4 from the split alone
    -- the sweep does exercise the hole VERDICT r4 named, and the equalised upload is what closes it"""
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    gold = np.load(os.path.join(GOLDEN, "stress_sweep.npz"))
    s = MS.sweep_inputs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    i = MS.SETS.index((MS.SEEDS[0], 2.0, 1e4))
    store = WeightStore(O.trained_like_weights(MS.SEEDS[0], sigma=2.0, outlier_gain=1e4))
    res = {}
    for eq in (False, True):
        eng = SdfEngine(store, equalise=eq)
        assert eng.weights.status["equalised"] is eq
        # the C ABI's guard (disn_conv_h2_gain_span): the raw upload is flagged, the equalised one is clean
        assert bool(eng.weights.status.get("gain_span_warnings")) is (not eq), eng.weights.status.get("packed_gain_span_log2")
        assert max(eng.weights.status["packed_gain_span_log2"]) <= (3.0 if eq else 64.0)
        forms, grid = _forms(eng, s, dev)
        e = _errors(forms, grid, gold, i)
        res[eq] = max(max(v) for v in e.values())
        print("[parity sweep] set %d %s: worst max |gpu - f64| %.3g" % (i, "equalised" if eq else "raw variables", res[eq]))
        del eng
        torch.cuda.empty_cache()
    assert res[True] <= BAR
    assert res[False] > 10.0 * res[True], "the raw upload was expected to lose precision on this set (measured: 1.8e-3)"
