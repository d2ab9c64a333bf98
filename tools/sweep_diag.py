"""Where the error of a batched call comes from on a sweep weight set (tests/golden/stress_sweep.npz): the encoder form
(16 images in one call: conv_h2w + the matrix-pipe fc head | every image alone: conv_h2 + row fc) crossed with the
point-MLP form (fused small-set kernels | layer by layer from the materialised map), per request, against the float64
oracle.  python tools/sweep_diag.py [set index ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_sweep as MS   # noqa: E402
from disn_amd import ops   # noqa: E402
from disn_amd.engine import Encoded, SdfEngine   # noqa: E402
from disn_amd.weights import WeightStore   # noqa: E402
from oracle import disn_oracle as O   # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "stress_sweep.npz"))
s = MS.sweep_inputs()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for i in [int(a) for a in sys.argv[1:]] or [4]:
    seed, sigma, outlier = MS.SETS[i]
    eng = SdfEngine(WeightStore(O.trained_like_weights(seed, sigma=sigma, outlier_gain=outlier)))
    imgs, tms = dev(s["imgs"]), dev(s["trans_mat"])
    enc_b = eng.encode(torch.cat([imgs, imgs]))                     # 16 images: the batched forms
    enc_b = Encoded(enc_b.resized[:8], [t[:8].contiguous() for t in enc_b.taps], enc_b.embedding[:8].contiguous(), None)
    singles = [eng.encode(imgs[b:b + 1]) for b in range(8)]
    enc_s = Encoded(torch.cat([e.resized for e in singles]), [torch.cat([e.taps[k] for e in singles]) for k in range(5)],
                    torch.cat([e.embedding for e in singles]), None)
    print("set %d (seed %d sigma %.1f outliers %.0e): |emb_batched - emb_single| max %.3g of %.3g; taps rel diff %s" % (
        i, seed, sigma, outlier, float((enc_b.embedding - enc_s.embedding).abs().max()), float(enc_s.embedding.abs().max()),
        ["%.2g" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(enc_b.taps, enc_s.taps)]))
    e64 = gold["emb64_%02d" % i]
    print("   image 0 embedding vs f64: batched %.3g  single %.3g (|emb| max %.3g)" % (
        float(np.abs(enc_b.embedding[0].cpu().numpy() - e64).max()), float(np.abs(enc_s.embedding[0].cpu().numpy() - e64).max()),
        float(np.abs(e64).max())))
    for j in (0, 1):
        pts = dev(s["pts"][:, j])
        ref = gold["pred64_%02d" % i][:, j]
        rows = {}
        for en, enc in (("enc batched", enc_b), ("enc single ", enc_s)):
            f = ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, tms, pts).cpu().numpy()
            enc.featmap = None
            u = eng.query(enc, pts, tms, fold=False, fused=False).cpu().numpy()
            rows[en + " + fused MLP   "] = np.abs(f - ref).max(1)
            rows[en + " + layer by layer"] = np.abs(u - ref).max(1)
        # mixed: batched taps + single embedding, and the reverse (which half of the encoder matters)
        mix1 = ops.query_taps_fused(eng.weights.mlp, enc_b.taps, enc_s.embedding, tms, pts).cpu().numpy()
        mix2 = ops.query_taps_fused(eng.weights.mlp, enc_s.taps, enc_b.embedding, tms, pts).cpu().numpy()
        rows["taps batched, emb single + fused"] = np.abs(mix1 - ref).max(1)
        rows["taps single, emb batched + fused"] = np.abs(mix2 - ref).max(1)
        for k, v in rows.items():
            print("   pts %d  %-36s worst %.3g   per image %s" % (j, k, v.max(), " ".join("%.1e" % x for x in v)))
    del eng
    torch.cuda.empty_cache()
