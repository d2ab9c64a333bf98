"""error of the encoder's stages by form on sweep set 4: taps (batched | single conv kernels) and the fc head
(matrix-pipe | VALU split-K) against float64 intermediates computed on the EQUALISED variables.

    python tools/sweep_diag2.py make     CPU, ~1 minute: writes tools/_diag_set04.npz (21 MB, git-ignored; it travels to
                                         the GPU box with the snapshot)
    python tools/sweep_diag2.py          GPU: the comparison (profiles/r05d_sweep_diag2.txt)
"""
import os
import sys

import numpy as np

ROOT0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SET = int(os.environ.get("DIAG_SET", "4"))   # sweep weight set (DIAG_SET=24 python tools/sweep_diag2.py [make])
if len(sys.argv) > 1 and sys.argv[1] == "make":
    sys.path.insert(0, ROOT0)
    sys.path.insert(0, os.path.join(ROOT0, "tests", "golden"))
    import make_golden_sweep as MS0
    from disn_amd.weights import WeightStore as WS0
    from oracle import disn_oracle as O0
    seed0, sigma0, outlier0 = MS0.SETS[SET]
    Weq = WS0(O0.trained_like_weights(seed0, sigma=sigma0, outlier_gain=outlier0)).equalised()[0].arrays
    _, emb0, _, eps0 = O0.encode(MS0.sweep_inputs()["imgs"], Weq, np.float64)
    taps0 = {nm: np.asarray(eps0["vgg_16/%s/%s" % (nm[:5], nm)], np.float64) for nm in O0.TAP_NAMES}
    np.savez_compressed(os.path.join(ROOT0, "tools", "_diag_set%02d.npz" % SET), emb64=np.asarray(emb0, np.float64),
                        pool5_64=O0.max_pool_2x2(taps0["conv5_3"]), **{"tap64_" + k: v[:, ::3, ::3, :] for k, v in taps0.items()})
    print("wrote tools/_diag_set%02d.npz" % SET)
    sys.exit(0)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_sweep as MS   # noqa: E402
from disn_amd import ops   # noqa: E402
from disn_amd.engine import SdfEngine   # noqa: E402
from disn_amd.weights import WeightStore   # noqa: E402
from oracle import disn_oracle as O   # noqa: E402

g = np.load(os.path.join(ROOT, "tools", "_diag_set%02d.npz" % SET))
s = MS.sweep_inputs()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
seed, sigma, outlier = MS.SETS[SET]
store = WeightStore(O.trained_like_weights(seed, sigma=sigma, outlier_gain=outlier))
eq = store.equalised()[0]
eng = SdfEngine(store)
imgs = dev(s["imgs"])
enc_b = eng.encode(torch.cat([imgs, imgs]))
singles = [eng.encode(imgs[b:b + 1]) for b in range(8)]
for k, nm in enumerate(O.TAP_NAMES):
    ref = g["tap64_" + nm]
    tb = enc_b.taps[k][:8, ::3, ::3, :].cpu().numpy().astype(np.float64)
    ts = torch.cat([e.taps[k] for e in singles])[:, ::3, ::3, :].cpu().numpy().astype(np.float64)
    sc = np.abs(ref).reshape(8, -1).max(1)
    eb = np.abs(tb - ref).reshape(8, -1).max(1) / sc
    es = np.abs(ts - ref).reshape(8, -1).max(1) / sc
    rb = np.sqrt(((tb - ref) ** 2).reshape(8, -1).mean(1)) / sc
    rs = np.sqrt(((ts - ref) ** 2).reshape(8, -1).mean(1)) / sc
    print("%s: max err / tap max  batched %s | single %s ;  rms/max batched %.2e single %.2e" % (
        nm, " ".join("%.1e" % v for v in eb), " ".join("%.1e" % v for v in es), rb.mean(), rs.mean()))
# the fc head alone, from the float64 pool5 rounded to fp32
W = eq.arrays
fcw = [dev(W["vgg_16/%s/weights" % n].reshape(-1, W["vgg_16/%s/weights" % n].shape[3])) for n in ("fc6", "fc7", "fc8")]
fcb = [dev(W["vgg_16/%s/biases" % n]) for n in ("fc6", "fc7", "fc8")]
def head(x):
    h = ops.fc(x, fcw[0], fcb[0], True)
    h = ops.fc(h, fcw[1], fcb[1], True)
    return ops.fc(h, fcw[2], fcb[2], False)
p5 = dev(g["pool5_64"].reshape(8, -1))
e64 = g["emb64"]
sc = np.abs(e64).max(1)
m = head(torch.cat([p5, p5]))[:8].cpu().numpy()
v = torch.cat([head(p5[b:b + 1]) for b in range(8)]).cpu().numpy()
print("fc head alone (exact pool5): err / |emb| max   16 rows (matrix pipe) %s | 1 row (VALU) %s" % (
    " ".join("%.1e" % x for x in np.abs(m - e64).max(1) / sc), " ".join("%.1e" % x for x in np.abs(v - e64).max(1) / sc)))
# the GPU's own pool5 of each conv form through each fc form
for cn, tap in (("batched conv", enc_b.taps[4][:8].contiguous()), ("single conv ", torch.cat([e.taps[4] for e in singles]))):
    p = ops.maxpool2x2(tap).reshape(8, -1).contiguous()
    m = head(torch.cat([p, p]))[:8].cpu().numpy()
    v = torch.cat([head(p[b:b + 1]) for b in range(8)]).cpu().numpy()
    print("%s: emb err / max   fc 16 rows %s | fc 1 row %s" % (cn, " ".join("%.1e" % x for x in np.abs(m - e64).max(1) / sc),
                                                                " ".join("%.1e" % x for x in np.abs(v - e64).max(1) / sc)))
print("engine embeddings: batched %s | single %s" % (
    " ".join("%.1e" % x for x in np.abs(enc_b.embedding[:8].cpu().numpy() - e64).max(1) / sc),
    " ".join("%.1e" % x for x in np.abs(torch.cat([e.embedding for e in singles]).cpu().numpy() - e64).max(1) / sc)))
