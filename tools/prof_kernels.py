"""Small, bounded workload for rocprofv3 passes (kernel trace or one PMC counter at a time):
PROF_STEPS x (encode + 2048-point query), one standalone gather at 2048 and 262144 points,
one 65536-point query through the layer-by-layer chain and through the fused kernels, one single-call step
(disn_encode_query: no feature map, gather from the taps), and LAST one convolution stack on PROF_BATCH images
(default 16: the batch bench.py's main line submits per call)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore

torch.cuda.set_device(0)
eng = SdfEngine(WeightStore.random_init(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
for _ in range(int(os.environ.get("PROF_STEPS", "3"))):
    enc = eng.encode(img)
    out = eng.query(enc, pts, tm)
# round 4: the fused small-set point MLP on the 16 x 2048 points of one call (here, between the steps and the gathers:
# the "first 13 convolutions" / "last dispatch of ..." selections of tools/pmc_traffic.py keep their meaning)
imgs16 = torch.from_numpy(rng.random((16, 137, 137, 3), dtype=np.float32)).cuda()
pts16 = torch.rand((16, 2048, 3), device="cuda") * 2 - 1
enc16 = eng.encode(imgs16)
for _ in range(2):
    ops.query_taps_fused(eng.weights.mlp, enc16.taps, enc16.embedding, tm.expand(16, -1, -1).contiguous(), pts16)
torch.cuda.synchronize()
del enc16, imgs16
for n in (2048, 262144):
    p = torch.rand((1, n, 3), device="cuda") * 2 - 1
    xy = ops.project(p, tm)
    for _ in range(3):
        f = ops.gather(enc.featmap, xy)
p = torch.rand((1, 65536, 3), device="cuda") * 2 - 1
for _ in range(2):
    eng.query(enc, p, tm, fused=False)      # layer-by-layer chunk: 8 GEMMs + gather_fold_kernel
for _ in range(2):
    eng.query(enc, p, tm, fused=True)       # mlp_fused_kernel<global>, <local>
eng.encode_query(img, pts, tm)
torch.cuda.synchronize()
PB = int(os.environ.get("PROF_BATCH", "16"))
if PB > 0:
    imgs = torch.from_numpy(rng.random((PB, 137, 137, 3), dtype=np.float32)).cuda()
    ops.ConvStackRun(eng.weights.vgg, imgs, want_pool5=False).run()
torch.cuda.synchronize()
print("done")
