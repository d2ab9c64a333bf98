"""Workload for the MFMA-utilisation PMC pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE):
the convolution stack of ONE image (conv_h2.hip) and of EIGHT / SIXTEEN images (conv_h2w.hip; sixteen = what bench.py's roofline times), the
point-MLP layer shapes of an eight-step call (dense_h2w.hip, 16384 rows) and of one step (dense_h2.hip, 2048 rows).
Every variant runs twice after a warm-up; tools/pmc_mfma.py reads the LAST dispatch of each kernel / grid."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
eng = SdfEngine(WeightStore.random_init(0, mode="he"))
rng = np.random.default_rng(0)
for B in (1, 8, 16):
    imgs = torch.from_numpy(rng.random((B, 137, 137, 3), dtype=np.float32)).cuda()
    r = ops.ConvStackRun(eng.weights.vgg, imgs, want_pool5=False)
    for _ in range(3):
        r.run()
    torch.cuda.synchronize()
for M, rows in ((2048, 0), (16384, 2048), (32768, 2048)):
    for k1, k2, N in ((256, 0, 512), (512, 0, 512), (512, 1536, 512), (512, 0, 256)):
        K = k1 + k2
        a1 = torch.rand((M, k1), device=dev)
        a2 = torch.rand((M, k2), device=dev) if k2 else None
        w = torch.randn((K, N), device=dev) * (2.0 / K) ** 0.5
        img = ops.pack_dense_h2(w)
        b = torch.zeros(N, device=dev)
        for _ in range(3):
            ops.dense_h2(a1, img, b, N, True, a2=a2, rows_per_image=rows)
        torch.cuda.synchronize()
# round 4: the fused small-set point MLP of a 16-step call (mlp_fused_kernel<local, FEAT>, <global>) and the fused
# kernels of a 65536-point query (folded map)
B = 16
imgs = torch.from_numpy(rng.random((B, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.from_numpy(rng.uniform(-1, 1, (B, 2048, 3)).astype(np.float32)).cuda()
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * B, device="cuda")
enc = eng.encode(imgs)
for _ in range(3):
    ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, tm, pts)
torch.cuda.synchronize()
enc1 = eng.encode(imgs[:1])
p64 = torch.from_numpy(rng.uniform(-1, 1, (1, 65536, 3)).astype(np.float32)).cuda()
for _ in range(3):
    eng.query(enc1, p64, tm[:1])
torch.cuda.synchronize()
print("done")
