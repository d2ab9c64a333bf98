"""Point-MLP layers at a 2048-row batch: dense_h2 (two-term f16) against the f32-input MFMA GEMM (disn_dense) and the
three-term kernel; kernel durations come from the rocprofv3 trace of this script (tools/trace_summary.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for K, N in ((64, 256), (256, 512), (512, 512), (512, 256), (1984, 512)):
    a = torch.rand((M, K), device=dev)
    w = torch.randn((K, N), device=dev) * (2.0 / K) ** 0.5
    b = torch.zeros(N, device=dev)
    pk = ops.pack_kn(w)
    for _ in range(5):
        ops.dense(a, pk, b, N, True)
    if K % 64 == 0 and K != 1984:
        img = ops.pack_dense_h2(w)
        for _ in range(5):
            ops.dense_h2(a, img, b, N, True)
    else:
        a1, a2 = a[:, :512].contiguous(), a[:, 512:].contiguous()
        img = ops.pack_dense_h2(w)
        for _ in range(5):
            ops.dense_h2(a1, img, b, N, True, a2=a2)
    torch.cuda.synchronize()
print("done")
