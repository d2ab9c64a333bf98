"""Time disn_fc (split-K GEMV stream + reduce) alone: python tools/fc_time.py [B=16]  -- fc6 / fc7 / the global fold shapes,
and the error of the result against float64"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for K, N in ((25088, 4096), (4096, 4096), (4096, 1024), (1024, 512)):
    w = (torch.randn((K, N), generator=g) * (2.0 / K) ** 0.5).to(dev)
    x = torch.rand((B, K), generator=g).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = ops.fc(x, w, b, True)
    ref = torch.relu(x.double() @ w.double() + b.double())
    err = float((out.double() - ref).abs().max())
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.fc(x, w, b, True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = min(ts)
    print("B=%d K=%d N=%d: %.1f us, %.2f TB/s of weights, max |gpu - f64| %.2e (max |ref| %.2f)" % (
        B, K, N, t * 1e3, K * N * 4 / t / 1e9, err, float(ref.abs().max())))
    del w, x
