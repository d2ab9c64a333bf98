"""Is the cfg2 step host-bound?  wall time of enqueueing K steps vs until the GPU has finished them."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
eng = SdfEngine(WeightStore.random_init(0, mode="he"))
dev = eng.device
img = torch.rand((1, 137, 137, 3), device=dev)
pts = torch.rand((1, 2048, 3), device=dev) * 2 - 1
tm = torch.tensor(np.array([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [68, 68, 2.0]]], dtype=np.float32), device=dev)
for _ in range(20): eng.encode_query(img, pts, tm)
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): eng.encode_query(img, pts, tm)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step, until done %.1f us/step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
