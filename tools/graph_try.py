"""Does a hipGraph of the cfg2 step (captured through torch.cuda.graph) beat stream launches?"""
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
for _ in range(5): ref = eng.encode_query(img, pts, tm)[1]
torch.cuda.synchronize()

def timeit(fn, K=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6

print("streams : enqueue %.1f us, done %.1f us per step" % timeit(lambda: eng.encode_query(img, pts, tm)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): eng.encode_query(img, pts, tm)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    enc, sdf = eng.encode_query(img, pts, tm)
g.replay(); torch.cuda.synchronize()
print("graph result equal:", torch.equal(sdf, ref))
print("graph   : enqueue %.1f us, done %.1f us per step" % timeit(g.replay))
