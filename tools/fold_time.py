"""time one folded 65536-point query (the dense grid's chunk)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
from disn_amd import ops
eng = SdfEngine(WeightStore.random_init(0, mode="he"))
dev = eng.device
img = torch.rand((1, 137, 137, 3), device=dev)
tm = torch.tensor(np.array([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], dtype=np.float32), device=dev)
enc = eng.encode(img)
p = ops.grid_points([-1, -1, -1, 1, 1, 1], 256, 8000000, 8000000 + 65536, dev)[None].contiguous()
f = lambda: eng.query(enc, p, tm, fold=True)
f(); f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f()
e.record(); e.synchronize()
print("folded chunk of grid points: %.1f us" % (s.elapsed_time(e) / 20 * 1e3))
