"""Wall-clock of a 257^3 dense-grid evaluation, sequential vs chunk-pipelined."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import create_sdf as cs
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
torch.cuda.set_device(0)
eng = SdfEngine(WeightStore.random_init(0))
img = torch.rand((1, 137, 137, 3), device="cuda")
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
enc = eng.encode(img)
for rnd in range(2):
    for pipe in (False, True):
        eng.query_grid(enc, 0, tm, [-1] * 3 + [1] * 3, 256, 0, 300000, pipelined=pipe)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eng.query_grid(enc, 0, tm, [-1] * 3 + [1] * 3, 256, pipelined=pipe)
        torch.cuda.synchronize(); print("pipelined=%s : %.4f s" % (pipe, time.perf_counter() - t0), flush=True)
