"""Workload for a kernel trace of ONE 65536-point query chunk (the unit of the dense grid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
eng = SdfEngine(WeightStore.random_init(0, mode="he"))
dev = eng.device
img = torch.rand((1, 137, 137, 3), device=dev)
tm = torch.tensor(np.array([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [68, 68, 2.0]]], dtype=np.float32), device=dev)
enc = eng.encode(img)
p = torch.rand((1, 65536, 3), device=dev) * 2 - 1
for _ in range(4):
    eng.query(enc, p, tm)
torch.cuda.synchronize()
