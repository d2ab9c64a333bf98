"""Time the full step (disn_encode_query) under the overlap variants (DISN_OVERLAP bitmask,
DISN_RESIZE_BG_BLOCKS)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
torch.cuda.set_device(0)
eng = SdfEngine(WeightStore.random_init(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
def run(steps=60):
    for _ in range(5): eng.encode_query(img, pts, tm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.encode_query(img, pts, tm)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rounds in range(2):
    for ov, bg in (("0", "256"), ("1", "64"), ("1", "256"), ("1", "1024"), ("1", "0"), ("2", "256"), ("3", "256"), ("3", "64")):
        os.environ["DISN_OVERLAP"] = ov; os.environ["DISN_RESIZE_BG_BLOCKS"] = bg
        print("overlap=%s bg_blocks=%-5s : %.4f ms/step" % (ov, bg, run()), flush=True)
