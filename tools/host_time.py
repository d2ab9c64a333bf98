"""Host time of one disn_encode_query call (enqueue only): sync, call, stop the clock, sync."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
from disn_amd import ops
eng = SdfEngine(WeightStore.random_init(0))
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
for _ in range(10): eng.encode_query(img, pts, tm)
torch.cuda.synchronize()
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.encode_query(img, pts, tm); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host time of engine.encode_query: median %.1f us (min %.1f)" % (np.median(ts) * 1e6, min(ts) * 1e6))
# the C call alone, buffers preallocated
from disn_amd._lib import lib
import ctypes as C
B, N = 1, 2048
ws = eng._workspace("encq", lib().disn_encode_query_workspace_bytes(B, N))
resized = torch.empty((B, 224, 224, 3), device="cuda")
taps = [torch.empty((B, hw, hw, ch), device="cuda") for hw, ch in ops.TAP_SHAPES]
emb = torch.empty((B, 1024), device="cuda"); sdf = torch.empty((B, N), device="cuda")
tp = (C.c_void_p * 5)(*[t.data_ptr() for t in taps])
st = torch.cuda.current_stream().cuda_stream
def call():
    return lib().disn_encode_query(eng._ctx, C.byref(eng.weights.vgg), C.byref(eng.weights.mlp), img.data_ptr(), tm.data_ptr(),
                                   pts.data_ptr(), pts.data_ptr(), B, N, resized.data_ptr(), C.byref(tp), emb.data_ptr(), None,
                                   sdf.data_ptr(), ws.data_ptr(), ws.numel(), st)
for _ in range(5): assert call() == 0
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host time of the C call alone:      median %.1f us (min %.1f)" % (np.median(ts) * 1e6, min(ts) * 1e6))
