import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from disn_amd import ops
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
from oracle import disn_oracle as O
eng = SdfEngine(WeightStore.random_init(2, mode="he"))
rng = np.random.default_rng(0)
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B, N in ((1, 8192), (1, 16384), (1, 65536), (2, 8192), (3, 16384)):
    imgs = torch.from_numpy(rng.random((B, 137, 137, 3), dtype=np.float32)).cuda()
    pts = torch.from_numpy(rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)).cuda()
    tms = torch.from_numpy(np.repeat(O.DEMO_TRANS_MAT, B, axis=0)).cuda()
    a = eng.encode_query(imgs, pts, tms)[1]
    def two():
        enc = eng.encode(imgs)
        return ops.query_taps_fused(eng.weights.mlp, enc.taps, enc.embedding, tms, pts)
    b = two()
    print("B=%d N=%d: encode_query %.3f ms, encode + query_taps_fused %.3f ms, max diff %.2e" % (
        B, N, t(lambda: eng.encode_query(imgs, pts, tms)), t(two), float((a - b).abs().max())))
