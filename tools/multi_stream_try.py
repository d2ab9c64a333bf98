"""Throughput of the full step (disn_encode_query) with S independent steps in flight: S engines (own workspace,
own auxiliary stream), each driven by its own host thread on its own HIP stream.  usage: multi_stream_try.py S..."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore

store = WeightStore.random_init(0)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
SHARE = os.environ.get("SHARE_WEIGHTS", "0") == "1"
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    engs = [SdfEngine(store)]
    engs += [SdfEngine(None, weights=engs[0].weights) if SHARE else SdfEngine(store) for _ in range(S - 1)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    K = 240 // S

    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                engs[i].encode_query(img, pts, tm)

    for i in range(S):
        work(i, 5)
    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=(i, K)) for i in range(S)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("share=%d S=%d: %.4f ms per step (%d steps), %.3g points/s" % (SHARE, S, dt / (K * S) * 1e3, K * S, K * S * 2048 / dt), flush=True)
    del engs
