"""Throughput of disn_encode_query when B independent (image, 2048 points) steps are submitted as ONE batched call,
optionally with S such batches in flight (StepPipeline contexts).  usage: batch_try.py B [S]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import StepPipeline
from disn_amd.weights import WeightStore
B = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if os.environ.get("KNOB"):      # tuning build only: KNOB=name=value[,name=value]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning
    for kv in os.environ["KNOB"].split(","):
        k, v = kv.split("=")
        _tuning.set_knob(k, int(v))
if os.environ.get("ORDER") == "bench":      # bisecting bench.py vs this tool: the same preamble as bench.py
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
pipe = StepPipeline(WeightStore.random_init(0), in_flight=S, batch=B)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
if os.environ.get("SRC") == "bench":
    rng = np.random.default_rng(1000)
    img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
    pts = torch.from_numpy((rng.random((1, 2048, 3), dtype=np.float32) * 2 - 1).astype(np.float32)).cuda()
tm1 = [[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
       [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]
tm = torch.tensor([tm1], device="cuda")
K = int(os.environ.get('K', '360'))
pipe.run([(img, pts, tm)] * (3 * S * B)); torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter(); pipe.run([(img, pts, tm)] * K); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / K * 1e3)
print("%s batch %d, %d batches in flight: %s ms per (image + 2048 points)" % (os.environ.get("KNOB", ""), B, S, " ".join("%.4f" % r for r in res)), flush=True)
