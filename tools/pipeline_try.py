"""disn_amd.engine.StepPipeline throughput for S steps in flight (one process per value: the stream -> hardware
queue assignment is made at creation).  usage: GPU_MAX_HW_QUEUES=8 python tools/pipeline_try.py S [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.engine import StepPipeline
from disn_amd.weights import WeightStore
S = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 240
if os.environ.get("KNOB"):      # tuning build only: KNOB=name=value
    import _tuning
    k, v = os.environ["KNOB"].split("=")
    _tuning.set_knob(k, int(v))
pipe = StepPipeline(WeightStore.random_init(0), in_flight=S)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
CH = {"1": True, "0": False}.get(os.environ.get("CHAINED", ""), None)
pipe.run([(img, pts, tm)] * (4 * S), chained=CH); torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter(); pipe.run([(img, pts, tm)] * K, chained=CH); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / K * 1e3)
print("chained=%s %s queues %s in_flight %d: %s ms per step" % (os.environ.get("CHAINED", "default"), os.environ.get("KNOB", ""), os.environ.get("GPU_MAX_HW_QUEUES", "default"), S, " ".join("%.4f" % r for r in res)), flush=True)
