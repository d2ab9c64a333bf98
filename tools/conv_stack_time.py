"""Conv stack only (resize + 13 convolutions, no fc head), repeated: per-launch durations under rocprofv3 show what
the layers cost when nothing else (fc6's 411 MB stream) passes through the caches between two forwards.
usage: conv_stack_time.py [B]   (B images per call, default 1)
KNOBS="name=v,name=v;name=v;..." (tuning build): one timing per ';'-separated knob set, taps compared bit for bit with
the first run (no knobs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
from disn_amd.engine import SdfEngine
from disn_amd.weights import WeightStore
eng = SdfEngine(WeightStore.random_init(0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = torch.from_numpy(np.random.default_rng(0).random((B, 137, 137, 3), dtype=np.float32)).cuda()
r = ops.ConvStackRun(eng.weights.vgg, img, want_pool5=False)


def timed(tag):
    for _ in range(5): r.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): r.run()
    torch.cuda.synchronize()
    print("conv stack, %d image(s) per call %s: %.1f us per call" % (B, tag, (time.perf_counter() - t0) / 100 * 1e6), flush=True)


timed("")
if os.environ.get("KNOBS"):
    import _tuning
    ref = [t.clone() for t in r.taps]
    for ks in os.environ["KNOBS"].split(";"):
        for kv in ks.split(","):
            k, v = kv.split("=")
            _tuning.set_knob(k, int(v))
        timed("[" + ks + "]")
        print("   taps equal the default path's: %s" % all(torch.equal(a, b) for a, b in zip(ref, r.taps)), flush=True)
