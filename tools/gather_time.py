"""project_gather_taps_kernel alone: B images x 2048 points, all five taps (the unit entry's 256-thread launch), HIP events;
KNOBS="gather_l16=0;gather_l16=1" (tuning build) compares the two load schedules."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hw, ch = [224, 112, 56, 28, 14], [64, 128, 256, 512, 512]
taps = [torch.rand((B, h, h, c), device=dev) for h, c in zip(hw, ch)]
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * B, device=dev)
pts = torch.rand((B, 2048, 3), device=dev) - 0.5
out = torch.empty((B, 2048, 1472), device=dev)


def t(tag):
    for _ in range(3): ops.gather_taps(taps, tm, pts, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.gather_taps(taps, tm, pts, out)
    e.record(); e.synchronize()
    print("gather from taps, %d x 2048 points %s: %.1f us" % (B, tag, s.elapsed_time(e) / 30 * 1e3), flush=True)
    return out.clone()


ref = t("")
if os.environ.get("KNOBS"):
    import _tuning
    for ks in os.environ["KNOBS"].split(";"):
        for kv in ks.split(","):
            k, v = kv.split("=")
            _tuning.set_knob(k, int(v))
        o = t("[" + ks + "]")
        print("   bits equal the default's: %s" % torch.equal(o, ref))
