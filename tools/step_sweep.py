"""Time the full step (disn_encode_query) for values of one tuning knob.  Needs the tuning build:
DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so python tools/step_sweep.py <knob> <v1> <v2> ..."""
import os, subprocess, sys
if sys.argv[1] == "--one":
    import time
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    import _tuning
    _tuning.set_knob(sys.argv[2], int(sys.argv[3]))
    eng = SdfEngine(WeightStore.random_init(0))
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).cuda()
    pts = torch.rand((1, 2048, 3), device="cuda") * 2 - 1
    tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                        [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]], device="cuda")
    for _ in range(10): eng.encode_query(img, pts, tm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): eng.encode_query(img, pts, tm)
    torch.cuda.synchronize(); print("%.4f ms/step" % ((time.perf_counter() - t0) / 200 * 1e3))
else:
    knob = sys.argv[1]
    for v in sys.argv[2:] + sys.argv[2:3]:      # the first value again at the end: drift check
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", knob, v], capture_output=True, text=True)
        print("%s=%s : %s" % (knob, v, out.stdout.strip() or out.stderr[-300:]), flush=True)
