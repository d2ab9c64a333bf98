"""Time the full step (disn_encode_query) on one stream (tuning knob overlap = 0) and on two (default).
Needs the tuning build (tools/_tuning.py).
(History: build r01c also had the tap up-samples on the auxiliary stream, bit 0 of the then bitmask,
at several grid throttles -- 0.86-0.98 ms against 0.797 ms without; that path was removed.)"""
import os, subprocess, sys
if len(sys.argv) > 1:
    import time
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    import _tuning
    _tuning.set_knob("overlap", int(sys.argv[1]))
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
    for ov in ("0", "1", "0", "1"):      # the switch is read once per process
        out = subprocess.run([sys.executable, os.path.abspath(__file__), ov], capture_output=True, text=True)
        print("overlap=%s : %s" % (ov, out.stdout.strip() or out.stderr[-200:]), flush=True)
