"""The point-MLP layer shapes of an eight-step call (16384 rows), batched form (dense_h2w.hip, rows_per_image = 2048)
against the four-k-wave tiles (dense_h2.hip, one scale for all rows): microseconds per launch from HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384


def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


KNOBS = [ks for ks in os.environ.get("KNOBS", "").split(";") if ks]
if KNOBS:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning
for k1, k2, N in ((64, 0, 256), (256, 0, 512), (512, 0, 512), (512, 1536, 512), (512, 0, 256)):
    K = k1 + k2
    a1 = torch.rand((M, k1), device=dev)
    a2 = torch.rand((M, k2), device=dev) if k2 else None
    w = torch.randn((K, N), device=dev) * (2.0 / K) ** 0.5
    b = torch.zeros(N, device=dev)
    img = ops.pack_dense_h2(w)
    us_w = t(lambda: ops.dense_h2(a1, img, b, N, True, a2=a2, rows_per_image=2048))
    us_o = t(lambda: ops.dense_h2(a1, img, b, N, True, a2=a2))
    fl = 2.0 * M * K * N
    for ks in KNOBS:
        for kv in ks.split(","):
            kk, vv = kv.split("=")
            _tuning.set_knob(kk, int(vv))
        print("   [%s] %7.1f us" % (ks, t(lambda: ops.dense_h2(a1, img, b, N, True, a2=a2, rows_per_image=2048))))
    if KNOBS:
        _tuning.set_knob("densew_m64", -1); _tuning.set_knob("densew_c128", -1)
    print("M %d K %4d N %3d: batched form %7.1f us (%.0f TFLOP/s executed f16, %.2f of 2.5 PF)   four-k-wave tiles %7.1f us   "
          "(both include the maxima passes over the inputs)" % (M, K, N, us_w, 3 * fl / us_w / 1e6, 3 * fl / us_w / 1e6 / 2500, us_o))
