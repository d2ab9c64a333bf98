"""Per-layer time of the 12 MFMA convolutions of one VGG-16 forward at B = 1 (HIP events on the launch stream):
the three-term bf16 implicit GEMM (+ its split-K reduce) against the two-term f16 halo kernel (conv_h2.hip).
usage: python tools/conv_h2_time.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
LAYERS = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 256, 56),
          (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14), (512, 512, 14)]


def ev_ms(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot3 = tot2 = flop = 0.0
lib = ops.lib()
for cin, cout, hw in LAYERS:
    x = torch.rand((B, hw, hw, cin), device=dev)
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    w3 = ops.pack_kn_x3(w)
    ws3 = torch.empty(max(lib.disn_conv3x3_x3_workspace_bytes(B, hw, hw, cin, cout), 256), dtype=torch.uint8, device=dev)
    o = torch.empty((B, hw, hw, cout), device=dev)
    t3 = ev_ms(lambda: ops.conv3x3_x3(x, w3, b, cout, True, ws3, o))
    img = ops.pack_conv_h2(w)
    t2 = {}
    for tiling in (0, 1, 2, 3, 4):
        t2[tiling] = ev_ms(lambda: ops.conv3x3_h2(x, img, b, cout, True, tiling=tiling, out=o))
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    tot3 += t3; tot2 += t2[0]; flop += fl
    print("cin %3d cout %3d hw %3d: x3 %7.1f us (%5.1f TF/s)   h2 %7.1f us (%5.1f TF/s)  [amax pass included]   tilings 1-4: %s"
          % (cin, cout, hw, t3 * 1e3, fl / t3 / 1e9, t2[0] * 1e3, fl / t2[0] / 1e9,
             " ".join("%.1f" % (t2[k] * 1e3) for k in (1, 2, 3, 4))), flush=True)
print("total: x3 %.1f us (%.1f TF/s), h2 %.1f us (%.1f TF/s)" % (tot3 * 1e3, flop / tot3 / 1e9, tot2 * 1e3, flop / tot2 / 1e9))
