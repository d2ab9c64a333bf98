"""Is a convolution layer's time set by the workgroups' own pace or by the chip's power / throughput?  The 28 x 28,
512 -> 512 layer in the batched form is ONE round of 16 B workgroups (B images x 4 patches x 4 n-tiles): time it at
B = 8 .. 16 -- if the time grows with B below 256 workgroups (all of them resident at once), the CUs slow each other
down (clock / power), and work removed from a launch is time removed even without shortening any workgroup.
usage: python tools/conv_occupancy_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

dev = torch.device("cuda:0")


def ev_us(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for hw, cin, cout in ((28, 512, 512), (56, 256, 256), (14, 512, 512)):
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    img = ops.pack_conv_h2(w)
    bias = torch.zeros(cout, device=dev)
    for B in (4, 8, 10, 12, 14, 16, 20, 24, 32):
        x = torch.rand((B, hw, hw, cin), device=dev)
        o = torch.empty((B, hw, hw, cout), device=dev)
        t = ev_us(lambda: ops.conv3x3_h2(x, img, bias, cout, True, out=o))
        print("hw %3d %3d -> %3d, B %2d: %7.1f us  (%.2f us per image; includes the maxima pass over the input)" % (hw, cin, cout, B, t, t / B), flush=True)
