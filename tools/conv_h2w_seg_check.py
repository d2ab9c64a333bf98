"""Speed AND accuracy of the batched convolution's variants per VGG layer shape (disn_conv3x3_h2 tilings): error of every
variant against a float64 convolution of the same operands (unfold + matmul in float64 on the GPU), in units of the layer's
output maximum.  python tools/conv_h2w_seg_check.py [B=16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TILINGS = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 11, 6, 8, 9, 12, 13, 14, 15]
dev = torch.device("cuda:0")
if len(sys.argv) > 3:
    LAYERS = [tuple(int(v) for v in l.split("x")) for l in sys.argv[3].split(",")]
else:
  LAYERS = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28)]


def ev_ms(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def ref64(x, w, cin, cout):
    # x [B,H,W,Cin] fp32, w [9*cin, cout] rows ordered (ky, kx, ci) -> float64 output of the first 2 images
    xb = x[:2].double().permute(0, 3, 1, 2)   # (B = 1: one image)
    cols = torch.nn.functional.unfold(xb, 3, padding=1)            # [2, cin*9, HW] rows ordered (ci, ky, kx)
    w64 = w.double().view(3, 3, cin, cout).permute(2, 0, 1, 3).reshape(cin * 9, cout)
    o = torch.einsum("bkp,kn->bpn", cols, w64)
    return torch.relu(o).view(xb.shape[0], x.shape[1], x.shape[2], cout)


torch.manual_seed(0)
for cin, cout, hw in LAYERS:
    x = torch.rand((B, hw, hw, cin), device=dev) * torch.rand((1, 1, 1, cin), device=dev)
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    img = ops.pack_conv_h2(w)
    o = torch.empty((B, hw, hw, cout), device=dev)
    r = ref64(x, w, cin, cout)
    scale = float(r.abs().max())
    res = []
    for tiling in TILINGS:
        try:
            t = ev_ms(lambda: ops.conv3x3_h2(x, img, b, cout, True, tiling=tiling, out=o))
            e = (o[:r.shape[0]].double() - r)
            res.append("%d: %.1f us rms %.2e max %.2e" % (tiling, t * 1e3, float(e.pow(2).mean().sqrt()) / scale, float(e.abs().max()) / scale))
        except Exception as ex:
            res.append("%d: -" % tiling)
    print("cin %3d cout %3d hw %3d B %d:\n   %s" % (cin, cout, hw, B, "\n   ".join(res)), flush=True)
