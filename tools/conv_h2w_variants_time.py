"""Per-layer time of the batched 3x3 convolutions (conv_h2w.hip) at B images per call through every forced variant
(disn_conv3x3_h2 tiling 5..9 = variants 1..5; 0 = the launcher's choice): python tools/conv_h2w_variants_time.py [B=16]
Variants 1..3 have one k-wave, 4..5 two: only variants with the launcher's number of k-waves give the launcher's bits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
LAYERS = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28)]


def ev_ms(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for cin, cout, hw in LAYERS:
    x = torch.rand((B, hw, hw, cin), device=dev)
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    img = ops.pack_conv_h2(w)
    o = torch.empty((B, hw, hw, cout), device=dev)
    ref = None
    res = []
    for tiling in (0, 5, 6, 7, 8, 9, 12, 13, 14, 15):
        try:
            t = ev_ms(lambda: ops.conv3x3_h2(x, img, b, cout, True, tiling=tiling, out=o))
            same = ""
            if tiling == 0:
                ref = o.clone()
            else:
                same = "=" if torch.equal(o, ref) else "~"
            res.append("%d:%.1f%s" % (tiling, t * 1e3, same))
        except Exception as e:
            res.append("%d:-" % tiling)
    print("cin %3d cout %3d hw %3d B %d: %s   (us; '=' the default's bits)" % (cin, cout, hw, B, "  ".join(res)), flush=True)
