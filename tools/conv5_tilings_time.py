import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from disn_amd import ops
dev = torch.device("cuda:0")
def ev_ms(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B in (16, 8, 4):
    cin, cout, hw = 512, 512, 14
    x = torch.rand((B, hw, hw, cin), device=dev)
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    img = ops.pack_conv_h2(w)
    o = torch.empty((B, hw, hw, cout), device=dev)
    res = []; ref = None
    for tiling in (0, 1, 2, 3, 4, 10, 0):
        try:
            t = ev_ms(lambda: ops.conv3x3_h2(x, img, b, cout, True, tiling=tiling, out=o))
            if ref is None: ref = o.clone(); same = ""
            else: same = "=" if torch.equal(o, ref) else "~"
            res.append("%d:%.1f%s" % (tiling, t * 1e3, same))
        except Exception as e:
            res.append("%d:-" % tiling)
    print("conv5 B %d: %s" % (B, "  ".join(res)), flush=True)
