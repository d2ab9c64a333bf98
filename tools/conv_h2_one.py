"""A few launches of one conv_h2 layer (for PMC passes): python tools/conv_h2_one.py cin cout hw [tiling] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops
cin, cout, hw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tiling = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
x = torch.rand((1, hw, hw, cin), device=dev)
w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
b = torch.zeros(cout, device=dev)
img = ops.pack_conv_h2(w)
o = torch.empty((1, hw, hw, cout), device=dev)
for _ in range(reps):
    ops.conv3x3_h2(x, img, b, cout, True, tiling=tiling, out=o)
torch.cuda.synchronize()
print("done")
