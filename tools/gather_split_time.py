"""Time disn_gather_taps_split alone on B x 2048 points (random taps of the VGG shapes): python tools/gather_split_time.py [B=16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
ch = (64, 128, 256, 512, 512)
taps = [torch.rand((B, 224 >> k, 224 >> k, ch[k]), generator=g).to(dev) for k in range(5)]
pts = (torch.rand((B, 2048, 3), generator=g) - 0.5).to(dev)
tm = torch.tensor([[60.0, 0, 0], [0, 60.0, 0], [0, 0, 0.0], [68.5, 68.5, 1.0]]).repeat(B, 1, 1).contiguous().to(dev)
amax = torch.ones(B, device=dev)
out = ops.gather_taps_split(taps, tm, pts, amax)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gather_taps_split(taps, tm, pts, amax)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print("B=%d gather_taps_split %.1f us (min of 5 x 20), checksum %d" % (B, min(ts) * 1e3, int(out.to(torch.int64).sum().item())))
