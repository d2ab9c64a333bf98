"""A/B of the gather from the taps (tuning build): the one-wave-per-point kernel (round 6, default) against the
thread-per-float4 kernel (KNOB gather_l16=2), fp32 rows and split rows, bit equality and time at B x 2048 points.
    DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so python tools/gather_wave_ab.py [B=16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
import _tuning
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
hw, ch = [224, 112, 56, 28, 14], [64, 128, 256, 512, 512]
torch.manual_seed(1)
taps = [torch.randn((B, h, h, c), device=dev).relu_() * (k + 1) for k, (h, c) in enumerate(zip(hw, ch))]
tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * B, device=dev)
pts = torch.rand((B, N, 3), device=dev) - 0.5
pts[:, :8] *= 40.0     # a few points projecting outside the image / onto its border
amax = torch.stack([torch.stack([t[b].abs().max() for t in taps]).max() for b in range(B)]).contiguous()


def ev(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


res = {}
for knob, tag in ((0, "one wave per point"), (2, "thread per float4")):
    _tuning.set_knob("gather_l16", knob)
    a = ops.gather_taps(taps, tm, pts)
    s = ops.gather_taps_split(taps, tm, pts, amax)
    res[knob] = (a.clone(), s.clone())
    bytes_alg = B * N * 29440
    t1, t2 = ev(lambda: ops.gather_taps(taps, tm, pts, a)), ev(lambda: ops.gather_taps_split(taps, tm, pts, amax))
    print("%-20s %d x %d points: fp32 rows %.1f us (%.2f TB/s, %.3f of 8)   split rows %.1f us (%.2f TB/s, %.3f of 8)" % (
        tag, B, N, t1, bytes_alg / t1 / 1e6, bytes_alg / t1 / 8e6, t2, bytes_alg / t2 / 1e6, bytes_alg / t2 / 8e6), flush=True)
print("bits equal: fp32 rows %s, split rows %s" % (torch.equal(res[0][0], res[2][0]), torch.equal(res[0][1], res[2][1])))
