"""Phase timeline of the conv_h2 kernel from in-kernel clock stamps (tuning build only):
python -m disn_amd.csrc.build --tuning; DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so python tools/conv_h2_stamps.py [B]
B > 1: B copies of the image in one launch -- the multi-round launches of a batched call (two-workgroups-per-CU
variants); KNOB=name=value[,name=value] sets tuning knobs first (e.g. conv_occ=1: the base tilings)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops, _lib

h = _lib.lib()
h.disn_tuning_set_ptr.restype, h.disn_tuning_set_ptr.argtypes = C.c_int, [C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if os.environ.get("KNOB"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning
    for kv in os.environ["KNOB"].split(","):
        k, v = kv.split("=")
        _tuning.set_knob(k, int(v))
LAYERS = [(64, 64, 224), (128, 128, 112), (256, 256, 56), (512, 512, 28), (512, 512, 14)]
if os.environ.get("LAYERS"):   # LAYERS=256x256x56,512x512x28 (cin x cout x hw)
    LAYERS = [tuple(int(v) for v in l.split("x")) for l in os.environ["LAYERS"].split(",")]
TILING = int(os.environ.get("TILING", "0"))   # disn_conv3x3_h2 tiling (12: the segmented batched variant -- STAMP_CK=32: a stamp per segment)
NAMES = ["setup", "prologue(load+split+barrier)"] + ["chunk%d" % i for i in range(8)] + ["last chunk", "k-reduce", "epilogue"]
for cin, cout, hw in LAYERS:
    x = torch.rand((1, hw, hw, cin), device=dev).repeat(B, 1, 1, 1)
    w = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    img = ops.pack_conv_h2(w)
    o = torch.empty((B, hw, hw, cout), device=dev)
    stamps = torch.zeros((65536, 16), dtype=torch.int64, device=dev)   # one row per workgroup
    for _ in range(3):
        ops.conv3x3_h2(x, img, b, cout, True, out=o, tiling=TILING)
    torch.cuda.synchronize()
    h.disn_tuning_set_ptr(0, stamps.data_ptr())
    ops.conv3x3_h2(x, img, b, cout, True, out=o, tiling=TILING)
    torch.cuda.synchronize()
    h.disn_tuning_set_ptr(0, None)
    s = stamps.cpu().numpy()
    s = s[s[:, 1] != 0]
    wall = (s[:, 0] - s[:, 0].min()) / 100.0          # wall_clock64: 100 MHz -> us
    nc = cin // int(os.environ.get("STAMP_CK", "64"))   # channels per chunk of the kernel that ran (conv_h2w: 16 or 32)
    idx = [1, 2, 3] + [4 + c for c in range(min(nc - 1, 8))] + [12, 13, 14]
    names = ["setup", "prologue"] + ["chunk%d" % c for c in range(min(nc - 1, 8))] + ["last chunk", "k-reduce", "epilogue"]
    print("B %d cin %d cout %d hw %d: %d workgroups; start skew: median %.2f us, max %.2f us (the launch's rounds show as "
          "steps of the start times)" % (B, cin, cout, hw, len(s), np.median(wall), wall.max()))
    if B > 1:
        q = np.percentile(wall, [10, 25, 50, 75, 90])
        print("   start time percentiles 10/25/50/75/90: %s us" % " ".join("%.1f" % v for v in q))
    tot = s[:, 14] - s[:, 1]
    for a, b_, n in zip(idx[:-1], idx[1:], names):
        d = s[:, b_] - s[:, a]
        print("   %-12s median %8.0f cycles   (min %8.0f max %8.0f)" % (n, np.median(d), d.min(), d.max()))
    print("   total        median %8.0f cycles   (min %8.0f max %8.0f)  [wave 0 of each workgroup]" % (np.median(tot), tot.min(), tot.max()))
