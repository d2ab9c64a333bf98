"""disn_conv3x3_backward at the VGG layer shapes of an 8-sample training step: time of the block (bias gradient, weight
gradient, data gradient) per precision mode, HIP events.  KNOBS="tn_interleave=0;tn_interleave=1" (tuning build) compares
schedules of the weight-gradient GEMM; results are compared with the first run (max |diff| / max |ref|).
usage: conv_bwd_time.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(224, 64, 64), (112, 64, 128), (112, 128, 128), (56, 128, 256), (56, 256, 256), (28, 256, 512), (28, 512, 512),
          (14, 512, 512)]
g = torch.Generator(device=dev); g.manual_seed(0)
data = {}
for H, Cin, Cout in SHAPES:
    x = torch.randn((B, H, H, Cin), device=dev, generator=g)
    w = torch.randn((3, 3, Cin, Cout), device=dev, generator=g) / (9 * Cin) ** 0.5
    dy = torch.randn((B, H, H, Cout), device=dev, generator=g)
    data[(H, Cin, Cout)] = (x, w, dy)


def run(tag, mode, ref):
    out = {}
    for key, (x, w, dy) in data.items():
        need_dx = os.environ.get("NO_DX", "0") != "1"
        for _ in range(2): r = ops.conv3x3_backward(x, w, None, dy.clone(), wd=0.0, need_dx=need_dx, compute_bf16=mode)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dys = [dy.clone() for _ in range(5)]
        s.record()
        for d in dys: r = ops.conv3x3_backward(x, w, None, d, wd=0.0, need_dx=need_dx, compute_bf16=mode)
        e.record(); e.synchronize()
        dw = r[1]
        out[key] = dw.clone()
        err = ""
        if ref is not None:
            err = "  dw vs first run: %.2e of max" % (float((dw - ref[key]).abs().max()) / float(ref[key].abs().max()))
        print("mode %d %-22s %3d x %3d, %3d -> %3d: %7.1f us%s" % (mode, tag, B, key[0], key[1], key[2],
                                                                   s.elapsed_time(e) / 5 * 1e3, err), flush=True)
    return out


modes = [int(m) for m in os.environ.get("MODES", "2,1").split(",")]
refs = {m: run("", m, None) for m in modes}
if os.environ.get("KNOBS"):
    import _tuning
    for ks in os.environ["KNOBS"].split(";"):
        for kv in ks.split(","):
            k, v = kv.split("=")
            _tuning.set_knob(k, int(v))
        for m in modes:
            run("[" + ks + "]", m, refs[m])
