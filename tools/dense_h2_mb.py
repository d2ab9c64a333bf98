"""dense_h2 tile heights at a batched call's row count: duration per layer shape for dense_mb = 1 / 2 / 4 (tuning
build: DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so) and bit-equality of the results.  usage: dense_h2_mb.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
import _tuning
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def ev(fn, reps=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for K, N in ((64, 256), (256, 512), (512, 512), (512, 256), (2048, 512)):
    a = torch.rand((M, K), device=dev)
    w = torch.randn((K, N), device=dev) * (2.0 / K) ** 0.5
    b = torch.zeros(N, device=dev)
    img = ops.pack_dense_h2(w)
    ref, line = None, []
    for mb, nw, kpw in ((1, 2, 0), (2, 2, 0), (4, 2, 0), (2, 4, 2), (2, 4, 4)):
        _tuning.set_knob("dense_mb", mb)
        _tuning.set_knob("dense_nw", nw)
        _tuning.set_knob("dense_kpw", kpw)
        out = ops.dense_h2(a, img, b, N, True)
        if ref is None:
            ref = out
        same = torch.equal(out, ref)
        us = ev(lambda: ops.dense_h2(a, img, b, N, True))
        line.append("mb%d nw%d kpw%d %6.1f us %5.1f TF%s" % (mb, nw, kpw, us, 2.0 * M * K * N / us / 1e6, "" if same else " DIFFERS"))
    print("M %d K %4d N %3d: %s" % (M, K, N, " | ".join(line)), flush=True)
for k in ("dense_mb", "dense_nw", "dense_kpw"):
    _tuning.set_knob(k, 0)
