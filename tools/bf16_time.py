"""Time the bf16-compute conv / dense kernels against the fp32-MFMA ones at the training shapes.
usage: python tools/bf16_time.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda")


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


lib = _lib.lib()
for cin, cout, hw in [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
                      (256, 512, 28), (512, 512, 28), (512, 512, 14)]:
    x = torch.rand((B, hw, hw, cin), device=dev)
    w = torch.randn((3, 3, cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device=dev)
    wp = ops.pack_kn(w.reshape(9 * cin, cout))
    out = torch.empty((B, hw, hw, cout), device=dev)
    ws = torch.empty(lib.disn_conv3x3_bf16_workspace_bytes(B, hw, hw, cin, cout) + 256, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t32 = timeit(lambda: ops.conv3x3(x, wp, b, cout, True))
    t16 = timeit(lambda: lib.disn_conv3x3_bf16(x.data_ptr(), B, hw, hw, cin, w.data_ptr(), b.data_ptr(), cout, 1, NS,
                                               out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    print("conv B%d %3dx%-3d %3d->%-3d  fp32 %7.1f us %6.1f TF | bf16 (incl. pack) %7.1f us %6.1f TF  x%.2f" % (
        B, hw, hw, cin, cout, t32, fl / t32 / 1e6, t16, fl / t16 / 1e6, t32 / t16), flush=True)
M = max(B * 2048, 2048)
for k1, k2, n in [(64, 0, 256), (256, 0, 512), (512, 1472, 512), (512, 0, 512), (512, 0, 256)]:
    a1 = torch.rand((M, k1), device=dev)
    a2 = torch.rand((M, k2), device=dev) if k2 else None
    w = torch.randn((k1 + k2, n), device=dev) * 0.05
    b = torch.zeros(n, device=dev)
    wp = ops.pack_kn(w)
    t32 = timeit(lambda: ops.dense(a1, wp, b, n, True, a2))
    t16 = timeit(lambda: ops.dense_bf16(a1, w, b, True, a2, NS))
    fl = 2.0 * M * n * (k1 + k2)
    print("dense M%d K%d N%d  fp32 %7.1f us %6.1f TF | bf16 (incl. pack+alloc) %7.1f us %6.1f TF  x%.2f" % (
        M, k1 + k2, n, t32, fl / t32 / 1e6, t16, fl / t16 / 1e6, t32 / t16), flush=True)
