"""Sweep tile shape x split-K for every GEMM-shaped launch of the path on the GPU and print the
best plan per shape (input for the planner table in gemm_mfma.hip).  Timing: events on the
launch stream, includes the split-K reduce when S > 1."""
import json, os, sys
import torch
import _tuning  # tuning build: DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd import ops

torch.cuda.set_device(0)
dev = torch.device("cuda")


def timeit(fn, reps=8):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


SVALS = [1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 36]


def wlist(M, N, bm, bn, ksteps):
    """workgroup counts to try: data-parallel, a few split-K multiples, stream-K 256*G"""
    tiles = -(-M // bm) * (N // bn)
    out = {tiles}
    for s in (2, 3, 4, 8):
        if tiles * s <= 2048 and ksteps // s >= 2:
            out.add(tiles * s)
    for w in (256, 512, 768, 1024):
        if w <= tiles * ksteps:
            out.add(w)
    return sorted(out)
results = {}
conv = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
        (256, 512, 28), (512, 512, 28), (512, 512, 14)]
B = int(os.environ.get("SWEEP_B", "1"))
for cin, cout, hw in conv:
    x = torch.rand((B, hw, hw, cin), device=dev)
    w = ops.pack_kn(torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5)
    b = torch.zeros(cout, device=dev)
    ksteps = 1 if cin == 3 else 9 * cin // 32
    flop = 2.0 * B * hw * hw * cout * 9 * cin
    rows = []
    for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        if cout % bn:
            continue
        for s in wlist(B * hw * hw, cout, bm, bn, ksteps):
            _tuning.gemm_force(bm, bn, s)
            us = timeit(lambda: ops.conv3x3(x, w, b, cout, True))
            rows.append((us, bm, bn, s))
    _tuning.gemm_force()
    auto = timeit(lambda: ops.conv3x3(x, w, b, cout, True))
    rows.sort()
    key = "conv B%d %dx%d %d->%d" % (B, hw, hw, cin, cout)
    results[key] = {"auto_us": auto, "best": rows[:4], "tflops_best": flop / rows[0][0] / 1e6}
    print("%-28s auto %7.1f us | best %s  -> %.1f TF" % (key, auto, ["%.1f@%d,%d,%d" % r for r in rows[:4]],
                                                         flop / rows[0][0] / 1e6), flush=True)

dense = [(64, 0, 256), (256, 0, 512), (512, 0, 512), (512, 1472, 512), (512, 0, 256)]
for M in (2048, 16384, 65536):
    for k1, k2, n in dense:
        a1 = torch.rand((M, k1), device=dev)
        a2 = torch.rand((M, k2), device=dev) if k2 else None
        w = ops.pack_kn(torch.randn((k1 + k2, n), device=dev) * 0.05)
        b = torch.zeros(n, device=dev)
        ksteps = (k1 + k2) // 32
        flop = 2.0 * M * n * (k1 + k2)
        rows = []
        for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
            for s in wlist(M, n, bm, bn, ksteps):
                _tuning.gemm_force(bm, bn, s)
                us = timeit(lambda: ops.dense(a1, w, b, n, True, a2))
                rows.append((us, bm, bn, s))
        _tuning.gemm_force()
        auto = timeit(lambda: ops.dense(a1, w, b, n, True, a2))
        rows.sort()
        key = "dense M%d K%d N%d" % (M, k1 + k2, n)
        results[key] = {"auto_us": auto, "best": rows[:4], "tflops_best": flop / rows[0][0] / 1e6}
        print("%-28s auto %7.1f us | best %s  -> %.1f TF" % (key, auto, ["%.1f@%d,%d,%d" % r for r in rows[:4]],
                                                             flop / rows[0][0] / 1e6), flush=True)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "sweep_gemm_B%d.json" % B)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(results, open(out, "w"), indent=1)
