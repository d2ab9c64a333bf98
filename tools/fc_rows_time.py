"""One-row fc layers from the transposed matrix (disn_fc_t -> gemv_rows_kernel<1, R, U>): time per (R, U) and bit equality.
    DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so python tools/fc_rows_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from disn_amd import ops
import _tuning
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
NAMES = {0: "default", 1: "<1,8>", 2: "<2,8>", 3: "<1,16>", 4: "<4,4>", 5: "<2,4>", 6: "<1,4>"}
for K, N, nm in ((4096, 4096, "fc7"), (4096, 1000, "fc8"), (1000, 512, "global bias fold")):
    wt = (torch.randn((N, K), generator=g) * (2.0 / K) ** 0.5).to(dev)
    x = torch.rand((1, K), generator=g).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)   # flush the Infinity Cache between runs: cold weights, as in a step
    ref, res = None, []
    for cfg in range(7):
        _tuning.set_knob("gemv_rows_cfg", cfg)
        out = ops.fc_t(x, wt, b, True)
        ref = out.clone() if ref is None else ref
        ts = []
        for _ in range(7):
            big.fill_(1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.fc_t(x, wt, b, True); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append("%s %.1f us%s" % (NAMES[cfg], sorted(ts)[len(ts) // 2], "" if torch.equal(out, ref) else " (BITS DIFFER)"))
    print("%s K %d N %d (%.1f MB): %s" % (nm, K, N, K * N * 4 / 1e6, "  ".join(res)), flush=True)
