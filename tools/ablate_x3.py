"""Where does the three-term conv kernel spend its time?  Builds libdisn_amd variants whose
gemm_bf16_mfma.hip is compiled with -DDISN_ABL=<mask> (results are WRONG by construction; timing only).
  python tools/ablate_x3.py build      # here: writes disn_amd/csrc/build/libdisn_abl<mask>.so
  python tools/ablate_x3.py time B     # on the GPU box: times the 12 conv layers with every variant"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "disn_amd", "csrc")
MASKS = [0, 1, 2, 4, 8, 16, 1 | 16, 2 | 4, 1 | 2 | 4 | 16, 31]

if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from disn_amd.csrc import build as B
    B.build()
    objs = {}
    for src, flags in B.SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in B.HEADERS]
        tag = B._digest([sp] + hdrs, " ".join(B.COMMON + flags))
        objs[src] = os.path.join(CSRC, "build", "%s.%s.o" % (src, tag))
    for m in MASKS[1:]:
        o = os.path.join(CSRC, "build", "abl%d.o" % m)
        subprocess.check_call([B.HIPCC] + B.COMMON + ["-DDISN_TUNING", "-DDISN_ABL=%d" % m, "-c", os.path.join(CSRC, "gemm_bf16_mfma.hip"), "-o", o])
        link = [o if s == "gemm_bf16_mfma.hip" else p for s, p in objs.items()]
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(CSRC, "build", "libdisn_abl%d.so" % m)] + link)
        print("built", m, flush=True)
elif sys.argv[1] == "time":
    Bn = sys.argv[2] if len(sys.argv) > 2 else "1"
    for m in MASKS:
        env = dict(os.environ)
        if m:
            env["DISN_AMD_LIB"] = os.path.join(CSRC, "build", "libdisn_abl%d.so" % m)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", Bn], env=env, capture_output=True, text=True)
        print("ABL=%-2d %s" % (m, r.stdout.strip() or r.stderr[-300:]), flush=True)
else:
    import torch
    sys.path.insert(0, ROOT)
    from disn_amd import ops
    Bn = int(sys.argv[2])
    dev = torch.device("cuda")
    out = []
    for cin, cout, hw in [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
                          (256, 512, 28), (512, 512, 28), (512, 512, 14)]:
        x = torch.rand((Bn, hw, hw, cin), device=dev)
        w = torch.randn((9 * cin, cout), device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        wp = ops.pack_kn_x3(w)
        o = torch.empty((Bn, hw, hw, cout), device=dev)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        f = lambda: ops.conv3x3_x3(x, wp, b, cout, True, ws, o)
        f(); f(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record(); e.synchronize()
        out.append("%5.1f" % (s.elapsed_time(e) / 20 * 1e3))
    # the point MLP at 65536 rows (all dense layers on the three-term kernel, 128x128 tiles): whole query
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore
    import numpy as np
    eng = SdfEngine(WeightStore.random_init(0, mode="he"))
    img = torch.rand((1, 137, 137, 3), device=dev)
    tm = torch.tensor(np.array([[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [68, 68, 2.0]]], dtype=np.float32), device=dev)
    enc = eng.encode(img)
    p = torch.rand((1, 65536, 3), device=dev) * 2 - 1
    f = lambda: eng.query(enc, p, tm)
    f(); f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        f()
    e.record(); e.synchronize()
    out.append("| query65536 %6.1f us" % (s.elapsed_time(e) / 10 * 1e3))
    print(" ".join(out))
