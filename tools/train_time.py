"""Time the training step on the GPU box: forward+backward (disn_train_step) and the Adam update.
usage: python tools/train_time.py [B] [N] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disn_amd.train_sdf import Trainer  # noqa: E402
from disn_amd.weights import WeightStore  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
precision = sys.argv[4] if len(sys.argv) > 4 else "f32"   # f32 | f32_mfma | bf16
if os.environ.get("KNOB"):   # tuning build: KNOB=wgrad_fork=0
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tuning
    for kv in os.environ["KNOB"].split(","):
        k, v = kv.split("=")
        _tuning.set_knob(k, int(v))
rng = np.random.default_rng(0)
tr = Trainer(WeightStore.random_init(seed=0, mode="he"), batch_size=B, precision=precision)
dev = tr.params.device
feed = {"imgs": torch.rand((B, 137, 137, 3), device=dev),
        "sample_pc": torch.rand((B, N, 3), device=dev) * 2 - 1,
        "trans_mat": torch.tensor([[[120.0, 0, 0], [0, 120.0, 0], [68.0, 68.0, 1.0], [68.0 * 2, 68.0 * 2, 2.0]]],
                                  device=dev).repeat(B, 1, 1).contiguous(),
        "sdf": 0.05 * torch.randn((B, N, 1), device=dev)}
feed["sample_pc_rot"] = feed["sample_pc"].clone()
for _ in range(2):
    tr.step(feed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.forward_backward(feed)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(steps):
    tr.apply_gradients()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(steps):
    _, losses, _ = tr.step(feed)
torch.cuda.synchronize()
t3 = time.perf_counter()
print("B=%d N=%d: fwd+bwd %.3f ms  adam %.3f ms  step %.3f ms  (%.1f samples/s)  loss %.4g" % (
    B, N, (t1 - t0) / steps * 1e3, (t2 - t1) / steps * 1e3, (t3 - t2) / steps * 1e3,
    B * steps / (t3 - t2), float(losses["overall_loss"])))
