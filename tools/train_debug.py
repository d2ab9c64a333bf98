"""Per-variable gradient error of disn_train_step against the float64 autograd oracle (GPU box).
usage: python tools/train_debug.py [B] [N] [float32|float64]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import disn_oracle as O  # noqa: E402
from oracle import train_oracle as T  # noqa: E402
from disn_amd.train_sdf import Trainer, VARIABLE_ORDER  # noqa: E402
from disn_amd.weights import WeightStore  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "float32") else np.float64
weights = O.init_weights(3, "he")
feed = O.synth_inputs(seed=21, batch=B, n_points=N)
rng = np.random.default_rng(22)
feed["sdf"] = (0.05 * rng.standard_normal((B, N, 1))).astype(np.float32)
L, grads, pred = T.loss_and_grads(feed, weights, dt)
tr = Trainer(WeightStore(weights), batch_size=B)
d = {k: torch.from_numpy(np.ascontiguousarray(feed[k], np.float32)).cuda()
     for k in ("imgs", "trans_mat", "sample_pc", "sample_pc_rot", "sdf")}
dpred, dl = tr.forward_backward(d)
torch.cuda.synchronize()
got = tr.flat.to_arrays(tr.grads)
print("losses", L, dl.cpu().numpy())
print("%-52s %10s %10s %10s %8s" % ("variable", "max|ref|", "max err", "rel", "frac>1e-3"))
for name in VARIABLE_ORDER:
    ref = grads[name].astype(np.float64)
    g = got[name].astype(np.float64)
    sc = max(np.abs(ref).max(), 1e-30)
    err = np.abs(g - ref)
    print("%-52s %10.3e %10.3e %10.3e %8.5f" % (name, sc, err.max(), err.max() / sc, (err > 1e-3 * sc).mean()))
