"""The data-parallel call sequence of the training step on ONE GPU: a one-rank 'nccl' (= RCCL) group
with the gradient exchange forced on (head bucket on the side stream gated by the library's
head-ready event, tail bucket after the step, Adam with grad_scale 1/world), compared with the
exchange-free trainer.  Run by tests/test_gpu_train.py; prints DDP1_OK.
env: RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=<port>"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from disn_amd.parallel import GradientReducer  # noqa: E402
from disn_amd.train_sdf import Trainer  # noqa: E402
from disn_amd.weights import WeightStore  # noqa: E402

B, N = 2, 512
g = torch.Generator(device="cuda").manual_seed(0)
feed = {"imgs": torch.rand((B, 137, 137, 3), device="cuda", generator=g),
        "sample_pc": torch.rand((B, N, 3), device="cuda", generator=g) - 0.5,
        "trans_mat": torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                                    [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * B,
                                  device="cuda"),
        "sdf": 0.05 * torch.randn((B, N, 1), device="cuda", generator=g)}
feed["sample_pc_rot"] = feed["sample_pc"].clone()
a = Trainer(WeightStore.random_init(1, mode="he"), batch_size=B)
b = Trainer(WeightStore.random_init(1, mode="he"), batch_size=B)
b.reducer = GradientReducer(b.reducer.head_offset, force=True)
assert b.reducer.active and b.reducer.side is not None
h = b.reducer.head_offset
a.step(feed)
b.step(feed)
torch.cuda.synchronize()
# step 1: the fc + MLP gradients are bit-reproducible, so the exchanged run must match exactly
assert torch.equal(a.params[h:], b.params[h:]), float((a.params[h:] - b.params[h:]).abs().max())
for _ in range(2):  # later steps see conv weights that differ in the last bits (atomics)
    a.step(feed)
    b.step(feed)
torch.cuda.synchronize()
# Adam moves a weight by ~lr = 1e-4 per step in the direction of sign(g): where g ~ 0 the last-bit noise
# flips the sign, so two runs may differ by up to 2*lr per step there
assert torch.allclose(a.params, b.params, rtol=0, atol=1e-3), float((a.params - b.params).abs().max())
assert float((a.params - b.params).abs().mean()) < 2e-6
a.close()
b.close()
dist.barrier()
dist.destroy_process_group()
print("DDP1_OK")
