"""Multi-GPU dense-grid evaluation: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all (every script pins one device,
train/train_sdf.py:88, test/create_sdf.py:83).  The path shards naturally: given one image's
encoder state every grid point is independent (1x1 convs models/sdfnet.py:71-90, per-point
gathers models/model_normalization.py:172-184).  Design:

  * the flat grid index range [0,(R+1)^3) of every image is cut into ``world`` contiguous
    slices (z-slab like: the flat order is z-major), rank r evaluates slice r;
  * every rank runs the encoder redundantly (~0.4 ms; cheaper and simpler than broadcasting
    the 110 MB feature map over xGMI) -- no collective on the data path;
  * ONE exchange step at the end.  ``exchange="all_gather"``: every rank ends with every image's full grid
    (all_gather of the padded per-rank slices; world x B x (R+1)^3 / world floats received per rank -- 543 MB at
    B = 8, R = 256).  ``exchange="all_to_all"``: rank r ends with the full grids of the images it OWNS (b % world
    == r: the ones it meshes, test/create_sdf.py:277-289) -- one all_to_all_single, each rank receives only
    (world - 1) / world of ITS images' points (68 MB per image at R = 256: 1/world of the all_gather's bytes) and
    nothing is reassembled for images it never touches.

``query_fn(image_index, k0, k1) -> 1-D tensor`` is injected so the sharding / gather logic is
testable on CPU with gloo; ``sharded_create_sdf`` binds it to the HIP engine.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of [0,total): sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    q, r = divmod(total, world)
    k0 = rank * q + min(rank, r)
    return k0, k0 + q + (1 if rank < r else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


def owned_images(n_images: int, world: int, rank: int) -> List[int]:
    """images whose full grid ends on ``rank`` with exchange="all_to_all" (round robin, as bench.py meshes them)"""
    return list(range(rank, n_images, world))


def sharded_grid(query_fn: Callable[[int, int, int], torch.Tensor], n_images: int, total: int,
                 device, group=None, exchange: str = "all_gather"):
    """Evaluate ``n_images`` grids of ``total`` points, sharded over the process group.
    exchange="all_gather": returns the full [n_images, total] result on every rank.
    exchange="all_to_all": returns ([n_owned, total], owned image indices) -- the full grids of this rank's images."""
    if exchange not in ("all_gather", "all_to_all"):
        raise ValueError("exchange must be 'all_gather' or 'all_to_all'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    k0, k1 = shard_range(total, world, rank)
    pad = (total + world - 1) // world           # the collectives need equal sizes
    mine = torch.zeros((n_images, pad), dtype=torch.float32, device=device)
    for b in range(n_images):
        if k1 > k0:
            mine[b, :k1 - k0] = query_fn(b, k0, k1)
    own = owned_images(n_images, world, rank)
    if world == 1 and not dist.is_initialized():
        return mine[:, :total] if exchange == "all_gather" else (mine[:, :total], own)
    # (a one-rank process group still goes through the collective: the same RCCL call sequence as
    #  the 8-rank job, which is how the path is exercised on a single-GPU box)
    if exchange == "all_to_all":
        # destination d gets this rank's slice of the images d owns: rows reordered owner-major, one
        # all_to_all_single with per-destination split sizes; from source s arrives s's slice of MY images
        order = [b for d in range(world) for b in owned_images(n_images, world, d)]
        send = mine[order].contiguous()
        in_split = [len(owned_images(n_images, world, d)) * pad for d in range(world)]
        recv = torch.empty((world, len(own), pad), dtype=torch.float32, device=device)
        dist.all_to_all_single(recv.view(-1), send.view(-1), output_split_sizes=[len(own) * pad] * world,
                               input_split_sizes=in_split, group=group)
        out = torch.empty((len(own), total), dtype=torch.float32, device=device)
        for r in range(world):
            a, b_ = shard_range(total, world, r)
            out[:, a:b_] = recv[r, :, :b_ - a]
        return out, own
    gathered = torch.empty((world, n_images, pad), dtype=torch.float32, device=device)
    if _supports_flat(group):      # RCCL: one flat collective into the contiguous buffer
        dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
    else:                          # gloo (CPU tests)
        dist.all_gather(list(gathered.unbind(0)), mine, group=group)
    out = torch.empty((n_images, total), dtype=torch.float32, device=device)
    for r in range(world):
        a, b_ = shard_range(total, world, r)
        out[:, a:b_] = gathered[r, :, :b_ - a]
    return out


def _supports_flat(group) -> bool:
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:  # pragma: no cover
        return False


def sharded_create_sdf(engine, imgs, trans_mats, sdf_params, sdf_res: int, sdf_weight: float = 10.0,
                       group=None, exchange: str = "all_gather"):
    """BASELINE config 4: every rank encodes the batch, evaluates its slice of every image's
    grid and the slices are exchanged.  Returns [B,(res+1)^3] = pred_sdf / SDF_WEIGHT on every rank
    (exchange="all_gather"), or ([n_owned,(res+1)^3], owned image indices) with exchange="all_to_all"."""
    import numpy as np
    from .create_sdf import dense_grid_sdf
    enc = engine.encode(imgs)
    B = enc.embedding.shape[0]
    total = (sdf_res + 1) ** 3
    sp = np.asarray(sdf_params, dtype=np.float64).reshape(B, 6)

    def query_fn(b, k0, k1):
        return dense_grid_sdf(engine, enc, b, trans_mats, sp[b], sdf_res, sdf_weight, k_range=(k0, k1))

    return sharded_grid(query_fn, B, total, engine.device, group, exchange)


# ---------------------------------------------------------------------------
# data-parallel training step (BASELINE config 5): gradient exchange
# ---------------------------------------------------------------------------
def shard_batch(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """[b0, b1) of the global batch this rank trains on (every rank must get the same count: the
    mean over the global batch is the mean of the per-rank means only then)."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by the world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradientReducer:
    """Sum all-reduce of the flat gradient buffer in two buckets, overlapped with the backward:

      head  = grads[head_offset:]   fc6..fc8 + both point MLPs (129 M floats, 96 % of the bytes),
              final while the convolution backward (about half of the step) is still running;
              reduced on a side stream as soon as ``head_ready`` fires
      tail  = grads[:head_offset]   the 13 convolutions (14.7 M floats), reduced after the step

    The 1/world factor is applied by the optimizer kernel (grad_scale), not here.  xGMI is
    point-to-point, a ring all-reduce is per-link bound: one 516 MB collective amortises the ring
    latency, and it is hidden under ~6 ms of MFMA-bound backward.  Device-agnostic (the CPU tests run
    it on gloo with ``side_stream=None``)."""

    def __init__(self, head_offset: int, group=None, use_side_stream: bool = True, force: bool = False):
        self.head_offset = int(head_offset)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force: run the collectives even in a one-rank group (exercises streams/events on one GPU)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.side = torch.cuda.Stream() if (use_side_stream and torch.cuda.is_available()) else None
        self._work = None

    def start_head(self, grads: torch.Tensor, head_ready=None) -> None:
        """call right after the step was enqueued; head_ready: event recorded when the head is final"""
        if not self.active:
            return
        head = grads[self.head_offset:]
        if self.side is not None:
            if head_ready is not None:
                self.side.wait_event(head_ready)
            else:
                self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self._work = dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work = dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, grads: torch.Tensor) -> None:
        """reduce the tail and join: afterwards `grads` holds the sum over ranks on the current stream"""
        if not self.active:
            return
        if self.head_offset > 0:
            dist.all_reduce(grads[:self.head_offset], op=dist.ReduceOp.SUM, group=self.group)
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
