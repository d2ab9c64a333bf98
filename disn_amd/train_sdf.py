"""Host side of the training step -- mirrors train/train_sdf.py of the reference.

  get_learning_rate   train/train_sdf.py:153-161 (staircase exponential decay, floor 1e-6)
  Trainer.step        one sess.run([train_op, step, lr, loss, pred_sdf, ...]) of :371-387
  Trainer.save/restore  tf.train.Saver over every global variable whose name lacks 'lr'/'batch'
                      (:285-286): the model variables AND the Adam slots '<var>/Adam', '<var>/Adam_1',
                      'beta1_power', 'beta2_power', as a TF V2 bundle (disn_amd/tf_checkpoint.py)

All device work is the HIP library (disn_train_step / disn_adam_update); data-parallel training is one
process per GPU, each on its own shard of the batch, with ONE sum all-reduce over the flat gradient
buffer (RCCL over xGMI; gloo on CPU tensors is not supported: the path has no CPU fallback) and the
1/world scale folded into the Adam kernel.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .parallel import GradientReducer
from .weights import WeightStore, variable_shapes

LOSS_NAMES = ("accuracy", "sdf_loss_realvalue", "sdf_loss", "regularization", "overall_loss")
VARIABLE_ORDER = tuple(variable_shapes())  # == the order of disn_param_layout
PRECISIONS = {"f32_mfma": 0, "bf16": 1, "f32": 2}   # -> compute_bf16 of disn_train_step
HEAD_FIRST_VAR = 26  # vgg_16/fc6/weights: everything from here on is final before the conv backward


def get_learning_rate(step: int, batch_size: int, base_lr: float = 1e-4, decay_step: int = 200000,
                      decay_rate: float = 0.9) -> float:
    """tf.train.exponential_decay(base, step*batch, decay_step, decay_rate, staircase=True) floored
    at 1e-6 (train/train_sdf.py:153-161; flags :36-40)"""
    return max(base_lr * decay_rate ** ((step * batch_size) // decay_step), 1e-6)


class FlatParams:
    """the 56 variables of the graph in ONE device buffer (include/disn_amd.h, disn_param_layout)"""

    def __init__(self, device):
        self.layout = ops.param_layout()
        self.total = int(self.layout.total)
        self.device = device
        self.shapes = variable_shapes()
        self.index = {n: i for i, n in enumerate(VARIABLE_ORDER)}

    def zeros(self) -> torch.Tensor:
        return torch.zeros(self.total, dtype=torch.float32, device=self.device)

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        i = self.index[name]
        o, c = int(self.layout.offset[i]), int(self.layout.count[i])
        return buf[o:o + c].view(self.shapes[name])

    def from_store(self, store: WeightStore) -> torch.Tensor:
        host = np.zeros(self.total, np.float32)
        for n, i in self.index.items():
            o, c = int(self.layout.offset[i]), int(self.layout.count[i])
            host[o:o + c] = np.asarray(store[n], np.float32).reshape(-1)
        return torch.from_numpy(host).to(self.device)

    def to_arrays(self, buf: torch.Tensor, suffix: str = "") -> Dict[str, np.ndarray]:
        host = buf.detach().cpu().numpy()
        out = {}
        for n, i in self.index.items():
            o, c = int(self.layout.offset[i]), int(self.layout.count[i])
            out[n + suffix] = host[o:o + c].reshape(self.shapes[n]).copy()
        return out


def adam_step_from_checkpoint(arrays: Dict[str, np.ndarray], beta2: float, current: int = 0,
                              underflow_step: Optional[int] = None) -> int:
    """Adam's timestep t of a restored bundle, from its slot variable beta2_power = beta2^(t+1) -- what a TF restore
    gives the optimizer back (train/train_sdf.py:285-286 saves every global variable except those named 'lr' / 'batch',
    so beta1_power / beta2_power ARE in a reference bundle; TF keeps multiplying them and uses
    lr_t = lr sqrt(1 - beta2_power) / (1 - beta1_power)).  While the power is a usable float32 (t < ~8e4; it goes
    denormal near 8.7e4 steps and reaches 0 near 1.03e5) t is recovered exactly; a power of exactly 0 means the bias
    correction is saturated (lr_t == lr in TF as well): ``underflow_step`` is then returned -- an explicit caller choice;
    None = max(current, 200000), any t at which beta2^t is 0 in the arithmetic of apply_gradients.  No beta2_power in
    the bundle: ``current``."""
    b2 = arrays.get("beta2_power")
    if b2 is not None and 0.0 < float(b2) < 1.0:
        return max(int(round(math.log(float(b2)) / math.log(beta2))) - 1, 0)
    if b2 is not None and float(b2) == 0.0:
        return int(underflow_step) if underflow_step is not None else max(int(current), 200000)
    return int(current)


def schedule_step_from_checkpoint(arrays: Dict[str, np.ndarray]) -> int:
    """the step the learning-rate schedule resumes at.  The reference's Saver leaves `batch` (its global step,
    train/train_sdf.py:232) OUT of every bundle (:285-286: names containing 'lr' or 'batch' are filtered), so a
    restored reference run restarts the schedule at step 0 with lr = base_lr -- and so does this function for a
    reference-written bundle.  A bundle written by Trainer.save(include_step=True) carries `batch` (an extension of
    this implementation, off by default) and resumes there."""
    gs = arrays.get("batch")
    if gs is not None and np.asarray(gs).size == 1:
        return max(int(np.asarray(gs).reshape(())), 0)
    return 0


class Trainer:
    def __init__(self, store: WeightStore, device="cuda:0", batch_size: int = 20, base_lr: float = 1e-4,
                 decay_step: int = 200000, decay_rate: float = 0.9, wd: float = 1e-5,
                 sdf_weight: float = 10.0, mask_weight: float = 4.0, beta1: float = 0.5,
                 beta2: float = 0.999, eps: float = 1e-8, process_group=None, compute_bf16: bool = False,
                 precision: Optional[str] = None):
        # precision of the conv / MLP GEMMs (everything else is fp32 in every mode):
        #   "f32"       fp32-accurate, the reference's precision: forward and data-gradient GEMMs as a
        #               three-term bf16 split on the bf16 MFMA pipes (same error as the f32-input MFMA,
        #               faster), weight gradients on the f32-input MFMA            [default]
        #   "f32_mfma"  every product on the f32-input MFMA
        #   "bf16"      mixed precision: bf16 multiply, fp32 accumulate / master weights / optimizer
        if precision is None:
            precision = "bf16" if compute_bf16 else "f32"
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (tuple(PRECISIONS),))
        self.precision = precision
        self.compute_bf16 = PRECISIONS[precision]
        self.flat = FlatParams(torch.device(device))
        self.params = self.flat.from_store(store)
        self.grads = self.flat.zeros()
        self.m = self.flat.zeros()
        self.v = self.flat.zeros()
        self.step_count = 0  # the reference's `batch` variable (global step): drives the learning-rate schedule
        self.adam_t = 0      # Adam's timestep (TF keeps it as beta1_power / beta2_power): drives the bias correction
        self.batch_size = batch_size  # GLOBAL batch (all ranks), as the LR schedule counts samples
        self.base_lr, self.decay_step, self.decay_rate = base_lr, decay_step, decay_rate
        self.wd, self.sdf_weight, self.mask_weight = wd, sdf_weight, mask_weight
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self._ws: Optional[torch.Tensor] = None
        # every stream / event this object owns lives on params.device, and every launch below runs under
        # torch.cuda.device(params.device): ops._stream() is the CURRENT device's current stream
        with torch.cuda.device(self.params.device):
            self.ctx = ops.ctx_create()  # auxiliary stream for the HBM-bound side work of the step
            # gradient exchange: fc + MLP bucket under the convolution backward, conv bucket at the end
            self.reducer = GradientReducer(int(self.flat.layout.offset[HEAD_FIRST_VAR]), process_group)
            self.head_ready = torch.cuda.Event()
            self.head_ready.record()  # creates the hipEvent_t handed to the library

    def close(self) -> None:
        if self.ctx:
            torch.cuda.synchronize(self.params.device)
            with torch.cuda.device(self.params.device):
                ops.ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # ---- one step ---------------------------------------------------------------------
    def forward_backward(self, feed: Dict[str, torch.Tensor]):
        """gradients of THIS rank's shard into self.grads; -> (pred, losses tensor[5])"""
        B, N = feed["sample_pc"].shape[:2]
        need = ops.lib().disn_train_workspace_bytes(B, N)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.params.device)
        with torch.cuda.device(self.params.device):
            out = ops.train_step(self.params, self.grads, feed["imgs"], feed["trans_mat"], feed["sample_pc"],
                                 feed["sample_pc_rot"], feed["sdf"], self.wd, self.sdf_weight,
                                 self.mask_weight, ws=self._ws, ctx=self.ctx, head_ready=self.head_ready,
                                 compute_bf16=self.compute_bf16)
            self.reducer.start_head(self.grads, self.head_ready)
        return out

    def learning_rate(self) -> float:
        return get_learning_rate(self.step_count, self.batch_size, self.base_lr, self.decay_step,
                                 self.decay_rate)

    def apply_gradients(self) -> float:
        lr = self.learning_rate()
        t = self.adam_t + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        with torch.cuda.device(self.params.device):
            self.reducer.finish(self.grads)
            ops.adam_update(self.params, self.grads, self.m, self.v, lr_t, self.beta1, self.beta2, self.eps,
                            1.0 / self.world)
        self.adam_t = t
        self.step_count += 1
        return lr

    def step(self, feed: Dict[str, torch.Tensor]):
        """-> (pred [B,N] device, losses dict name -> device scalar, lr)"""
        pred, losses = self.forward_backward(feed)
        lr = self.apply_gradients()
        return pred, {n: losses[i] for i, n in enumerate(LOSS_NAMES)}, lr

    # ---- checkpoints --------------------------------------------------------------------
    def state_arrays(self, include_step: bool = False) -> Dict[str, np.ndarray]:
        """what the reference's Saver writes (train/train_sdf.py:285-286): every variable, the Adam slots and the two
        beta powers -- NOT `batch` / the learning rate.  include_step: also `batch` (int32, as TF creates it), an
        extension that lets restore() resume the learning-rate schedule (see schedule_step_from_checkpoint)."""
        out = self.flat.to_arrays(self.params)
        out.update(self.flat.to_arrays(self.m, "/Adam"))
        out.update(self.flat.to_arrays(self.v, "/Adam_1"))
        out["beta1_power"] = np.asarray(self.beta1 ** (self.adam_t + 1), np.float32)
        out["beta2_power"] = np.asarray(self.beta2 ** (self.adam_t + 1), np.float32)
        if include_step:
            out["batch"] = np.asarray(self.step_count, np.int32)
        return out

    def weight_store(self) -> WeightStore:
        return WeightStore(self.flat.to_arrays(self.params))

    def save(self, prefix: str, include_step: bool = False, max_to_keep: int = 5) -> None:
        """variables + Adam slots + beta powers as a TF Saver-V2 bundle, and the `checkpoint` state file next to it
        (what saver.save writes, train/train_sdf.py:285-286,322-328), so that restore_latest / get_checkpoint_state
        find it.  The state file keeps the last ``max_to_keep`` prefixes in all_model_checkpoint_paths, as
        tf.train.Saver does (older bundles stay on disk here; TF would delete them)."""
        from . import tf_checkpoint as tfc
        tfc.save_checkpoint(prefix, self.state_arrays(include_step))
        d = os.path.dirname(os.path.abspath(prefix))
        base = os.path.basename(prefix)
        paths = [p for p in tfc.all_checkpoint_paths(d) if p != base] + [base]
        tfc.write_checkpoint_state(d, base, paths[-max(1, int(max_to_keep)):])

    def restore(self, prefix: str, underflow_step: Optional[int] = None) -> int:
        """prefix + exact-shape match, as load_model (train/train_sdf.py:190-219); -> #restored.  Adam's timestep comes
        back from beta2_power (adam_step_from_checkpoint; ``underflow_step`` for a power that underflowed to 0), the
        learning-rate schedule restarts at 0 unless the bundle carries `batch` (schedule_step_from_checkpoint) -- the
        reference's behaviour for its own bundles."""
        from . import tf_checkpoint as tfc
        arrays = tfc.load_checkpoint(prefix)
        n = 0
        for buf, suffix in ((self.params, ""), (self.m, "/Adam"), (self.v, "/Adam_1")):
            for name in VARIABLE_ORDER:
                a = arrays.get(name + suffix)
                if a is not None and tuple(a.shape) == tuple(self.flat.shapes[name]):
                    self.flat.view(buf, name).copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)))
                    n += 1
        self.adam_t = adam_step_from_checkpoint(arrays, self.beta2, self.adam_t, underflow_step)
        self.step_count = schedule_step_from_checkpoint(arrays)
        if "batch" not in arrays and self.adam_t > 0:
            # the reference's Saver leaves `batch` out (train/train_sdf.py:285-286), so its own resume restarts the
            # learning-rate decay too -- say so instead of doing it silently; save(include_step=True) keeps it
            import warnings
            warnings.warn("restored a bundle without `batch` after %d Adam steps: the learning-rate schedule restarts at "
                          "step 0 (the reference's behaviour); save with include_step=True to carry the step" % self.adam_t)
        return n


def feed_from_batch(batch_data: Dict[str, np.ndarray], device, rank: int = 0, world: int = 1
                    ) -> Dict[str, torch.Tensor]:
    """the feed_dict of train/train_sdf.py:371-378 ('sdf' = sdf_val - 0.003, :375) as device tensors;
    with world > 1 this rank's shard of the batch (parallel.shard_batch)"""
    from .parallel import shard_batch
    B = batch_data["img"].shape[0]
    b0, b1 = shard_batch(B, world, rank) if world > 1 else (0, B)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a[b0:b1], np.float32)).to(device, non_blocking=True)

    return {"imgs": dev(batch_data["img"]), "sample_pc": dev(batch_data["sdf_pt"]),
            "sample_pc_rot": dev(batch_data["sdf_pt_rot"]), "trans_mat": dev(batch_data["trans_mat"]),
            "sdf": dev(batch_data["sdf_val"] - np.float32(0.003))}


def train_one_epoch(trainer: Trainer, dataset, num_batches: int, log=print, log_every: int = 20,
                    rank: int = 0) -> Dict[str, float]:
    """train/train_sdf.py:349-440: fetch -> feed -> one optimizer step -> running means of the five
    losses, a log line every `log_every` batches.  Loss scalars stay on the device until a log line
    needs them (one host sync per log line instead of one per batch)."""
    import time
    sums = torch.zeros(len(LOSS_NAMES), dtype=torch.float64, device=trainer.params.device)
    window = torch.zeros_like(sums)
    fetch_time, tic = 0.0, time.time()
    for batch_idx in range(num_batches):
        t0 = time.time()
        batch_data = dataset.fetch()
        fetch_time += time.time() - t0
        feed = feed_from_batch(batch_data, trainer.params.device, rank, trainer.world)
        _, losses, lr = trainer.step(feed)
        vals = torch.stack([losses[n] for n in LOSS_NAMES]).to(torch.float64)
        sums += vals
        window += vals
        if (batch_idx + 1) % log_every == 0:
            w = (window / log_every).tolist()
            log("batch %d/%d lr %.3g  %s  | %.3f s/batch, fetch %.3f s" % (
                batch_idx + 1, num_batches, lr, "  ".join("%s %.5g" % (n, v) for n, v in zip(LOSS_NAMES, w)),
                (time.time() - tic) / log_every, fetch_time / log_every))
            window.zero_()
            fetch_time, tic = 0.0, time.time()
    means = (sums / max(num_batches, 1)).tolist()
    return dict(zip(LOSS_NAMES, means))
