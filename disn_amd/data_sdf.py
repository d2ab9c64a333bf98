"""Training-data loader -- host mirror of data/data_sdf_h5_queue.py (`Pt_sdf_img`).

Same constructor, same producer thread + bounded queue, same `fetch()` / `shutdown()` contract and the
same batch dictionary (data/data_sdf_h5_queue.py:238-303):

    pc [B,num_points,3]  sdf_pt [B,S,3]  sdf_pt_rot [B,S,3]  sdf_val [B,S,1]  norm_params [B,4]
    sdf_params [B,6]  img [B,H,W,3]  trans_mat [B,4,3]  cat_id / obj_nm / view_id (lists)

Per-sample sources, as the reference lays them out (:71-76, :121-186):
    <sdf_dir>/<cat>/<obj>/ori_sample.h5   pc_sdf_original [n,4], pc_sdf_sample [m,4], norm_params [4], sdf_params [6]
    <rendered_dir>/<cat>/<obj>/%02d.h5    img_arr [H,W,4] uint8 (RGBA), trans_mat [4,3], obj_rot_mat [3,3], regress_mat [4,3]
HDF5 needs h5py, which this image does not have: every file may instead be a `.npz` with the same
keys next to (or instead of) the `.h5` (`save_sample` / `save_view` write them); a `.h5` without h5py
raises a clear error instead of guessing.

Epoch order (:305-317): a fresh shuffle of all samples at the start of every epoch, taking at most
`cats_limit[cat]` samples of each category; sampling of the S query points (:267-275): without
replacement when the object has at least S, else with replacement; `sdf_pt_rot = sdf_pt @ obj_rot_mat`
under FLAGS.rot.  Only the img_feat_twostream / regression branch of the reference is mirrored (the
one the SDF path uses); colour augmentation flags are accepted and ignored -- in the reference they
are no-ops too (the augmented values are computed and dropped, :158-170).
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np


def _load(path_h5: str, keys: Sequence[str]) -> Dict[str, np.ndarray]:
    npz = os.path.splitext(path_h5)[0] + ".npz"
    if os.path.exists(npz):
        with np.load(npz) as z:
            return {k: z[k] for k in keys if k in z.files}
    if os.path.exists(path_h5):      # the reference's own files (data/data_sdf_h5_queue.py:121-186)
        try:
            import h5py
        except ImportError:
            from .hdf5_lite import Hdf5File        # the subset of HDF5 those files use, restated in plain Python
            f = Hdf5File(path_h5)
            return {k: f[k] for k in keys if k in f}
        with h5py.File(path_h5, "r") as f:
            return {k: f[k][:] for k in keys if k in f.keys()}
    raise FileNotFoundError(path_h5)


def save_sample(sdf_dir: str, cat_id: str, obj: str, pc_sdf_original, pc_sdf_sample, norm_params, sdf_params):
    d = os.path.join(sdf_dir, cat_id, obj)
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, "ori_sample.npz"), pc_sdf_original=np.asarray(pc_sdf_original, np.float32),
             pc_sdf_sample=np.asarray(pc_sdf_sample, np.float32), norm_params=np.asarray(norm_params, np.float32),
             sdf_params=np.asarray(sdf_params, np.float32))


def save_view(rendered_dir: str, cat_id: str, obj: str, num: int, img_arr, trans_mat, obj_rot_mat, regress_mat):
    d = os.path.join(rendered_dir, cat_id, obj)
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, "%02d.npz" % num), img_arr=np.asarray(img_arr, np.uint8),
             trans_mat=np.asarray(trans_mat, np.float32), obj_rot_mat=np.asarray(obj_rot_mat, np.float32),
             regress_mat=np.asarray(regress_mat, np.float32))


class Pt_sdf_img(threading.Thread):
    def __init__(self, FLAGS, listinfo=None, info=None, qsize=64, cats_limit=None, shuffle=True, seed=None):
        super().__init__(daemon=True)
        self.queue: "queue.Queue" = queue.Queue(qsize)
        self.stopped = False
        self.bno = 0
        self.listinfo: List = list(listinfo)
        self.num_points = FLAGS.num_points
        self.gen_num_pt = FLAGS.num_sample_points
        self.batch_size = FLAGS.batch_size
        self.img_dir = info["rendered_dir"]
        self.sdf_dir = info["sdf_dir"]
        self.data_num = len(self.listinfo)
        self.FLAGS = FLAGS
        self.shuffle = shuffle
        self.num_batches = self.data_num // self.batch_size
        if cats_limit is None:
            cats_limit = {}
            for cat_id, _, _ in self.listinfo:
                cats_limit[cat_id] = cats_limit.get(cat_id, 0) + 1
        self.cats_limit, self.epoch_amount = self.set_cat_limit(dict(cats_limit))
        self.data_order = list(range(self.data_num))
        self.order = self.data_order
        self.rng = np.random.default_rng(seed)

    def set_cat_limit(self, cats_limit):
        cap = getattr(self.FLAGS, "cat_limit", None)
        total = 0
        for cat in cats_limit:
            if cap is not None:
                cats_limit[cat] = min(cap, cats_limit[cat])
            total += cats_limit[cat]
        return cats_limit, total

    def __len__(self):
        return self.epoch_amount

    # ---- per-sample sources ---------------------------------------------------------------
    def get_sdf_h5_filenm(self, cat_id, obj):
        return os.path.join(self.sdf_dir, cat_id, obj, "ori_sample.h5")

    def get_sdf_h5(self, sdf_h5_file, cat_id, obj):
        d = _load(sdf_h5_file, ("pc_sdf_original", "pc_sdf_sample", "norm_params", "sdf_params"))
        if not all(k in d for k in ("pc_sdf_original", "pc_sdf_sample", "norm_params")):
            raise Exception(cat_id, obj, "no sdf and sample")
        ori = d["pc_sdf_original"].astype(np.float32)
        smp = d["pc_sdf_sample"].astype(np.float32)
        if smp.shape[1] == 4:
            sample_pt, sample_val = smp[:, :3], smp[:, 3]
        else:
            sample_pt, sample_val = None, smp[:, 0]
        return ori[:, :3], None, sample_pt, sample_val, d["norm_params"], d["sdf_params"]

    def get_img(self, img_dir, num):
        d = _load(os.path.join(img_dir, "%02d.h5" % num), ("img_arr", "trans_mat", "obj_rot_mat", "regress_mat"))
        raw = d["img_arr"]
        img = raw[:, :, :3].astype(np.float32)
        if getattr(self.FLAGS, "backcolorwhite", False) and raw.shape[2] > 3:
            img[raw[:, :, 3] == 0] = 255.0
        img = np.clip(img, 0, 255) / np.float32(255.0)
        return (img, None, None, d["trans_mat"].astype(np.float32), d["obj_rot_mat"].astype(np.float32),
                d["regress_mat"].astype(np.float32))

    def getitem(self, index):
        cat_id, obj, num = self.listinfo[index]
        ori_pt, ori_val, sample_pt, sample_val, norm_params, sdf_params = self.get_sdf_h5(
            self.get_sdf_h5_filenm(cat_id, obj), cat_id, obj)
        return (ori_pt, ori_val, sample_pt, sample_val, norm_params, sdf_params,
                os.path.join(self.img_dir, cat_id, obj), None, cat_id, obj, num)

    # ---- batches ------------------------------------------------------------------------------
    def get_batch(self, index):
        B, S = self.batch_size, self.gen_num_pt
        if index + B > self.epoch_amount:
            index = index + B - self.epoch_amount
        out = {"pc": np.zeros((B, self.num_points, 3), np.float32), "sdf_pt": np.zeros((B, S, 3), np.float32),
               "sdf_pt_rot": np.zeros((B, S, 3), np.float32), "sdf_val": np.zeros((B, S, 1), np.float32),
               "norm_params": np.zeros((B, 4), np.float32), "sdf_params": np.zeros((B, 6), np.float32),
               "img": np.zeros((B, self.FLAGS.img_h, self.FLAGS.img_w, 3), np.float32),
               "trans_mat": np.zeros((B, 4, 3), np.float32), "cat_id": [], "obj_nm": [], "view_id": []}
        for cnt, i in enumerate(range(index, index + B)):
            ori_pt, _, sample_pt, sample_val, norm_params, sdf_params, img_dir, _, cat_id, obj, num = \
                self.getitem(self.order[i])
            img, _, _, trans_mat, obj_rot_mat, _ = self.get_img(img_dir, num)
            out["pc"][cnt] = ori_pt[self.rng.integers(ori_pt.shape[0], size=self.num_points)]
            n = sample_pt.shape[0]
            choice = self.rng.integers(n, size=S) if S > n else self.rng.choice(n, size=S, replace=False)
            out["sdf_pt"][cnt] = sample_pt[choice]
            out["sdf_val"][cnt, :, 0] = sample_val[choice]
            out["sdf_pt_rot"][cnt] = sample_pt[choice] @ obj_rot_mat if getattr(self.FLAGS, "rot", False) \
                else sample_pt[choice]
            out["norm_params"][cnt] = norm_params
            out["sdf_params"][cnt] = sdf_params
            out["img"][cnt] = img
            out["trans_mat"][cnt] = trans_mat
            out["cat_id"].append(cat_id)
            out["obj_nm"].append(obj)
            out["view_id"].append(num)
        return out

    def refill_data_order(self):
        order = list(self.data_order)
        self.rng.shuffle(order)
        quota = dict(self.cats_limit)
        epoch_order = []
        for idx in order:
            if len(epoch_order) >= self.epoch_amount:
                break
            cat_id = self.listinfo[idx][0]
            if quota.get(cat_id, 0) > 0:
                epoch_order.append(idx)
                quota[cat_id] -= 1
        return epoch_order

    def work(self, epoch, index):
        if index == 0 and self.shuffle:
            self.order = self.refill_data_order()
        return self.get_batch(index)

    def run(self):
        per_epoch = self.num_batches * self.batch_size
        while per_epoch > 0 and (self.bno // per_epoch) < self.FLAGS.max_epoch and not self.stopped:
            batch = self.work(self.bno // per_epoch, self.bno % per_epoch)
            while not self.stopped:
                try:
                    self.queue.put(batch, timeout=0.2)
                    break
                except queue.Full:
                    continue
            self.bno += self.batch_size

    def fetch(self, timeout: Optional[float] = None):
        if self.stopped:
            return None
        return self.queue.get(timeout=timeout)

    def shutdown(self):
        self.stopped = True
        while not self.queue.empty():
            try:
                self.queue.get_nowait()
            except queue.Empty:
                break
