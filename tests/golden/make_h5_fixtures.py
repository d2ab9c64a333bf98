"""HDF5 fixtures WRITTEN BY libhdf5 (VERDICT r4 #7 / missing #4): the reference's own writer calls executed with the h5py
that ships in this image's conda tree -- h5py 3.3.0 on HDF5 1.10.6, /opt/conda/bin/python3.9 (not importable from the
system interpreter; found in round 5) -- so that disn_amd/hdf5_lite.py is checked against files it did not shape itself.

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py        (writes tests/golden/h5/*.h5 + expected.npz)

Files (small versions of what data/data_sdf_h5_queue.py:121-186 opens):
  ori_sample.h5   preprocessing/create_point_sdf_grid.py:154-158 verbatim: pc_sdf_original / pc_sdf_sample / norm_params /
                  sdf_params, compression='gzip', compression_opts=4 (h5py picks the chunk shapes, as for the reference)
  img_00.h5       preprocessing/create_img_h5.py:188-200 verbatim: img_arr uint8 [137,137,4] + trans_mat / K / RT /
                  obj_rot_mat / regress_mat with dtype='float32' (float64 data converted by libhdf5 on write)
  layouts.h5      what else h5py may produce for such datasets: contiguous (no compression), compact-size scalars, a
                  chunked + shuffle + gzip dataset, a multi-chunk dataset whose chunk B-tree has more than one leaf entry
                  per dimension, a dataset in a sub-group
expected.npz holds the arrays as h5py reads them back (float32 / uint8 / float64 as stored).
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    exp = {}
    # ---- ori_sample.h5 : create_point_sdf_grid.py:153-158 ----
    ori_verts = rng.normal(size=(3000, 3))
    samplesdf = np.concatenate([rng.uniform(-1, 1, (12000, 3)), rng.normal(0, 0.05, (12000, 1))], axis=1)
    centroid, m = rng.normal(size=3).astype(np.float32), 1.37
    norm_params = np.concatenate((centroid, np.asarray([m]).astype(np.float32)))
    param = [-0.51, -0.52, -0.53, 0.54, 0.55, 0.56]
    p = os.path.join(OUT, "ori_sample.h5")
    f1 = h5py.File(p, 'w')
    f1.create_dataset('pc_sdf_original', data=ori_verts.astype(np.float32), compression='gzip', compression_opts=4)
    f1.create_dataset('pc_sdf_sample', data=samplesdf.astype(np.float32), compression='gzip', compression_opts=4)
    f1.create_dataset('norm_params', data=norm_params, compression='gzip', compression_opts=4)
    f1.create_dataset('sdf_params', data=param, compression='gzip', compression_opts=4)
    f1.close()
    # ---- img_00.h5 : create_img_h5.py:188-200 ----
    img_arr = rng.integers(0, 256, (137, 137, 4)).astype(np.uint8)
    img_arr[:40] = 255                                           # a compressible region (white background)
    trans_mat_right, K, RT = rng.normal(size=(4, 3)), rng.normal(size=(3, 3)), rng.normal(size=(3, 4))
    obj_rot_mat, regress_mat = rng.normal(size=(3, 3)), rng.normal(size=(4, 3))
    p2 = os.path.join(OUT, "img_00.h5")
    with h5py.File(p2, 'w') as f1:
        f1.create_dataset('img_arr', data=img_arr, compression='gzip', dtype='uint8', compression_opts=4)
        f1.create_dataset('trans_mat', data=trans_mat_right, compression='gzip', dtype='float32', compression_opts=4)
        f1.create_dataset('K', data=K, compression='gzip', dtype='float32', compression_opts=4)
        f1.create_dataset('RT', data=RT, compression='gzip', dtype='float32', compression_opts=4)
        f1.create_dataset('obj_rot_mat', data=obj_rot_mat, compression='gzip', dtype='float32', compression_opts=4)
        f1.create_dataset('regress_mat', data=regress_mat, compression='gzip', dtype='float32', compression_opts=4)
    # ---- layouts.h5 ----
    p3 = os.path.join(OUT, "layouts.h5")
    with h5py.File(p3, 'w') as f:
        f.create_dataset('contig_f32', data=rng.normal(size=(33, 5)).astype(np.float32))
        f.create_dataset('contig_f64', data=rng.normal(size=(7,)))
        f.create_dataset('scalar_i32', data=np.int32(-12345))
        f.create_dataset('shuffled', data=rng.normal(size=(300, 4)).astype(np.float32), compression='gzip',
                         compression_opts=9, shuffle=True, chunks=(64, 4))
        f.create_dataset('many_chunks', data=np.arange(50 * 70, dtype=np.int64).reshape(50, 70), chunks=(8, 16),
                         compression='gzip', compression_opts=1)
        f.create_dataset('u8_plain', data=rng.integers(0, 256, (20, 3)).astype(np.uint8))
        g = f.create_group('grp')
        g.create_dataset('inner', data=rng.normal(size=(4, 4)).astype(np.float32), compression='gzip')
    for fn in ("ori_sample.h5", "img_00.h5", "layouts.h5"):
        with h5py.File(os.path.join(OUT, fn), 'r') as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    exp["%s:%s" % (fn, name)] = obj[()]
                    print("%-14s %-16s %-10s %-14s chunks %-10s compression %s shuffle %s" % (
                        fn, name, obj.dtype, obj.shape, obj.chunks, obj.compression, obj.shuffle))
            f.visititems(visit)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version,
          {fn: os.path.getsize(os.path.join(OUT, fn)) for fn in os.listdir(OUT)})


if __name__ == "__main__":
    main()
