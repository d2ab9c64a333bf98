"""Camera head: oracle properties on CPU, HIP parity on the GPU."""
import numpy as np
import pytest

from conftest import report_close
from oracle import cam_oracle as CO


def test_oracle_rotation_is_a_scaled_right_handed_frame():
    rng = np.random.default_rng(0)
    W = CO.init_weights(1)
    emb = np.maximum(rng.standard_normal((5, 1024)), 0).astype(np.float32)
    rot, trans, RT = CO.get_cam_mat(emb, W, np.float64)
    assert rot.shape == (5, 3, 3) and trans.shape == (5, 1, 3) and RT.shape == (5, 4, 3)
    for b in range(5):
        s = np.linalg.norm(rot[b][:, 0])
        # columns orthogonal, equal norm |s|, and (x, y, z) right-handed up to the sign of s
        G = rot[b].T @ rot[b]
        assert np.allclose(G, np.eye(3) * s * s, atol=1e-9 * max(1, s * s))
        assert np.isclose(abs(np.linalg.det(rot[b])), abs(s) ** 3, rtol=1e-9)
        assert np.array_equal(RT[b, :3], rot[b]) and np.array_equal(RT[b, 3], trans[b, 0])
    tm = CO.pred_trans_mat(RT)
    assert np.allclose(tm, RT @ CO.K_DEFAULT.T.astype(np.float64))


def test_oracle_ortho6d_known_answers():
    # identity frame from axis-aligned inputs; degenerate inputs hit the 1e-8 clamp, not a NaN
    p = np.array([[2, 0, 0, 0, 3, 0], [0, 0, 5, 1, 0, 0], [0, 0, 0, 0, 0, 0]], np.float64)
    R = CO.compute_rotation_matrix_from_ortho6d(p)
    assert np.allclose(R[0], np.eye(3))
    # x = e_z, z = n(e_z x e_x) = e_y, y = z x x = e_x
    assert np.allclose(R[1][:, 0], [0, 0, 1]) and np.allclose(R[1][:, 2], [0, 1, 0])
    assert np.allclose(R[1][:, 1], [1, 0, 0])
    assert np.isfinite(R[2]).all() and np.allclose(R[2], 0)


def test_variable_names_match_the_reference_scopes():
    names = set(CO.variable_shapes())
    assert "cameraprediction/ortho6d/fc1/weights" in names and len(names) == 18
    assert CO.variable_shapes()["cameraprediction/translation/fc3/weights"] == (64, 3)
    from disn_amd import posenet
    assert posenet.variable_shapes() == CO.variable_shapes()


@pytest.mark.gpu
def test_cam_head_matches_oracle():
    import torch
    from disn_amd.posenet import CameraHead
    rng = np.random.default_rng(3)
    W = CO.init_weights(2)
    emb = np.maximum(rng.standard_normal((7, 1024)), 0).astype(np.float32)
    head = CameraHead(W)
    rot, tr, RT, tm = head.run(torch.from_numpy(emb).cuda())
    r64, t64, RT64 = CO.get_cam_mat(emb, W, np.float64)
    # fp32 dots of 1024 terms + a normalisation: 1e-5 relative to the value scale
    report_close("rotation", rot.cpu().numpy(), r64, atol=1e-5 * np.abs(r64).max())
    report_close("translation", tr.cpu().numpy(), t64[:, 0], atol=1e-5 * np.abs(t64).max())
    report_close("RT", RT.cpu().numpy(), RT64, atol=1e-5 * np.abs(RT64).max())
    report_close("trans_mat", tm.cpu().numpy(), CO.pred_trans_mat(RT64), atol=1e-5 * np.abs(CO.pred_trans_mat(RT64)).max())
    K2 = np.array([[100.0, 0, 60], [0, 110.0, 70], [0, 0, 1]], np.float32)
    _, _, _, tm2 = head.run(torch.from_numpy(emb).cuda(), K2)
    report_close("trans_mat K", tm2.cpu().numpy(), CO.pred_trans_mat(RT64, K2), atol=1e-5 * 200)
    a, b, c = head.get_cam_mat(torch.from_numpy(emb).cuda())
    assert b.shape == (7, 1, 3) and torch.equal(c[:, :3], a) and torch.equal(c[:, 3:], b)


@pytest.mark.gpu
def test_camera_estimator_feeds_the_sdf_path():
    """image -> camera network -> pred_trans_mat -> SDF query with that camera (the reference's
    estimated-camera pipeline, cam_est/train_sdf_cam.py:568-612 + test/create_sdf.py)"""
    import torch
    from oracle import disn_oracle as O
    from disn_amd.engine import SdfEngine
    from disn_amd.posenet import CameraEstimator
    from disn_amd.weights import WeightStore
    store = WeightStore(O.init_weights(4, "he"))
    Wc = CO.init_weights(5)
    est = CameraEstimator(store, Wc)
    feed = O.synth_inputs(seed=6, batch=2, n_points=128)
    ep = est.get_model(feed["imgs"])
    emb = ep["embedding"].cpu().numpy()   # the encoder itself is covered by tests/test_gpu_model.py
    _, _, RT = CO.get_cam_mat(emb, Wc, np.float64)
    report_close("pred_trans_mat", ep["pred_trans_mat"].cpu().numpy(), CO.pred_trans_mat(RT),
                 atol=1e-5 * np.abs(CO.pred_trans_mat(RT)).max())
    eng = SdfEngine(store)
    pts = torch.from_numpy(feed["sample_pc"]).cuda()
    sdf = eng.encode_query(torch.from_numpy(feed["imgs"]).cuda(), pts, ep["pred_trans_mat"])[1]
    assert sdf.shape == (2, 128) and bool(torch.isfinite(sdf).all())
