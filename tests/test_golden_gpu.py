"""GPU parity against the committed golden vectors (tests/golden/*.npz, made by make_golden.py from
the CPU oracle): nothing here needs /root/reference or recomputes the expected values."""
import os

import numpy as np
import pytest
import torch

from visual_odometry_amd import cv2_shim, features, ops, params, synth
from visual_odometry_amd.twoDtwoD import calculate_relative_pose, twoDtwoD_from_matches

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def _d(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("n", [64, 1024])
def test_knn_golden(dev, n):
    g = load("knn.npz")
    idx, dist, _ = features.knn2_arrays(g[f"q{n}"].astype(np.float32), g[f"t{n}"].astype(np.float32))
    assert np.array_equal(idx, g[f"idx{n}"]) and np.array_equal(dist, g[f"dist{n}"])
    if n == 1024:
        for T in (0.6, 0.8):
            qi, ti = features.match_features_indices(g["q1024"].astype(np.float32), g["t1024"].astype(np.float32), T)
            keep = g[f"keep1024_{int(T * 10)}"]
            assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, g["idx1024"][keep, 0])


def test_knn_4096_golden(dev):
    g = load("knn.npz")
    q, t = synth.sift_like_pair(4096, 4096, seed=int(g["seed4096"][0]))
    if not np.array_equal(g["checksum4096"], [q.sum(dtype=np.float64), t.sum(dtype=np.float64)]):
        pytest.skip("numpy's generator stream differs from the one that made the fixture")
    idx, dist, _ = features.knn2_arrays(q, t)
    assert np.array_equal(idx, g["idx4096"]) and np.array_equal(dist, g["dist4096"])


def test_knn_4096_hashed_golden(dev):
    """The same size with inputs from the integer-hash generator: regenerated bit for bit on any numpy, so this one
    cannot skip (a checksum mismatch is a failure)."""
    g = load("knn.npz")
    q, t = synth.hashed_descriptor_pair(4096, 4096, seed=int(g["seedh4096"][0]))
    w = np.arange(128) + 1
    assert [int((q.astype(np.int64) * w).sum()), int((t.astype(np.int64) * w).sum())] == [int(v) for v in g["checksumh4096"]]
    idx, dist, _ = features.knn2_arrays(q, t)
    assert np.array_equal(idx, g["idxh4096"]) and np.array_equal(dist, g["disth4096"])


def test_scoring_golden(dev):
    g = load("scoring.npz")
    pts4 = _d(np.concatenate([g["p1"], g["p2"]], 1).astype(np.float32), dev)
    Fm = _d(g["fund_models"], dev)
    for thr in (0.1, 1.0):
        assert np.array_equal(ops.score_fund(Fm, pts4, thr).cpu().numpy(), g[f"fund_counts_{thr}"])
    b = int(g["fund_best"][0])
    assert np.array_equal(ops.mask_fund(Fm[b], pts4, 1.0).cpu().numpy(), g["fund_best_mask"])
    x4 = _d(np.concatenate([g["x1"], g["x2"]], 1), dev)
    assert np.array_equal(ops.score_ess(_d(g["ess_models"], dev), x4, float(g["ess_thr"][0])).cpu().numpy(), g["ess_counts"])
    for thr in (1.0, 5.0):
        cnt = ops.score_pnp(_d(g["pnp_models"], dev), _d(g["obj"], dev), _d(g["img"], dev), g["Kp"], thr)
        assert np.array_equal(cnt.cpu().numpy(), g[f"pnp_counts_{thr}"])


@pytest.mark.parametrize("c", range(6))
def test_replay_golden(dev, c):
    g = load("replay.npz")
    res = ops.ransac_replay(_d(g[f"counts{c}"], dev), _d(g[f"nmodels{c}"], dev), 1000, 7, 0.9999, 500)
    assert np.array_equal(res.cpu().numpy(), g[f"result{c}"])


def test_subsets_golden():
    g = load("rng.npz")
    for count in (15, 100, 8192):
        for m in (4, 5, 7):
            assert np.array_equal(ops.gen_subsets(count, m, 64), g[f"subsets_{count}_{m}"])
    assert np.array_equal(ops.gen_subsets(100, 7, 64, np.concatenate([g["fund_p1"], g["fund_p2"]], 1)), g["subsets_fund_100"])


def test_end_to_end_golden(dev):
    g = load("end_to_end.npz")
    thr, conf, it = g["f_args"]
    F, mask = cv2_shim.findFundamentalMat(g["f_p1"], g["f_p2"], cv2_shim.RANSAC, thr, conf, int(it))
    assert np.array_equal(F, g["f_F"]) and np.array_equal(mask, g["f_mask"])
    prob, thr, it = g["e_args"]
    E, emask = cv2_shim.findEssentialMat(g["f_p1"], g["f_p2"], g["e_K"], cv2_shim.RANSAC, prob, thr, int(it))
    assert np.array_equal(E, g["e_E"]) and np.array_equal(emask, g["e_mask"])
    it, thr, conf = g["p_args"]
    ok, rvec, tvec, inl = cv2_shim.solvePnPRansac(g["p_obj"], g["p_img"], g["p_K"], None, iterationsCount=int(it),
                                                  reprojectionError=thr, confidence=conf, flags=cv2_shim.SOLVEPNP_P3P)
    assert ok and np.array_equal(inl, g["p_inliers"])
    assert np.allclose(rvec, g["p_rvec"], rtol=1e-6, atol=1e-9) and np.allclose(tvec, g["p_tvec"], rtol=1e-6, atol=1e-9)


def test_pose_tail_end_to_end(oracle_mod, dev, capsys):
    """twoDtwoD's F -> E -> decompose -> disambiguate chain (reference src/twoDtwoD.py:57-94) on a
    synthetic pair: recovers the true rotation and translation direction."""
    params.which_dataset(2)          # Parking: 10000 iterations, thr 0.1
    try:
        sc = synth.two_view_scene(1500, K=synth.K_PARKING, size=synth.IMAGE_SIZE["parking"], seed=77,
                                  noise_px=0.02, outlier_frac=0.2)
        p1, p2 = sc["p1"].astype(np.float64), sc["p2"].astype(np.float64)
        R, t = calculate_relative_pose(p1, p2, sc["K"])
        assert np.abs(R - sc["R"]).max() < 2e-2
        assert np.abs(t - sc["t"]).max() < 0.1 and abs(np.linalg.norm(t) - 1) < 1e-12
        M, P_world, pi, pj = twoDtwoD_from_matches(p1, p2, sc["K"])
        assert M.shape == (4, 4) and P_world.shape[1] == 3 and pi.shape == pj.shape == (P_world.shape[0], 2)
        assert P_world.shape[0] > 300
    finally:
        params.which_dataset(0)


def test_pose_golden(dev):
    """tests/golden/pose.npz: triangulation bit for bit, the disambiguation chain and the EPnP refit to
    1e-6, the LMedS branch bit for bit."""
    from visual_odometry_amd import ransac
    g = load("pose.npz")
    X = ops.triangulate(g["tri_P1"], g["tri_P2"], _d(g["tri_x1"], dev), _d(g["tri_x2"], dev)).cpu().numpy()
    assert np.array_equal(X, g["tri_X"])
    n = len(g["tail_p1"])
    pts4 = np.concatenate([g["tail_p1"], g["tail_p2"]], 1).astype(np.float32)
    out = ops.pose_tail_batch(_d(g["tail_F"].reshape(1, 9), dev), _d(g["tail_mask"].reshape(1, n), dev), _d(pts4[None], dev),
                              g["tail_K"], 50.0)
    sel = g["tail_mask"].ravel() == 1
    keep = out["keep"][0].cpu().numpy().astype(bool)
    full = np.zeros(n, bool); full[np.nonzero(sel)[0][g["tail_keep"]]] = True
    assert np.array_equal(keep, full)
    assert np.allclose(out["T"][0].cpu().numpy(), g["tail_T"], rtol=1e-6, atol=1e-9)
    assert np.allclose(out["X"][0].cpu().numpy()[keep], g["tail_P"], rtol=1e-6, atol=1e-8)
    m = len(g["epnp_obj"])
    Rt, ok = ops.epnp_batch(_d(g["epnp_obj"][None], dev), _d(g["epnp_img"][None], dev), _d(np.ones((1, m), np.uint8), dev), g["epnp_K"])
    assert int(ok[0]) == 1
    Rt = Rt[0].cpu().numpy()
    assert np.allclose(Rt[:9].reshape(3, 3), g["epnp_R"], rtol=1e-6, atol=1e-8) and np.allclose(Rt[9:], g["epnp_t"].ravel(), rtol=1e-6, atol=1e-8)
    for k in (9, 13):
        F, mask, info = ransac.find_fundamental_ransac(g[f"lmeds{k}_p1"], g[f"lmeds{k}_p2"], 1.0, 0.99, 200)
        assert np.array_equal(mask, g[f"lmeds{k}_mask"])
        assert (info.best_iter, info.best_model, info.max_good, info.niters) == tuple(int(v) for v in g[f"lmeds{k}_info"])
        gF = g[f"lmeds{k}_F"]
        assert (F is None and gF.shape[0] == 0) or np.array_equal(F, gF)
