"""GPU parity of the 2D-2D pose tail: cv.triangulatePoints bit for bit against the
oracle's restatement; E = K^T F K -> decomposeEssentialMatrix -> disambiguateRelativePose for a
batch of pairs against the oracle chain (numpy SVD for E there, Jacobi here: tolerance 1e-6 on
poses and points, identical candidate scores and masks on non-degenerate scenes)."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import cv2_shim, ops, pipeline, synth
from visual_odometry_amd.twoDtwoD import twoDtwoD_from_pipeline

pytestmark = pytest.mark.gpu


def _d(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_triangulate_against_reference_linear_triangulation(dev):
    """tests/golden/ref_helpers.npz holds the output of the reference's own numpy DLT triangulation
    (src/utils/linear_triangulation.py) on exact projections of a noise-free scene; the device
    triangulation (through the cv2-shaped entry point the reference calls) must land on the same points."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_helpers.npz"))
    X = cv2_shim.triangulatePoints(g["tri_M1"], g["tri_M2"], g["tri_p1"].T, g["tri_p2"].T)
    P = X / X[3]
    scale = np.linalg.norm(g["tri_P"][:3], axis=0)
    assert (np.linalg.norm(P[:3] - g["tri_P"][:3], axis=0) / scale).max() < 1e-9


@pytest.mark.parametrize("n,seed", [(1, 0), (63, 1), (257, 2), (5000, 3)])
def test_triangulate_equals_oracle(det_math, dev, n, seed):
    """bit for bit against the oracle in its det-math mode (hypot evaluated as sqrt(a*a+b*b), as the
    device does; the libm-mode oracle agrees to rounding)."""
    oracle_mod = det_math
    sc = synth.two_view_scene(max(n, 8), seed=seed)
    K = sc["K"]
    P1 = K @ np.eye(3, 4)
    P2 = K @ np.c_[sc["R"], sc["t"]]
    x1 = sc["p1"][:n].astype(np.float64); x2 = sc["p2"][:n].astype(np.float64)
    X = ops.triangulate(P1, P2, _d(x1, dev), _d(x2, dev)).cpu().numpy()
    ref = oracle_mod.triangulate_points(P1, P2, x1.T, x2.T).T
    assert np.array_equal(X, ref)
    # the cv2-shaped entry point: (2,N) in, (4,N) out
    Xs = cv2_shim.triangulatePoints(P1, P2, x1.T, x2.T)
    assert Xs.shape == (4, n) and np.array_equal(Xs.T, ref)
    oracle_mod.set_det_math(False)
    lib = oracle_mod.triangulate_points(P1, P2, x1.T, x2.T).T
    oracle_mod.set_det_math(True)
    assert np.allclose(lib / lib[:, 3:4], ref / ref[:, 3:4], rtol=1e-7, atol=1e-9)
    # and it is a triangulation: reprojection of inlier points lands near the observations
    Xh = ref / ref[:, 3:4]
    rp = (P1 @ Xh.T).T
    rp = rp[:, :2] / rp[:, 2:3]
    assert np.median(np.linalg.norm(rp - x1, axis=1)) < 2.0


def test_triangulate_degenerate_inputs(det_math, dev):
    oracle_mod = det_math
    """identical cameras (rank-deficient system), points at the principal point, huge coordinates."""
    K = synth.K_KITTI
    P1 = K @ np.eye(3, 4)
    x = np.array([[K[0, 2], K[1, 2]], [0.0, 0.0], [1e6, -1e6], [100.0, 50.0]])
    for P2 in (P1, K @ np.c_[np.eye(3), [1.0, 0, 0]]):
        X = ops.triangulate(P1, P2, _d(x, dev), _d(x[::-1].copy(), dev)).cpu().numpy()
        with np.errstate(all="ignore"):
            ref = oracle_mod.triangulate_points(P1, P2, x.T, x[::-1].T).T
        assert np.array_equal(X, ref, equal_nan=True)
    assert ops.triangulate(P1, P1, _d(x[:0], dev), _d(x[:0], dev)).shape == (0, 4)


def _oracle_tail(o, F, mask, p1, p2, K, max_depth):
    E = K.T @ F @ K
    Rs, u3 = o.decompose_essential_matrix(E)
    sel = mask.ravel() == 1
    ph1 = np.c_[p1[sel].astype(np.float64), np.ones(sel.sum())].T
    ph2 = np.c_[p2[sel].astype(np.float64), np.ones(sel.sum())].T
    T, P, m = o.disambiguate_relative_pose(Rs, u3, ph1, ph2, K, K, max_depth)
    return T, P, m, sel


def test_pose_tail_batch_vs_oracle_chain(det_math, dev):
    o = det_math
    B, N = 5, 1200
    K = synth.K_KITTI
    Fs, masks, pts = [], [], []
    scenes = [synth.two_view_scene(N - 100 * b, K=K, seed=40 + b, outlier_frac=0.25) for b in range(B)]
    n = np.array([len(sc["p1"]) for sc in scenes], np.int32)
    for sc in scenes:
        F, mask, info = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.999, 300)
        assert F is not None and mask.sum() > 100
        Fs.append(F.reshape(9)); masks.append(np.pad(mask.ravel(), (0, N - len(mask))))
        pts.append(np.pad(np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32), ((0, N - len(mask)), (0, 0))))
    result = np.zeros((B, 4), np.int32)
    result[B - 1, 0] = -1                                  # last pair: "no model" -> identity, nothing kept
    out = ops.pose_tail_batch(_d(np.array(Fs), dev), _d(np.array(masks, np.uint8), dev), _d(np.array(pts), dev), K, 50.0,
                              result=_d(result, dev), n_dev=_d(n, dev))
    T, X, keep, score = (out[k].cpu().numpy() for k in ("T", "X", "keep", "score"))
    for b, sc in enumerate(scenes[:-1]):
        oT, oP, om, sel = _oracle_tail(o, Fs[b].reshape(3, 3), masks[b][:n[b]], sc["p1"], sc["p2"], K, 50.0)
        assert np.allclose(T[b], oT, rtol=1e-6, atol=1e-9), b
        full = np.zeros(N, bool); full[np.nonzero(sel)[0][om]] = True
        assert np.array_equal(keep[b].astype(bool), full), b
        assert np.allclose(X[b][full], oP, rtol=1e-6, atol=1e-8), b
        assert not X[b][~np.pad(sel, (0, N - len(sel)))].any()
        # the winner has the unique best cheirality score, and it is (nearly) everything
        assert np.sort(score[b])[-1] > np.sort(score[b])[-2] and score[b].max() >= 2 * sel.sum() - 4
        # rotation is a rotation
        assert np.allclose(T[b][:3, :3] @ T[b][:3, :3].T, np.eye(3), atol=1e-12) and np.linalg.det(T[b][:3, :3]) > 0
    assert np.array_equal(T[B - 1], np.eye(4)) and not keep[B - 1].any() and not score[B - 1].any()


def test_pose_tail_from_pipeline_equals_host_drop_in(det_math, dev):
    """pipeline.match_and_ransac + pose_tail + twoDtwoD_from_pipeline vs the host drop-in on the same
    matches (oracle chain): the tuple twoDtwoD returns (src/twoDtwoD.py:57-79)."""
    from test_pipeline_gpu import _pair
    o = det_math
    B, nq, nt = 2, 1500, 1700
    pairs = [_pair(300 + b, nq, nt) for b in range(B)]
    st = lambda k: _d(np.stack([p[k] for p in pairs]), dev)
    K = synth.K_MALAGA
    r = pipeline.match_and_ransac(st(0), st(1), st(2), st(3), 0.8, 1.0, 0.999, 300)
    tail = pipeline.pose_tail(r, K, 50.0)
    for b, (q, t, kq, kt, xyz) in enumerate(pairs):
        oq, ot, _, _ = o.match_features_indices(q, t, 0.8)
        p1, p2 = kq[oq], kt[ot]
        F, mask, _ = o.find_fundamental_mat(p1, p2, 1.0, 0.999, 300)
        oT, oP, om, sel = _oracle_tail(o, F, mask, p1, p2, K, 50.0)
        M, P_world, pi, pj = twoDtwoD_from_pipeline(r, tail, b)
        assert np.allclose(M, o.hom_inv(oT), rtol=1e-6, atol=1e-9)
        assert np.allclose(P_world, oP[:, :3], rtol=1e-6, atol=1e-8)
        assert np.array_equal(pi, p1[sel][om].astype(np.float64)) and np.array_equal(pj, p2[sel][om].astype(np.float64))


def test_triangulate_track_equals_oracle_formula(det_math, dev, capsys):
    """src/tracking.py:150-172 with cv.triangulatePoints on the device vs the same lines over the oracle."""
    from visual_odometry_amd.landmarks import triangulate_track
    from visual_odometry_amd.utils.utils import hom_inv
    o = det_math
    sc = synth.two_view_scene(800, K=synth.K_PARKING, size=synth.IMAGE_SIZE["parking"], seed=9, outlier_frac=0.2)
    K = synth.K_PARKING
    c2w_i = np.eye(4); c2w_i[:3, 3] = [0.3, -0.1, 2.0]
    T_ij = np.eye(4); T_ij[:3, :3] = sc["R"]; T_ij[:3, 3] = sc["t"].ravel()
    c2w_j = c2w_i @ hom_inv(T_ij)
    kp_s, kp_e = sc["p1"].astype(np.float64), sc["p2"].astype(np.float64)
    P, kpj, mask = triangulate_track(K, c2w_i, c2w_j, kp_s, kp_e, 50.0)
    assert "new landmarks" in capsys.readouterr().out
    Ti, Tj = np.linalg.inv(c2w_i), np.linalg.inv(c2w_j)
    ref = o.triangulate_points(K @ Ti[:3], K @ Tj[:3], kp_s.T, kp_e.T).T
    ref = ref / ref[:, 3:4]
    pc = (Tj @ ref.T).T
    rmask = np.logical_and(pc[:, 2] > 0, np.abs(np.linalg.norm(pc, axis=1)) < 50.0)
    assert np.array_equal(mask, rmask) and 100 < mask.sum() < len(mask)
    assert np.array_equal(P, ref[rmask]) and np.array_equal(kpj[:, :2], kp_e[rmask])
    # the inlier landmarks are where the scene put them (camera-i frame -> world)
    good = sc["inlier"][mask]
    Xw = (c2w_i @ np.c_[sc["X"], np.ones(len(sc["X"]))].T).T[mask]
    err = np.linalg.norm(P[good, :3] - Xw[good, :3], axis=1) / np.linalg.norm(Xw[good, :3], axis=1)
    assert np.median(err) < 0.1


def test_triangulation_and_disambiguation_drop_ins(det_math, dev, capsys):
    """the host-shaped drop-ins (numpy in / numpy out) ride on the device triangulation."""
    oracle_mod = det_math
    from visual_odometry_amd.utils.decompose_essential_matrix import decomposeEssentialMatrix
    from visual_odometry_amd.utils.disambiguate_relative_pose import disambiguateRelativePose
    from visual_odometry_amd.utils.triangulate import triangulatePoints
    for seed in range(4):
        sc = synth.two_view_scene(400, seed=50 + seed, noise_px=0.2, outlier_frac=0.1)
        K, R, t = sc["K"], sc["R"], sc["t"]
        P1, P2 = K @ np.eye(3, 4), K @ np.c_[R, t]
        a = triangulatePoints(P1, P2, sc["p1"].T, sc["p2"].T)
        b = oracle_mod.triangulate_points(P1, P2, sc["p1"].T.astype(np.float64), sc["p2"].T.astype(np.float64))
        assert np.array_equal(a, b)
        good = sc["inlier"]
        assert np.abs((a / a[3])[:3, good].T - sc["X"][good]).max() < 0.2 * 40
        E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
        Rs, u3 = decomposeEssentialMatrix(E)
        p1h = np.c_[sc["p1"].astype(np.float64), np.ones(400)].T; p2h = np.c_[sc["p2"].astype(np.float64), np.ones(400)].T
        T, P, mask = disambiguateRelativePose(Rs, u3, p1h, p2h, K, K)
        oT, oP, omask = oracle_mod.disambiguate_relative_pose(Rs, u3, p1h, p2h, K, K)
        assert "rejected points during 3D reconstruction" in capsys.readouterr().out
        assert np.array_equal(T, oT) and np.array_equal(mask, omask) and np.array_equal(P, oP)
        assert np.allclose(T[:3, :3], R, atol=1e-9) and np.allclose(T[:3, 3], t, atol=1e-9)



def test_epnp_batch_vs_host_and_oracle(oracle_mod, dev):
    """device EPnP over masked correspondences of a ragged batch vs the host refit and the oracle."""
    from visual_odometry_amd.epnp import epnp_pose
    sizes = [3000, 500, 40, 8, 6, 5, 3, 700]
    N = max(sizes)
    K = synth.K_MALAGA
    rng = np.random.default_rng(3)
    obj = np.zeros((len(sizes), N, 3), np.float32); img = np.zeros((len(sizes), N, 2), np.float32)
    mask = np.zeros((len(sizes), N), np.uint8); result = np.zeros((len(sizes), 4), np.int32)
    for b, n in enumerate(sizes):
        ps = synth.pnp_scene(max(n, 8), K=K, seed=70 + b, outlier_frac=0.3 if n > 50 else 0.0, noise_px=0.4)
        obj[b, :n], img[b, :n] = ps["obj"][:n], ps["img"][:n]
        mask[b, :n] = ps["inlier"][:n] & (rng.uniform(size=n) < 0.9 if n > 50 else True)
    result[7, 0] = -1                                      # "no model" pair
    Rt, ok = ops.epnp_batch(_d(obj, dev), _d(img, dev), _d(mask, dev), K, n_dev=_d(np.array(sizes, np.int32), dev),
                            result=_d(result, dev))
    Rt, ok = Rt.cpu().numpy(), ok.cpu().numpy()
    for b, n in enumerate(sizes):
        sel = mask[b, :n].astype(bool)
        if b == 7 or sel.sum() < 4:
            assert ok[b] == 0 and not Rt[b].any()
            continue
        o64, i64 = obj[b, :n][sel].astype(np.float64), img[b, :n][sel].astype(np.float64)
        hok, hR, ht = epnp_pose(o64, i64, K)
        ook, oR, ot = oracle_mod.epnp(o64, i64, K)
        assert ok[b] == 1 and hok and ook
        R, t = Rt[b, :9].reshape(3, 3), Rt[b, 9:]
        assert np.allclose(R, hR, rtol=1e-6, atol=1e-8) and np.allclose(t, ht.ravel(), rtol=1e-6, atol=1e-8), b
        assert np.allclose(R, oR, rtol=1e-6, atol=1e-8) and np.allclose(t, ot.ravel(), rtol=1e-6, atol=1e-8), b
        assert abs(np.linalg.det(R) - 1) < 1e-9


def test_pipeline_refined_pose_equals_pnp_drop_in(det_math, dev):
    """pipeline.match_and_ransac's pose_refined = what cv.solvePnPRansac (oracle) returns for the pair."""
    from test_pipeline_gpu import _pair
    from visual_odometry_amd.rodrigues import rodrigues_vector_to_matrix
    o = det_math
    B, nq, nt = 2, 1500, 1700
    pairs = [_pair(500 + b, nq, nt) for b in range(B)]
    st = lambda k: _d(np.stack([p[k] for p in pairs]), dev)
    K = synth.K_MALAGA
    r = pipeline.match_and_ransac(st(0), st(1), st(2), st(3), 0.8, 1.0, 0.999, 200, xyz_q=st(4), K=K, p_threshold=2.0,
                                  p_confidence=0.999, p_iters=300, first_block_pnp=64)
    Rt, okd = r.pose_refined.cpu().numpy(), r.p_ok.cpu().numpy()
    for b, (q, t, kq, kt, xyz) in enumerate(pairs):
        oq, ot, _, _ = o.match_features_indices(q, t, 0.8)
        ok, rvec, tvec, inl, _ = o.solve_pnp_ransac(xyz[oq], kt[ot], K, 300, 2.0, 0.999)
        assert ok and okd[b] == 1
        assert np.allclose(Rt[b, :9].reshape(3, 3), rodrigues_vector_to_matrix(rvec.ravel()), rtol=1e-6, atol=1e-8)
        assert np.allclose(Rt[b, 9:], tvec.ravel(), rtol=1e-6, atol=1e-8)


def test_pose_tail_deferred_equals_waited(dev):
    """pipeline.pose_tail(wait=False) (side stream, not waited for by the caller's stream; what bench.py times) gives the
    bits of the waited call, also with a second batch enqueued behind it; ragged inlier densities (a pair without a
    model, a pair whose matches are nearly all inliers) go through the per-chunk inlier gathering."""
    B, nq, nt = 4, 2600, 2500
    K = synth.K_KITTI
    fps = [synth.frame_pair(nq, nt, seed=400 + b, K=K, size=synth.IMAGE_SIZE["kitti"]) for b in range(B)]
    st = lambda k: _d(np.stack([f[k] for f in fps]), dev)
    args = (st("q"), st("t"), st("kp_q"), st("kp_t"))
    r = pipeline.match_and_ransac(*args, 0.8, 1.0, 0.999, 200)
    r.f_mask[1, : int(r.n_match[1])] = 1                     # a dense pair: every match an "inlier"
    r.f_result[2, 0] = -1                                    # a pair without a model -> identity, no points
    want = pipeline.pose_tail(r, K, 50.0)
    torch.cuda.synchronize()
    got = pipeline.pose_tail(r, K, 50.0, wait=False)
    r2 = pipeline.match_and_ransac(*args, 0.8, 1.0, 0.999, 200)       # more work behind it on the caller's stream
    torch.cuda.current_stream().wait_event(got["done"])
    torch.cuda.synchronize()
    for k in ("cand", "score", "T", "X", "keep"):
        assert np.array_equal(got[k].cpu().numpy(), want[k].cpu().numpy(), equal_nan=True), k
    assert int(want["keep"][1].sum()) > 0 and int(want["keep"][2].sum()) == 0
    assert np.array_equal(want["T"][2].cpu().numpy().reshape(4, 4), np.eye(4))
    del r2
