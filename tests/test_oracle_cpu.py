"""No GPU: pins the CPU oracle -- against the committed golden vectors (regression), against
independent numpy restatements (kNN in exact integers, error formulas), against the algebraic
identities the solvers must satisfy, and against the reference's own pure-numpy helpers
(tests/golden/ref_helpers.npz).  Real OpenCV is absent: parity with it is unpinned."""
import os

import numpy as np
import pytest

from visual_odometry_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------------------------ kNN
def _knn_int(q, t):
    """independent restatement in exact integers: two smallest (D, j) lexicographically."""
    q = q.astype(np.int64); t = t.astype(np.int64)
    D = (q * q).sum(1)[:, None] + (t * t).sum(1)[None, :] - 2 * q @ t.T
    key = D * (1 << 20) + np.arange(t.shape[0])[None, :]
    o2 = np.argsort(key, axis=1, kind="stable")[:, :2]
    return o2.astype(np.int32), np.take_along_axis(D, o2, 1)


@pytest.mark.parametrize("n", [64, 1024])
def test_knn_oracle_vs_golden_and_integers(oracle_mod, n):
    g = load("knn.npz")
    q, t = g[f"q{n}"].astype(np.float32), g[f"t{n}"].astype(np.float32)
    idx, dist = oracle_mod.knn2(q, t)
    assert np.array_equal(idx, g[f"idx{n}"]) and np.array_equal(dist, g[f"dist{n}"])
    ii, DD = _knn_int(q, t)
    assert np.array_equal(idx, ii)
    assert np.array_equal(dist, np.sqrt(DD.astype(np.float32)))      # f32 accumulation was exact
    idx_mt, dist_mt = oracle_mod.knn2(q, t, mt=True)
    assert np.array_equal(idx, idx_mt) and np.array_equal(dist, dist_mt)


def test_knn_golden_structure():
    g = load("knn.npz")
    idx, dist = g["idx64"], g["dist64"]
    assert tuple(idx[7]) == (3, 5) and dist[7, 0] == 0 and dist[7, 1] == 0     # duplicates: lower index first
    assert tuple(idx[11]) == (2, 33)                                            # 2, 33, 34 tie: first two
    assert np.all(dist[:, 0] <= dist[:, 1])


def test_knn_4096_seeded(oracle_mod):
    g = load("knn.npz")
    q, t = synth.sift_like_pair(4096, 4096, seed=int(g["seed4096"][0]))
    if not np.array_equal(g["checksum4096"], [q.sum(dtype=np.float64), t.sum(dtype=np.float64)]):
        pytest.skip("numpy's generator stream differs from the one that made the fixture")
    idx, dist = oracle_mod.knn2(q, t, mt=True)
    assert np.array_equal(idx, g["idx4096"]) and np.array_equal(dist, g["dist4096"])


def test_ratio_test_golden(oracle_mod):
    g = load("knn.npz")
    for T in (0.6, 0.8):
        assert np.array_equal(oracle_mod.ratio_test(g["dist1024"], T), g[f"keep1024_{int(T * 10)}"])
    # strictness: equal distances never pass with T = 1
    d = np.array([[2.0, 2.0], [1.0, 2.0]], np.float32)
    assert list(oracle_mod.ratio_test(d, 1.0)) == [False, True]


def test_knn_edge_shapes(oracle_mod):
    q, t = synth.sift_like_pair(5, 1, 3)
    idx, dist = oracle_mod.knn2(q, t)
    assert np.all(idx[:, 0] == 0) and np.all(idx[:, 1] == -1) and np.all(dist[:, 1] == np.finfo(np.float32).max)
    idx, dist = oracle_mod.knn2(q[:0], t)
    assert idx.shape == (0, 2)


# ------------------------------------------------------------------------------------ RNG / subsets
def test_rng_stream_is_the_documented_mwc(oracle_mod):
    """cv::RNG: state = (u64)(u32)state * 4164903690 + (state >> 32), output = (u32)state."""
    g = load("rng.npz")
    state, out = 0xFFFFFFFFFFFFFFFF, []
    for _ in range(64):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(state & 0xFFFFFFFF)
    assert np.array_equal(oracle_mod.rng_stream(64), np.array(out, dtype=np.uint32))
    assert np.array_equal(g["stream64"], np.array(out, dtype=np.uint32))


def test_subsets_golden_and_python_restatement(oracle_mod):
    g = load("rng.npz")
    for count in (15, 100, 8192):
        for m in (4, 5, 7):
            assert np.array_equal(oracle_mod.gen_subsets(count, m, 64), g[f"subsets_{count}_{m}"])
    # pure-Python getSubset for one case
    state = 0xFFFFFFFFFFFFFFFF
    def nxt():
        nonlocal state
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return state & 0xFFFFFFFF
    rows = []
    for _ in range(64):
        idx = []
        while len(idx) < 4:
            v = nxt() % 15
            if v not in idx:
                idx.append(v)
        rows.append(idx)
    assert np.array_equal(g["subsets_15_4"], np.array(rows, dtype=np.int32))
    assert np.array_equal(oracle_mod.gen_subsets(100, 7, 64, g["fund_p1"], g["fund_p2"]), g["subsets_fund_100"])


def test_update_num_iters_golden_and_formula(oracle_mod):
    g = load("rng.npz")
    for (p, ep, m, mi), exp in zip(g["update_cases"], g["update_expected"]):
        assert oracle_mod.update_num_iters(p, ep, int(m), int(mi)) == exp
    # closed form away from the clamps
    p, ep, m = 0.99, 0.5, 7
    k = np.log(1 - p) / np.log(1 - (1 - ep) ** m)
    assert oracle_mod.update_num_iters(p, ep, m, 100000) == int(np.rint(k))
    assert oracle_mod.update_num_iters(p, 0.0, m, 1000) == 0
    assert oracle_mod.update_num_iters(p, 1.0, m, 1000) == 1000


@pytest.mark.parametrize("c", range(6))
def test_replay_golden_and_python_restatement(oracle_mod, c):
    g = load("replay.npz")
    cnts, nm, exp = g[f"counts{c}"], g[f"nmodels{c}"], tuple(int(v) for v in g[f"result{c}"])
    assert oracle_mod.ransac_replay(cnts, nm, 1000, 7, 0.9999, 500) == exp
    niters, max_good, best = 500, 0, (-1, -1)
    it = 0
    while it < niters and it < len(cnts):
        for k in range(nm[it]):
            if cnts[it, k] > max(max_good, 6):
                max_good, best = int(cnts[it, k]), (it, k)
                niters = oracle_mod.update_num_iters(0.9999, (1000 - max_good) / 1000, 7, niters)
        it += 1
    assert exp == (best[0], best[1], max_good, niters)


# ------------------------------------------------------------------------------------ error functions
def test_scoring_golden_and_numpy_formulas(oracle_mod):
    g = load("scoring.npz")
    p1, p2 = g["p1"], g["p2"]
    for thr in (0.1, 1.0):
        cnt = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), thr)[0] for m in g["fund_models"]])
        assert np.array_equal(cnt, g[f"fund_counts_{thr}"])
    b = int(g["fund_best"][0])
    F = g["fund_models"][b].reshape(3, 3)
    x1 = np.c_[p1.astype(np.float64), np.ones(len(p1))]; x2 = np.c_[p2.astype(np.float64), np.ones(len(p2))]
    l2 = x1 @ F.T; l1 = x2 @ F
    d2 = (x2 * l2).sum(1) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
    d1 = (x1 * l1).sum(1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    assert np.allclose(np.maximum(d1, d2), g["fund_best_err"], rtol=1e-5)
    assert np.array_equal(g["fund_best_mask"], (g["fund_best_err"] <= np.float32(1.0)).astype(np.uint8))

    cnt = np.array([oracle_mod.find_inliers(oracle_mod.sampson_error(m, g["x1"], g["x2"]), float(g["ess_thr"][0]))[0]
                    for m in g["ess_models"]])
    assert np.array_equal(cnt, g["ess_counts"])
    for thr in (1.0, 5.0):
        cnt = np.array([oracle_mod.find_inliers(oracle_mod.pnp_error(m[:9], m[9:], g["Kp"], g["obj"], g["img"]), thr)[0]
                        for m in g["pnp_models"]])
        assert np.array_equal(cnt, g[f"pnp_counts_{thr}"])
    m = g["pnp_models"][0]
    X = g["obj"].astype(np.float64) @ m[:9].reshape(3, 3).T + m[9:]
    uv = X[:, :2] / X[:, 2:3] * [g["Kp"][0, 0], g["Kp"][1, 1]] + [g["Kp"][0, 2], g["Kp"][1, 2]]
    assert np.allclose(((uv - g["img"]) ** 2).sum(1), oracle_mod.pnp_error(m[:9], m[9:], g["Kp"], g["obj"], g["img"]), rtol=1e-3, atol=1e-3)


def test_find_inliers_threshold_rounding(oracle_mod):
    """t = (float)(thr*thr): the product is formed in f64 and rounded once."""
    thr = 0.1
    t = np.float32(thr * thr)
    err = np.array([t, np.nextafter(t, np.float32(1)), np.nextafter(t, np.float32(0))], np.float32)
    good, mask = oracle_mod.find_inliers(err, thr)
    assert good == 2 and list(mask) == [1, 0, 1]


# ------------------------------------------------------------------------------------ solvers
def test_7point_identities(oracle_mod):
    sc = synth.two_view_scene(200, seed=5, noise_px=0.5, outlier_frac=0.0)
    subs = oracle_mod.gen_subsets(200, 7, 50, sc["p1"], sc["p2"])
    n_models = 0
    for s in subs:
        for F in oracle_mod.run7point(sc["p1"][s], sc["p2"][s]):
            n_models += 1
            Fn = F / np.linalg.norm(F)
            assert abs(np.linalg.det(Fn)) < 1e-9
            x1 = np.c_[sc["p1"][s].astype(np.float64), np.ones(7)]; x2 = np.c_[sc["p2"][s].astype(np.float64), np.ones(7)]
            res = np.abs(np.einsum("ni,ij,nj->n", x2, Fn, x1))
            assert res.max() < 1e-6 * max(1.0, np.abs(x1).max() * np.abs(x2).max())
    assert n_models >= 50


def test_7point_recovers_noise_free_F(oracle_mod):
    sc = synth.two_view_scene(50, seed=6, noise_px=0.0, outlier_frac=0.0)
    K, R, t = sc["K"], sc["R"], sc["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ft = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    Ft /= Ft[2, 2]
    best = min(np.abs(F / F[2, 2] - Ft).max() / np.abs(Ft).max() for F in oracle_mod.run7point(sc["p1"][:7], sc["p2"][:7]))
    assert best < 1e-3            # f32 pixel coordinates limit the accuracy


def test_5point_identities_and_recovery(oracle_mod):
    sc = synth.two_view_scene(100, seed=7, noise_px=0.0, outlier_frac=0.0)
    Kinv = np.linalg.inv(sc["K"])
    x1 = (np.c_[sc["p1"].astype(np.float64), np.ones(100)] @ Kinv.T)[:, :2]
    x2 = (np.c_[sc["p2"].astype(np.float64), np.ones(100)] @ Kinv.T)[:, :2]
    t = sc["t"]
    Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ sc["R"]
    Et /= np.linalg.norm(Et)
    found = 0
    for s in oracle_mod.gen_subsets(100, 5, 20):
        Es = oracle_mod.run5point(x1[s], x2[s])
        assert 1 <= len(Es) <= 10
        for E in Es:
            En = E / np.linalg.norm(E)
            assert abs(np.linalg.det(En)) < 1e-8
            assert np.abs(2 * En @ En.T @ En - np.trace(En @ En.T) * En).max() < 1e-7
            h1 = np.c_[x1[s], np.ones(5)]; h2 = np.c_[x2[s], np.ones(5)]
            assert np.abs(np.einsum("ni,ij,nj->n", h2, En, h1)).max() < 1e-8
        found += min(min(np.abs(E / np.linalg.norm(E) - Et).max(), np.abs(E / np.linalg.norm(E) + Et).max()) for E in Es) < 1e-4
    assert found >= 18


def test_p3p_reprojects_its_sample_and_recovers_pose(oracle_mod):
    ps = synth.pnp_scene(200, seed=8, noise_px=0.0, outlier_frac=0.0)
    obj, img, K = ps["obj"], ps["img"], ps["K"]
    good = 0
    for s in oracle_mod.gen_subsets(200, 4, 40):
        ns, rvec, tvec, R = oracle_mod.p3p_kernel(obj[s], img[s], K)
        if ns == 0:
            continue
        X = obj[s].astype(np.float64) @ R.T + tvec
        uv = X[:, :2] / X[:, 2:3] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
        assert np.abs(uv[:3] - img[s][:3]).max() < 1e-2
        assert np.allclose(oracle_mod.rodrigues_v2m(rvec), R, atol=1e-12)
        good += np.abs(R - ps["R"]).max() < 1e-3 and np.abs(tvec - ps["t"]).max() < 1e-2
    assert good >= 30


def test_det_math_close_to_libm(oracle_mod):
    """the deterministic elementary functions (used so the GPU solvers can be bit-identical)
    change solver outputs by rounding noise only (amplified by the conditioning of the cubic's
    roots: observed up to ~1e-9 relative, bounded here at 1e-6)."""
    sc = synth.two_view_scene(300, seed=9)
    subs = oracle_mod.gen_subsets(300, 7, 60, sc["p1"], sc["p2"])
    a = [oracle_mod.run7point(sc["p1"][s], sc["p2"][s]) for s in subs]
    oracle_mod.set_det_math(True)
    try:
        b = [oracle_mod.run7point(sc["p1"][s], sc["p2"][s]) for s in subs]
    finally:
        oracle_mod.set_det_math(False)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        if len(x):
            assert np.abs(x - y).max() <= 1e-6 * max(1.0, np.abs(x).max())


def test_end_to_end_golden(oracle_mod):
    g = load("end_to_end.npz")
    oracle_mod.set_det_math(True)
    try:
        thr, conf, it = g["f_args"]
        F, mask, info = oracle_mod.find_fundamental_mat(g["f_p1"], g["f_p2"], thr, conf, int(it))
        assert np.array_equal(F, g["f_F"]) and np.array_equal(mask, g["f_mask"]) and np.array_equal(info, g["f_info"])
        prob, thr, it = g["e_args"]
        E, emask, _ = oracle_mod.find_essential_mat(g["f_p1"], g["f_p2"], g["e_K"], prob, thr, int(it))
        assert np.array_equal(E, g["e_E"]) and np.array_equal(emask, g["e_mask"])
        it, thr, conf = g["p_args"]
        ok, rvec, tvec, inl, _ = oracle_mod.solve_pnp_ransac(g["p_obj"], g["p_img"], g["p_K"], int(it), thr, conf)
        assert ok and np.array_equal(inl, g["p_inliers"]) and np.array_equal(rvec, g["p_rvec"]) and np.array_equal(tvec, g["p_tvec"])
    finally:
        oracle_mod.set_det_math(False)
    # decisions are the same with libm, values agree to rounding noise
    assert np.array_equal(g["f_mask"], g["f_mask_libm"]) and np.array_equal(g["p_inliers"], g["p_inliers_libm"])
    assert np.allclose(g["f_F"], g["f_F_libm"], rtol=1e-9, atol=1e-12)
    assert np.allclose(g["p_rvec"], g["p_rvec_libm"], rtol=1e-9) and np.allclose(g["p_tvec"], g["p_tvec_libm"], rtol=1e-9)
    # and the answer is right
    R = oracle_mod.rodrigues_v2m(g["p_rvec"])
    assert np.abs(R - g["p_R_true"]).max() < 5e-3 and np.abs(g["p_tvec"].ravel() - g["p_t_true"]).max() < 5e-2
    assert g["f_mask"].sum() > 300


def test_small_counts(oracle_mod):
    sc = synth.two_view_scene(20, seed=3)
    F, mask, _ = oracle_mod.find_fundamental_mat(sc["p1"][:6], sc["p2"][:6], 1.0, 0.99, 10)
    assert F is None


def test_oracle_knn_on_hashed_4096_golden(oracle_mod):
    """The 4096 x 4096 fixture whose inputs come from the integer-hash generator (no numpy Generator stream, so it
    never skips): the oracle reproduces the stored table, and an independent exact-integer numpy kNN agrees."""
    g = load("knn.npz")
    q, t = synth.hashed_descriptor_pair(4096, 4096, seed=int(g["seedh4096"][0]))
    w = np.arange(128) + 1
    assert [int((q.astype(np.int64) * w).sum()), int((t.astype(np.int64) * w).sum())] == [int(v) for v in g["checksumh4096"]]
    idx, dist = oracle_mod.knn2(q, t, mt=True)
    assert np.array_equal(idx, g["idxh4096"]) and np.array_equal(dist, g["disth4096"])
    qi, ti = q[:512].astype(np.int64), t.astype(np.int64)
    D = (qi ** 2).sum(1)[:, None] + (ti ** 2).sum(1)[None, :] - 2 * qi @ ti.T
    key = D * 8192 + np.arange(4096)[None, :]                  # ties -> lower train index
    order = np.argsort(key, axis=1)[:, :2]
    assert np.array_equal(order, idx[:512])
    assert np.array_equal(np.sqrt(np.take_along_axis(D, order, 1).astype(np.float32)), dist[:512])


# ------------------------------------------------------------------------------------ reference helpers
def test_pose_tail_against_reference_helpers(oracle_mod):
    from visual_odometry_amd.utils.decompose_essential_matrix import decomposeEssentialMatrix
    from visual_odometry_amd.utils.utils import hom_inv
    g = load("ref_helpers.npz")
    for E, R, u3, T, Tinv in zip(g["E"], g["R"], g["u3"], g["T"], g["Tinv"]):
        for fn in (decomposeEssentialMatrix, oracle_mod.decompose_essential_matrix):
            r, u = fn(E.copy())
            assert np.allclose(r, R, atol=1e-12) and np.allclose(u, u3, atol=1e-12)
        assert np.allclose(hom_inv(T), Tinv, atol=1e-15) and np.allclose(oracle_mod.hom_inv(T), Tinv, atol=1e-15)
        assert np.allclose(hom_inv(T) @ T, np.eye(4), atol=1e-12)


def test_triangulation_against_reference_linear_triangulation(oracle_mod):
    """The reference's own numpy DLT (`src/utils/linear_triangulation.py`, run in the build container by
    make_golden.py) and the restatement of cv.triangulatePoints solve the same homogeneous system on
    noise-free data (exact f64 projections): the dehomogenised points agree to 1e-9 relative."""
    g = load("ref_helpers.npz")
    X = oracle_mod.triangulate_points(g["tri_M1"], g["tri_M2"], g["tri_p1"].T, g["tri_p2"].T)
    P = X / X[3]
    scale = np.linalg.norm(g["tri_P"][:3], axis=0)
    assert (np.linalg.norm(P[:3] - g["tri_P"][:3], axis=0) / scale).max() < 1e-9


def test_oracle_lmeds_small_point_sets(oracle_mod):
    """8..14 points go through the LMedS restatement: on noise-free data the winner has a vanishing
    median, (nearly) every point is an inlier of the sigma floor (0.001 px: f32-rounded coordinates through a
    minimal solve can miss it) and F satisfies the epipolar constraint; the
    iteration count is the fixed one (outlier ratio 0.45), never the RANSAC one."""
    for n in (8, 10, 14):
        sc = synth.two_view_scene(n, seed=60 + n, noise_px=0.0, outlier_frac=0.0)
        F, mask, info = oracle_mod.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.99, 500)
        assert F is not None and mask.sum() >= 7
        assert info[3] == max(oracle_mod.update_num_iters(0.99, 0.45, 7, 500), 3)
        x1 = np.c_[sc["p1"].astype(np.float64), np.ones(n)]; x2 = np.c_[sc["p2"].astype(np.float64), np.ones(n)]
        res = np.abs(np.einsum("ni,ij,nj->n", x2, F, x1)) / np.linalg.norm(F)
        # (every root of a 7-point sample fits its 7 points, so the first model already has a zero median;
        # it need not be the scene's F, only a majority of the points must lie on it)
        assert np.sort(res)[n // 2] < 1e-3 and abs(np.linalg.det(F)) < 1e-9 * np.abs(F).max() ** 3 + 1e-15
    # 15 points: back on the RANSAC path (info[3] then is the adaptive count)
    sc = synth.two_view_scene(15, seed=5, noise_px=0.0, outlier_frac=0.0)
    F, mask, info = oracle_mod.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.99, 500)
    assert F is not None and info[3] < 500


def test_gen_subsets_attempt_limits_equal_oracle(oracle_mod):
    """getSubset's maxAttempts: 10000 in the RANSAC registrator, 1000 in the LMedS one (8..14 points).  Host helper of
    the library against the oracle on point sets whose last-point collinearity check rejects most or all samples."""
    import ctypes as C
    from visual_odometry_amd import ops
    rng = np.random.default_rng(3)
    lib = oracle_mod.lib()
    for n, frac_line in ((12, 1.0), (12, 0.75), (40, 0.9), (10, 0.0)):
        p = rng.uniform(0, 500, (n, 4)).astype(np.float32)
        k = int(frac_line * n)
        p[:k, 1] = 2.0 * p[:k, 0] + 1.0                    # first image: k points on one line
        for attempts in (1000, 10000, 3):
            got = ops.gen_subsets(n, 7, 60, p, max_attempts=attempts)
            exp = np.zeros((60, 7), np.int32)
            a, b = np.ascontiguousarray(p[:, :2]), np.ascontiguousarray(p[:, 2:])
            m = lib.vo_or_gen_subsets_attempts(n, 7, 60, 1, a.ctypes.data, b.ctypes.data, exp, attempts)
            assert len(got) == m and np.array_equal(got, exp[:m]), (n, frac_line, attempts)
    assert len(ops.gen_subsets(12, 7, 60, np.tile(np.arange(12, dtype=np.float32)[:, None], (1, 4)))) == 0
