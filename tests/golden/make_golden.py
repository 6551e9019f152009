#!/usr/bin/env python3
"""Generator of the committed golden vectors in tests/golden/*.npz.

The reference ships no tests, fixtures or seeds, and the library that holds its hot-path arithmetic
(OpenCV 4.6.0) is absent from this environment, so most vectors below are outputs of THIS
repository's CPU oracle on seeded inputs: they pin the oracle (and through it the GPU path) against
regressions, not against real OpenCV -- parity unpinned.  The exception is `ref_helpers.npz`: its
expected outputs come from the reference's own pure-numpy helpers
(src/utils/decompose_essential_matrix.py, src/utils/utils.py:hom_inv), imported from
/root/reference in the build container (they need no cv2).

Run from the repo root:  python tests/golden/make_golden.py      (needs /root/reference only for
ref_helpers.npz; skipped with a message if it is absent).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle as o  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print(f"{name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items()))


def knn_cases():
    out = {}
    # 64 x 64, hand-checkable structure: exact duplicates (tie -> lower index), a zero row,
    # duplicates inside one 32-row tile
    q, t = synth.sift_like_pair(64, 64, seed=101)
    t[5] = t[3]; t[40] = t[3]; q[7] = t[3]; q[9] = 0; t[33] = t[2]; t[34] = t[2]; q[11] = t[2]
    idx, dist = o.knn2(q, t)
    out.update(q64=q.astype(np.uint8), t64=t.astype(np.uint8), idx64=idx, dist64=dist)
    q, t = synth.sift_like_pair(1024, 1024, seed=102)
    idx, dist = o.knn2(q, t)
    out.update(q1024=q.astype(np.uint8), t1024=t.astype(np.uint8), idx1024=idx, dist1024=dist)
    for T in (0.6, 0.8):
        out[f"keep1024_{int(T * 10)}"] = o.ratio_test(dist, T)
    # 4096 x 4096: inputs regenerated from the seed, outputs stored
    q, t = synth.sift_like_pair(4096, 4096, seed=103)
    idx, dist = o.knn2(q, t, mt=True)
    out.update(idx4096=idx, dist4096=dist, seed4096=np.array([103]),
               checksum4096=np.array([q.sum(dtype=np.float64), t.sum(dtype=np.float64)]))
    # 4096 x 4096 from the integer-hash generator (same bytes on every numpy): inputs regenerated, never skipped
    q, t = synth.hashed_descriptor_pair(4096, 4096, seed=7)
    idx, dist = o.knn2(q, t, mt=True)
    out.update(idxh4096=idx, disth4096=dist, seedh4096=np.array([7]),
               checksumh4096=np.array([int((q.astype(np.int64) * (np.arange(128) + 1)).sum()), int((t.astype(np.int64) * (np.arange(128) + 1)).sum())]))
    save("knn.npz", **out)


def rng_cases():
    out = {"stream64": o.rng_stream(64)}
    for count in (15, 100, 8192):
        for m in (4, 5, 7):
            out[f"subsets_{count}_{m}"] = o.gen_subsets(count, m, 64)
    sc = synth.two_view_scene(100, seed=7)
    out["fund_p1"], out["fund_p2"] = sc["p1"], sc["p2"]
    out["subsets_fund_100"] = o.gen_subsets(100, 7, 64, sc["p1"], sc["p2"])
    cases = [(0.99, 0.5, 7, 1000), (0.9999, 0.3, 4, 10000), (0.9999, 0.9, 7, 100), (0.999, 0.0, 5, 50),
             (0.999, 1.0, 5, 50), (0.9999, 0.65, 7, 4096), (0.5, 0.5, 4, 65536)]
    out["update_cases"] = np.array(cases, dtype=np.float64)
    out["update_expected"] = np.array([o.update_num_iters(p, ep, int(m), int(mi)) for p, ep, m, mi in cases], dtype=np.int32)
    save("rng.npz", **out)


def scoring_cases():
    o.set_det_math(False)
    sc = synth.two_view_scene(1000, seed=201)
    p1, p2, K = sc["p1"], sc["p2"], sc["K"]
    subs = o.gen_subsets(1000, 7, 200, p1, p2)
    Fm = np.array([F.reshape(9) for s in subs for F in o.run7point(p1[s], p2[s])])[:256]
    out = dict(p1=p1, p2=p2, K=K, fund_models=Fm)
    for thr in (0.1, 1.0):
        errs = [o.fund_error(m, p1, p2) for m in Fm]
        out[f"fund_counts_{thr}"] = np.array([o.find_inliers(e, thr)[0] for e in errs], dtype=np.int32)
    b = int(np.argmax(out["fund_counts_1.0"]))
    out["fund_best"] = np.array([b])
    out["fund_best_mask"] = o.find_inliers(o.fund_error(Fm[b], p1, p2), 1.0)[1]
    out["fund_best_err"] = o.fund_error(Fm[b], p1, p2)

    Kinv = np.linalg.inv(K)
    x1 = (np.c_[p1.astype(np.float64), np.ones(len(p1))] @ Kinv.T)[:, :2]
    x2 = (np.c_[p2.astype(np.float64), np.ones(len(p2))] @ Kinv.T)[:, :2]
    subs = o.gen_subsets(1000, 5, 60)
    Em = np.array([E.reshape(9) for s in subs for E in o.run5point(x1[s], x2[s])])[:256]
    thr = 1.0 / K[0, 0]
    out.update(x1=x1, x2=x2, ess_models=Em, ess_thr=np.array([thr]),
               ess_counts=np.array([o.find_inliers(o.sampson_error(m, x1, x2), thr)[0] for m in Em], dtype=np.int32))

    ps = synth.pnp_scene(1000, seed=202)
    obj, img, Kp = ps["obj"], ps["img"], ps["K"]
    subs = o.gen_subsets(1000, 4, 300)
    pm = []
    for s in subs:
        ns, rv, tv, R = o.p3p_kernel(obj[s], img[s], Kp)
        if ns > 0:
            pm.append(np.concatenate([R.reshape(9), tv]))
    pm = np.array(pm)[:256]
    out.update(obj=obj, img=img, Kp=Kp, pnp_models=pm)
    for thr in (1.0, 5.0):
        out[f"pnp_counts_{thr}"] = np.array(
            [o.find_inliers(o.pnp_error(m[:9], m[9:], Kp, obj, img), thr)[0] for m in pm], dtype=np.int32)
    save("scoring.npz", **out)


def replay_cases():
    rng = np.random.default_rng(5)
    out = {}
    for c in range(6):
        n_it, mm = 500, 3
        cnts = rng.integers(0, 40, size=(n_it, mm)).astype(np.int32)
        nm = rng.integers(0, mm + 1, size=n_it).astype(np.int32)
        if c == 1:
            cnts[100, 2] = 600; nm[100] = 3; cnts[300, 0] = 900; nm[300] = 1
        if c == 2:
            cnts[:] = 5
        if c == 3:
            cnts[50:60] = 700; nm[50:60] = 3
        if c == 4:
            cnts[3, 1] = 990; nm[3] = 3; cnts[4, 0] = 995; nm[4] = 1     # niters collapses right after
        if c == 5:
            cnts[499, 2] = 800; nm[499] = 3
        out[f"counts{c}"], out[f"nmodels{c}"] = cnts, nm
        out[f"result{c}"] = np.array(o.ransac_replay(cnts, nm, 1000, 7, 0.9999, n_it), dtype=np.int32)
    save("replay.npz", **out)


def end_to_end_cases():
    """whole RANSAC calls in det-math mode (the mode the GPU solvers reproduce bit for bit)."""
    o.set_det_math(True)
    out = {}
    sc = synth.two_view_scene(800, seed=301)
    F, mask, info = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.999, 200)
    out.update(f_p1=sc["p1"], f_p2=sc["p2"], f_F=F, f_mask=mask, f_info=info, f_args=np.array([1.0, 0.999, 200]))
    E, emask, einfo = o.find_essential_mat(sc["p1"], sc["p2"], sc["K"], 0.999, 1.0, 150)
    out.update(e_K=sc["K"], e_E=E, e_mask=emask, e_info=einfo, e_args=np.array([0.999, 1.0, 150]))
    ps = synth.pnp_scene(800, seed=302)
    ok, rvec, tvec, inl, pinfo = o.solve_pnp_ransac(ps["obj"], ps["img"], ps["K"], 300, 2.0, 0.999)
    out.update(p_obj=ps["obj"], p_img=ps["img"], p_K=ps["K"], p_rvec=rvec, p_tvec=tvec, p_inliers=inl,
               p_info=pinfo, p_args=np.array([300, 2.0, 0.999]), p_R_true=ps["R"], p_t_true=ps["t"])
    o.set_det_math(False)
    # the same calls with libm elementary functions: decisions must not depend on the maths mode here
    F2, mask2, info2 = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.999, 200)
    ok2, rvec2, tvec2, inl2, pinfo2 = o.solve_pnp_ransac(ps["obj"], ps["img"], ps["K"], 300, 2.0, 0.999)
    out.update(f_F_libm=F2, f_mask_libm=mask2, p_inliers_libm=inl2, p_rvec_libm=rvec2, p_tvec_libm=tvec2)
    save("end_to_end.npz", **out)


def pose_cases():
    """the widened rows: cv.triangulatePoints, the disambiguation chain, the EPnP refit and the LMedS
    branch of findFundamentalMat, in det-math mode (hypot evaluated as sqrt(a*a+b*b), like the device)."""
    o.set_det_math(True)
    out = {}
    sc = synth.two_view_scene(300, seed=501, outlier_frac=0.1)
    K = sc["K"]
    P1, P2 = K @ np.eye(3, 4), K @ np.c_[sc["R"], sc["t"]]
    x1, x2 = sc["p1"].astype(np.float64), sc["p2"].astype(np.float64)
    out.update(tri_P1=P1, tri_P2=P2, tri_x1=x1, tri_x2=x2, tri_X=o.triangulate_points(P1, P2, x1.T, x2.T).T)
    F, mask, _ = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.999, 200)
    Rs, u3 = o.decompose_essential_matrix(K.T @ F @ K)
    sel = mask.ravel() == 1
    T, P, m = o.disambiguate_relative_pose(Rs, u3, np.c_[x1[sel], np.ones(sel.sum())].T, np.c_[x2[sel], np.ones(sel.sum())].T, K, K)
    out.update(tail_K=K, tail_F=F, tail_mask=mask, tail_p1=sc["p1"], tail_p2=sc["p2"], tail_T=T, tail_P=P, tail_keep=m)
    ps = synth.pnp_scene(500, seed=502, outlier_frac=0.0)
    ok, R, t = o.epnp(ps["obj"].astype(np.float64), ps["img"].astype(np.float64), ps["K"])
    out.update(epnp_obj=ps["obj"], epnp_img=ps["img"], epnp_K=ps["K"], epnp_R=R, epnp_t=t)
    for n in (9, 13):
        sc = synth.two_view_scene(n, seed=503 + n, outlier_frac=0.2)
        F, mask, info = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.99, 200)
        out.update({f"lmeds{n}_p1": sc["p1"], f"lmeds{n}_p2": sc["p2"], f"lmeds{n}_F": F if F is not None else np.zeros((0, 3)),
                    f"lmeds{n}_mask": mask, f"lmeds{n}_info": info})
    o.set_det_math(False)
    save("pose.npz", **out)


def ref_helper_cases():
    ref_src = "/root/reference/src"
    if not os.path.isdir(ref_src):
        print("ref_helpers.npz: /root/reference absent, keeping the committed file")
        return
    sys.path.insert(0, ref_src)
    from utils.decompose_essential_matrix import decomposeEssentialMatrix  # reference, pure numpy
    from utils.utils import hom_inv                                        # reference, pure numpy
    rng = np.random.default_rng(9)
    Es, Rs, u3s, Ts, Tinvs = [], [], [], [], []
    for k in range(8):
        sc = synth.two_view_scene(10, seed=400 + k)
        t = sc["t"]
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ sc["R"] * rng.uniform(0.5, 2.0)
        if k >= 6:
            E = rng.normal(size=(3, 3))            # not a valid essential matrix: still decomposes
        R, u3 = decomposeEssentialMatrix(E.copy())
        Es.append(E); Rs.append(R); u3s.append(u3)
        T = np.eye(4); T[:3, :3] = sc["R"]; T[:3, 3] = rng.normal(size=3) * 5
        Ts.append(T); Tinvs.append(hom_inv(T))
    # the reference's own numpy DLT triangulation (src/utils/linear_triangulation.py) on a noise-free scene:
    # same null vector as cv.triangulatePoints' 4-row system there, so it pins the triangulation of the
    # pose tail against reference code (dehomogenised, as the reference returns it)
    from utils.linear_triangulation import linearTriangulation                   # reference, pure numpy
    sc = synth.two_view_scene(40, K=synth.K_KITTI, size=synth.IMAGE_SIZE["kitti"], seed=808, noise_px=0.0, outlier_frac=0.0)
    Kk = synth.K_KITTI
    M1 = Kk @ np.hstack([np.eye(3), np.zeros((3, 1))])
    M2 = Kk @ np.hstack([sc["R"], sc["t"].reshape(3, 1)])
    proj = lambda M, X: (lambda q: (q[:2] / q[2]).T)(M @ np.vstack([X.T, np.ones((1, X.shape[0]))]))
    p1, p2 = proj(M1, sc["X"]), proj(M2, sc["X"])                                # exact f64 projections
    h = lambda p: np.vstack([p.T, np.ones((1, p.shape[0]))])
    tri_P = linearTriangulation(h(p1), h(p2), M1, M2)
    save("ref_helpers.npz", E=np.array(Es), R=np.array(Rs), u3=np.array(u3s), T=np.array(Ts), Tinv=np.array(Tinvs),
         tri_M1=M1, tri_M2=M2, tri_p1=p1, tri_p2=p2, tri_P=tri_P)


if __name__ == "__main__":
    o.build()
    if len(sys.argv) > 1 and sys.argv[1] == "pose":        # only the fixtures added later
        pose_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ref":         # only the fixtures made by importing reference helpers
        ref_helper_cases()
        sys.exit(0)
    knn_cases(); rng_cases(); scoring_cases(); replay_cases(); end_to_end_cases(); pose_cases(); ref_helper_cases()
