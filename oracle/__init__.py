"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes front end of ``oracle/_build/libvo_oracle.so`` (plain-C restatement of the OpenCV 4.6.0
routines the reference's hot path calls; see ``oracle/vo_oracle.h``) plus numpy restatements of the
reference's own O(N) Python glue.  **Parity unpinned**: OpenCV is not available in this environment
and the reference has no tests or golden vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``visual_odometry_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_c_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_c_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_c_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_c_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_c_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile both oracle libraries with gcc (no GPU, no reference sources involved)."""
    out = os.path.join(_HERE, "_build", "libvo_oracle.so")
    if force or not os.path.exists(out):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _sig(lib):
    lib.vo_or_set_det_math.argtypes = [C.c_int]
    lib.vo_or_get_det_math.restype = C.c_int
    lib.vo_or_knn2.argtypes = [_c_f32p, C.c_int, _c_f32p, C.c_int, C.c_int, _c_i32p, _c_f32p]
    lib.vo_or_rng_stream.argtypes = [C.c_uint64, C.c_int, _c_u32p]
    lib.vo_or_gen_subsets.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _c_i32p]
    lib.vo_or_gen_subsets.restype = C.c_int
    lib.vo_or_gen_subsets_attempts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _c_i32p, C.c_int]
    lib.vo_or_gen_subsets_attempts.restype = C.c_int
    lib.vo_or_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    lib.vo_or_update_num_iters.restype = C.c_int
    lib.vo_or_ransac_replay.argtypes = [_c_i32p, _c_i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_int, _c_i32p]
    lib.vo_or_find_inliers.argtypes = [_c_f32p, C.c_int, C.c_double, _c_u8p]
    lib.vo_or_find_inliers.restype = C.c_int
    lib.vo_or_fund_error.argtypes = [_c_f64p, _c_f32p, _c_f32p, C.c_int, _c_f32p]
    lib.vo_or_fund_check_subset.argtypes = [_c_f32p, _c_f32p, C.c_int]
    lib.vo_or_fund_check_subset.restype = C.c_int
    lib.vo_or_run7point.argtypes = [_c_f32p, _c_f32p, _c_f64p, C.c_int]
    lib.vo_or_run7point.restype = C.c_int
    lib.vo_or_find_fundamental_ransac.argtypes = [_c_f32p, _c_f32p, C.c_int, C.c_double, C.c_double,
                                                  C.c_int, C.c_int, _c_f64p, _c_u8p, _c_i32p]
    lib.vo_or_find_fundamental_ransac.restype = C.c_int
    lib.vo_or_find_fundamental_lmeds.argtypes = [_c_f32p, _c_f32p, C.c_int, C.c_double, C.c_int, C.c_int,
                                                 _c_f64p, _c_u8p, _c_i32p]
    lib.vo_or_find_fundamental_lmeds.restype = C.c_int
    lib.vo_or_sampson_error.argtypes = [_c_f64p, _c_f64p, _c_f64p, C.c_int, _c_f32p]
    lib.vo_or_run5point.argtypes = [_c_f64p, _c_f64p, _c_f64p]
    lib.vo_or_run5point.restype = C.c_int
    lib.vo_or_find_essential_ransac.argtypes = [_c_f64p, _c_f64p, C.c_int, _c_f64p, C.c_double,
                                                C.c_double, C.c_int, _c_f64p, _c_u8p, _c_i32p]
    lib.vo_or_find_essential_ransac.restype = C.c_int
    lib.vo_or_rodrigues_v2m.argtypes = [_c_f64p, _c_f64p]
    lib.vo_or_rodrigues_m2v.argtypes = [_c_f64p, _c_f64p]
    lib.vo_or_project_points.argtypes = [_c_f64p, _c_f64p, _c_f64p, _c_f32p, C.c_int, _c_f64p]
    lib.vo_or_pnp_error.argtypes = [_c_f64p, _c_f64p, _c_f64p, _c_f32p, _c_f32p, C.c_int, _c_f32p]
    lib.vo_or_p3p_kernel.argtypes = [_c_f32p, _c_f32p, _c_f64p, _c_f64p, _c_f64p, _c_f64p]
    lib.vo_or_p3p_kernel.restype = C.c_int
    lib.vo_or_epnp.argtypes = [_c_f64p, _c_f64p, C.c_int, _c_f64p, _c_f64p, _c_f64p]
    lib.vo_or_epnp.restype = C.c_int
    lib.vo_or_solve_pnp_ransac.argtypes = [_c_f32p, _c_f32p, C.c_int, _c_f64p, C.c_int, C.c_float,
                                           C.c_double, _c_f64p, _c_f64p, _c_u8p, _c_i32p]
    lib.vo_or_solve_pnp_ransac.restype = C.c_int
    lib.vo_or_triangulate.argtypes = [_c_f64p, _c_f64p, _c_f64p, _c_f64p, C.c_int, _c_f64p]
    lib.vo_or_jacobi_svd.argtypes = [_c_f64p, C.c_int, C.c_int, C.c_int, _c_f64p, _c_f64p]
    lib.vo_or_solve_cubic.argtypes = [_c_f64p, _c_f64p]
    lib.vo_or_solve_cubic.restype = C.c_int
    lib.vo_or_solve_deg4.argtypes = [C.c_double] * 5 + [_c_f64p]
    lib.vo_or_solve_deg4.restype = C.c_int
    lib.vo_or_solve_poly.argtypes = [_c_f64p, C.c_int, _c_f64p, _c_f64p]
    lib.vo_or_solve_poly.restype = C.c_int
    return lib


def lib(mt: bool = False):
    key = "mt" if mt else "st"
    if key not in _LIBS:
        name = "libvo_oracle_mt.so" if mt else "libvo_oracle.so"
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path):
            build()
        _LIBS[key] = _sig(C.CDLL(path))
    return _LIBS[key]


def set_det_math(on: bool) -> None:
    lib().vo_or_set_det_math(1 if on else 0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def K4_of(K) -> np.ndarray:
    K = np.asarray(K, dtype=np.float64)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dtype=np.float64)


# ---------------------------------------------------------------------------- A.1 matching
def knn2(des_q, des_t, mt: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """BFMatcher(NORM_L2).knnMatch(q, t, k=2): (idx (Nq,2) i32, dist (Nq,2) f32)."""
    q, t = _f32(des_q), _f32(des_t)
    nq, nt = q.shape[0], t.shape[0]
    dim = q.shape[1] if q.ndim == 2 else 0
    idx = np.full((nq, 2), -1, dtype=np.int32)
    dist = np.full((nq, 2), np.finfo(np.float32).max, dtype=np.float32)
    if nq and nt:
        lib(mt).vo_or_knn2(q, nq, t, nt, dim, idx, dist)
    return idx, dist


def ratio_test(dist: np.ndarray, threshold: float) -> np.ndarray:
    """reference src/features.py:104-106: ``m.distance < T * n.distance`` (f32 -> Python float)."""
    d = dist.astype(np.float64)
    return d[:, 0] < float(threshold) * d[:, 1]


def match_features_indices(des_i, des_j, threshold: float):
    """Indices kept by reference match_features (src/features.py:95-124), ascending queryIdx."""
    idx, dist = knn2(des_i, des_j)
    keep = ratio_test(dist, threshold)
    qi = np.nonzero(keep)[0].astype(np.int32)
    return qi, idx[keep, 0].astype(np.int32), idx, dist


# ---------------------------------------------------------------------------- A.2 RANSAC bits
def rng_stream(n: int, seed: int = 0xFFFFFFFFFFFFFFFF) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().vo_or_rng_stream(seed, n, out)
    return out


def gen_subsets(count: int, model_points: int, max_iters: int, m1=None, m2=None) -> np.ndarray:
    subsets = np.zeros((max(max_iters, 1), model_points), dtype=np.int32)
    if m1 is not None:
        a, b = _f32(m1), _f32(m2)
        n = lib().vo_or_gen_subsets(count, model_points, max_iters, 1, a.ctypes.data, b.ctypes.data, subsets)
    else:
        n = lib().vo_or_gen_subsets(count, model_points, max_iters, 0, None, None, subsets)
    return subsets[:n]


def update_num_iters(p, ep, model_points, max_iters) -> int:
    return lib().vo_or_update_num_iters(p, ep, model_points, max_iters)


def ransac_replay(counts, nmodels, count, model_points, confidence, max_iters):
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    if counts.ndim == 1:
        counts = counts[:, None]
    nmodels = np.ascontiguousarray(nmodels, dtype=np.int32)
    out = np.zeros(4, dtype=np.int32)
    lib().vo_or_ransac_replay(counts, nmodels, counts.shape[0], counts.shape[1], count, model_points,
                              confidence, max_iters, out)
    return tuple(int(v) for v in out)     # best_iter, best_k, max_good, niters


def find_inliers(err, thresh):
    err = _f32(err)
    mask = np.zeros(err.shape[0], dtype=np.uint8)
    good = lib().vo_or_find_inliers(err, err.shape[0], float(thresh), mask)
    return good, mask


# ---------------------------------------------------------------------------- A.3 fundamental
def fund_error(F, m1, m2) -> np.ndarray:
    m1, m2 = _f32(m1), _f32(m2)
    err = np.zeros(m1.shape[0], dtype=np.float32)
    lib().vo_or_fund_error(_f64(F).reshape(9), m1, m2, m1.shape[0], err)
    return err


def run7point(ms1, ms2, hartley: bool = False) -> np.ndarray:
    out = np.zeros(27, dtype=np.float64)
    n = lib().vo_or_run7point(_f32(ms1), _f32(ms2), out, int(hartley))
    return out.reshape(3, 3, 3)[: max(n, 0)]


def find_fundamental_mat(points1, points2, thr, conf, max_iters, hartley: bool = False):
    """cv2.findFundamentalMat(p1,p2,cv2.RANSAC,thr,conf,maxIters) -> (F|None, mask (N,1) u8, info)."""
    m1, m2 = _f32(points1).reshape(-1, 2), _f32(points2).reshape(-1, 2)
    n = m1.shape[0]
    F = np.zeros(9, dtype=np.float64)
    mask = np.zeros(n, dtype=np.uint8)
    info = np.zeros(4, dtype=np.int32)
    if 8 <= n <= 14:                  # OpenCV 4.6.0 switches to LMedS below 15 points (A.3)
        ok = lib().vo_or_find_fundamental_lmeds(m1, m2, n, conf, max_iters, int(hartley), F, mask, info)
    else:
        ok = lib().vo_or_find_fundamental_ransac(m1, m2, n, thr, conf, max_iters, int(hartley), F, mask, info)
    return (F.reshape(3, 3) if ok else None), mask.reshape(-1, 1), info


# ---------------------------------------------------------------------------- A.4 essential
def sampson_error(E, x1, x2) -> np.ndarray:
    x1, x2 = _f64(x1), _f64(x2)
    err = np.zeros(x1.shape[0], dtype=np.float32)
    lib().vo_or_sampson_error(_f64(E).reshape(9), x1, x2, x1.shape[0], err)
    return err


def run5point(q1, q2) -> np.ndarray:
    out = np.zeros(90, dtype=np.float64)
    n = lib().vo_or_run5point(_f64(q1), _f64(q2), out)
    return out.reshape(10, 3, 3)[:n]


def find_essential_mat(points1, points2, K, prob, thr, max_iters):
    p1, p2 = _f64(points1).reshape(-1, 2), _f64(points2).reshape(-1, 2)
    n = p1.shape[0]
    E = np.zeros(9, dtype=np.float64)
    mask = np.zeros(n, dtype=np.uint8)
    info = np.zeros(4, dtype=np.int32)
    ok = lib().vo_or_find_essential_ransac(p1, p2, n, _f64(K).reshape(9), prob, thr, max_iters, E, mask, info)
    return (E.reshape(3, 3) if ok else None), mask.reshape(-1, 1), info


# ---------------------------------------------------------------------------- A.5/A.6 PnP
def rodrigues_v2m(rvec) -> np.ndarray:
    R = np.zeros(9, dtype=np.float64)
    lib().vo_or_rodrigues_v2m(_f64(rvec).reshape(3), R)
    return R.reshape(3, 3)


def rodrigues_m2v(R) -> np.ndarray:
    r = np.zeros(3, dtype=np.float64)
    lib().vo_or_rodrigues_m2v(_f64(R).reshape(9), r)
    return r


def pnp_error(R, t, K, obj, img) -> np.ndarray:
    obj, img = _f32(obj), _f32(img)
    err = np.zeros(obj.shape[0], dtype=np.float32)
    lib().vo_or_pnp_error(_f64(R).reshape(9), _f64(t).reshape(3), K4_of(K), obj, img, obj.shape[0], err)
    return err


def p3p_kernel(obj4, img4, K):
    rvec, tvec, R = np.zeros(3), np.zeros(3), np.zeros(9)
    ns = lib().vo_or_p3p_kernel(_f32(obj4), _f32(img4), K4_of(K), rvec, tvec, R)
    return ns, rvec, tvec, R.reshape(3, 3)


def epnp(obj, img, K):
    obj, img = _f64(obj), _f64(img)
    R, t = np.zeros(9), np.zeros(3)
    ok = lib().vo_or_epnp(obj, img, obj.shape[0], K4_of(K), R, t)
    return ok, R.reshape(3, 3), t


def solve_pnp_ransac(object_points, image_points, K, iterations_count, reprojection_error, confidence):
    """cv.solvePnPRansac(..., flags=SOLVEPNP_P3P, distCoeffs=0) -> (ok, rvec (3,1), tvec (3,1), inliers|None, info)."""
    obj, img = _f32(object_points).reshape(-1, 3), _f32(image_points).reshape(-1, 2)
    n = obj.shape[0]
    rvec, tvec = np.zeros(3), np.zeros(3)
    mask = np.zeros(n, dtype=np.uint8)
    info = np.zeros(4, dtype=np.int32)
    rc = lib().vo_or_solve_pnp_ransac(obj, img, n, K4_of(K), iterations_count, float(reprojection_error),
                                      confidence, rvec, tvec, mask, info)
    ok = rc > 0
    inliers = np.nonzero(mask)[0].astype(np.int32).reshape(-1, 1) if ok else None
    return ok, rvec.reshape(3, 1), tvec.reshape(3, 1), inliers, info


def triangulate_points(P1, P2, x1, x2) -> np.ndarray:
    """cv.triangulatePoints(P1,P2,x1 (2,N),x2 (2,N)) -> (4,N) homogeneous, un-normalised."""
    a, b = _f64(np.asarray(x1).T), _f64(np.asarray(x2).T)
    out = np.zeros((a.shape[0], 4), dtype=np.float64)
    lib().vo_or_triangulate(_f64(P1).reshape(12), _f64(P2).reshape(12), a, b, a.shape[0], out)
    return out.T.copy()


# ---------------------------------------------------------------------------- reference glue (numpy)
def hom_inv(T):
    """reference src/utils/utils.py:100-106."""
    T = np.asarray(T, dtype=np.float64)
    I = np.zeros_like(T)
    I[0:3, 0:3] = T[0:3, 0:3].T
    I[0:3, 3] = -T[0:3, 0:3].T @ T[0:3, 3]
    I[3, 3] = 1
    return I


def decompose_essential_matrix(E):
    """reference src/utils/decompose_essential_matrix.py:4-37."""
    u, _, vh = np.linalg.svd(E)
    u3 = u[:, 2].copy()
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float64)
    R = np.zeros((3, 3, 2))
    R[:, :, 0] = u @ W @ vh
    R[:, :, 1] = u @ W.T @ vh
    for i in range(2):
        if np.linalg.det(R[:, :, i]) < 0:
            R[:, :, i] *= -1
    if np.linalg.norm(u3) != 0:
        u3 /= np.linalg.norm(u3)
    return R, u3


def disambiguate_relative_pose(Rots, u3, p0_h, p1_h, K1, K2, max_depth: float = 50.0):
    """reference src/utils/disambiguate_relative_pose.py:5-72 (cv.triangulatePoints -> oracle)."""
    M1 = K1 @ np.eye(3, 4)
    best = -np.inf
    R = T = P_cami = None
    for iRot in range(2):
        R_t = Rots[:, :, iRot]
        for iSign in range(2):
            T_t = u3 * (-1) ** iSign
            M2 = K2 @ np.c_[R_t, T_t]
            P_C1 = triangulate_points(M1, M2, p0_h[:2, :], p1_h[:2, :])
            P_C1 = P_C1 / P_C1[3, :]
            P_C2 = np.c_[R_t, T_t] @ P_C1
            total = np.sum(P_C1[2, :] > 0) + np.sum(P_C2[2, :] > 0)
            if total > best:
                R, T, P_cami, best = R_t, T_t, P_C1, total
    T_cami_camj = np.eye(4)
    T_cami_camj[:3, :3] = R
    T_cami_camj[:3, 3] = T.ravel()
    min_depth = min(0, T_cami_camj[2, 2])
    mask = np.logical_and(P_cami[2, :] > min_depth, np.abs(np.linalg.norm(P_cami, axis=0)) < max_depth)
    return T_cami_camj, P_cami[:, mask].T, mask


def pnp(points_2d, points_3d, K, iterations, threshold, confidence):
    """reference src/pnp.py:13-41 -> (cam_to_world 4x4, inlier index array | None)."""
    ok, rvec, tvec, inl, _ = solve_pnp_ransac(points_3d, points_2d, K, iterations, threshold, confidence)
    R = rodrigues_v2m(rvec)
    M = np.vstack([np.hstack([R, tvec.reshape(3, 1)]), np.array([[0, 0, 0, 1.0]])])
    return hom_inv(M), inl
