"""RANSAC on the MI355X as one batch per call: host-side sample stream -> device minimal solvers
-> device scoring of every hypothesis against every correspondence -> device replay of OpenCV's
sequential best-model bookkeeping -> device mask of the winner.

These are the engines behind the cv2-shaped entry points in ``cv2_shim`` and the reference-shaped
ones in ``utils.geometry`` / ``pnp`` (reference src/utils/geometry.py:37-45, src/pnp.py:13-41);
semantics follow OpenCV 4.6.0's ``RANSACPointSetRegistrator::run``.
The only host<->device traffic per call is: correspondences + sample indices up, 4 ints + one
model + one mask down.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .epnp import epnp_pose
from .parallel import sharded_ransac
from .rodrigues import rodrigues_matrix_to_vector

_DBL_EPS = float(np.finfo(np.float64).eps)


@dataclass
class RansacInfo:
    """What the replay decided (handy for tests and tracing; the reference never sees it)."""
    best_iter: int = -1
    best_model: int = -1
    max_good: int = 0
    niters: int = 0
    n_generated: int = 0


def _device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


def _h2d(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(_device(), non_blocking=False)


# ------------------------------------------------------------------------------- fundamental
def _find_fundamental_lmeds(pts4: np.ndarray, confidence: float, max_iters: int, info: "RansacInfo"):
    """8..14 points: OpenCV 4.6.0's findFundamentalMat runs its LMedS registrator even when RANSAC was
    asked for.  The iteration count is fixed up front (outlier ratio 0.45 assumed, at
    least 3), models are ranked by the median of their errors (first strictly smallest wins), inliers
    lie within sigma = 2.5 * 1.4826 * (1 + 5/(n-7)) * sqrt(median), at least 0.001."""
    n = pts4.shape[0]
    if confidence < _DBL_EPS or confidence > 1 - _DBL_EPS:
        confidence = 0.99
    niters = max(ops.update_num_iters(confidence, 0.45, 7, max_iters), 3)
    info.niters = niters
    mask0 = np.zeros((n, 1), np.uint8)
    subsets = ops.gen_subsets(n, 7, niters, pts4, max_attempts=1000)      # LMedS: getSubset's default maxAttempts
    info.n_generated = len(subsets)
    if len(subsets) == 0:
        return None, mask0, info
    pts_d = _h2d(pts4)
    models, nm = ops.solve_7pt(_h2d(subsets), pts_d)                      # (h,3,9), (h,)
    valid = (torch.arange(3, device=nm.device, dtype=torch.int32)[None, :] < nm[:, None]).to(torch.uint8)
    med = ops.lmeds_median_fund(models.reshape(-1, 9), pts_d, valid.reshape(-1)).cpu().numpy().astype(np.float64)
    best, min_median = -1, float(np.finfo(np.float64).max)
    for i, m in enumerate(med):                                           # "median < minMedian", in order
        if m < min_median:
            best, min_median = i, float(m)
    if best < 0:
        return None, mask0, info
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * float(np.sqrt(min_median)), 0.001)
    model = models.reshape(-1, 9)[best]
    mask = ops.mask_fund(model, pts_d, sigma).cpu().numpy().reshape(n, 1)
    info.best_iter, info.best_model, info.max_good = best // 3, best % 3, int(mask.sum())
    if info.max_good < 7:
        return None, mask, info
    return model.reshape(3, 3).cpu().numpy(), mask, info


def find_fundamental_ransac(points1, points2, threshold: float, confidence: float, max_iters: int,
                            group=None) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], RansacInfo]:
    """cv2.findFundamentalMat(p1, p2, cv2.RANSAC, threshold, confidence, max_iters).

    group: None = this process alone.  ``parallel.WORLD`` or a ProcessGroup = the samples are split
    over that group's ranks (every rank calls with the same inputs, all get the same result)."""
    m1 = np.ascontiguousarray(np.asarray(points1).reshape(-1, 2), dtype=np.float32)
    m2 = np.ascontiguousarray(np.asarray(points2).reshape(-1, 2), dtype=np.float32)
    if m1.shape[0] != m2.shape[0]:
        raise ValueError("findFundamentalMat: point sets differ in length")
    n = m1.shape[0]
    info = RansacInfo()
    if n < 7:
        return None, None, info
    pts4 = np.concatenate([m1, m2], axis=1)
    if n == 7:
        # direct 7-point solve; OpenCV stacks all solutions (3k x 3) and sets the mask to ones
        models, nm = ops.solve_7pt(_h2d(np.arange(7, dtype=np.int32)[None]), _h2d(pts4))
        k = int(nm.cpu()[0])
        if k <= 0:
            return None, np.ones((n, 1), np.uint8), info
        return models[0, :k].reshape(3 * k, 3).cpu().numpy(), np.ones((n, 1), np.uint8), info
    if n < 15:
        return _find_fundamental_lmeds(pts4, confidence, max_iters, info)
    if threshold <= 0:
        threshold = 3
    if confidence < _DBL_EPS or confidence > 1 - _DBL_EPS:
        confidence = 0.99

    subsets = ops.gen_subsets(n, 7, max_iters, pts4)
    info.n_generated = len(subsets)
    mask0 = np.zeros((n, 1), np.uint8)
    if len(subsets) == 0:
        return None, mask0, info
    pts_d = _h2d(pts4)

    def solve_and_score(lo, hi):
        models, nm = ops.solve_7pt(_h2d(subsets[lo:hi]), pts_d)          # (h,3,9), (h,)
        valid = (torch.arange(3, device=nm.device, dtype=torch.int32)[None, :] < nm[:, None]).to(torch.uint8)
        counts = ops.score_fund(models.reshape(-1, 9), pts_d, threshold, valid.reshape(-1))
        return counts.reshape(-1, 3), nm, models

    def replay(counts, nm):
        return tuple(int(v) for v in ops.ransac_replay(counts, nm, n, 7, confidence, max_iters).cpu())

    best_iter, best_k, max_good, niters, model = sharded_ransac(len(subsets), 3, 9, solve_and_score, replay, group)
    info.best_iter, info.best_model, info.max_good, info.niters = best_iter, best_k, max_good, niters
    if best_iter < 0 or max_good <= 0:
        return None, mask0, info
    mask = ops.mask_fund(model, pts_d, threshold)
    return model.reshape(3, 3).cpu().numpy(), mask.cpu().numpy().reshape(n, 1), info


# ------------------------------------------------------------------------------- essential
def find_essential_ransac(points1, points2, K, prob: float, threshold: float, max_iters: int,
                          group=None) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], RansacInfo]:
    """cv2.findEssentialMat(p1, p2, K, cv2.RANSAC, prob, threshold, max_iters)."""
    p1 = np.asarray(points1, dtype=np.float64).reshape(-1, 2)
    p2 = np.asarray(points2, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(K, dtype=np.float64)
    n = p1.shape[0]
    info = RansacInfo()
    if p2.shape[0] != n:
        raise ValueError("findEssentialMat: point sets differ in length")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    x4 = np.empty((n, 4), dtype=np.float64)
    x4[:, 0] = p1[:, 0] * ifx + (-cx * ifx)
    x4[:, 1] = p1[:, 1] * ify + (-cy * ify)
    x4[:, 2] = p2[:, 0] * ifx + (-cx * ifx)
    x4[:, 3] = p2[:, 1] * ify + (-cy * ify)
    threshold = threshold / ((fx + fy) / 2)
    if n < 5:
        return None, None, info
    if n == 5:
        models, nm = ops.solve_5pt(_h2d(np.arange(5, dtype=np.int32)[None]), _h2d(x4))
        k = int(nm.cpu()[0])
        if k <= 0:
            return None, np.ones((n, 1), np.uint8), info
        # OpenCV quirk: with exactly 5 points every solution comes back, stacked (3k x 3)
        return models[0, :k].reshape(3 * k, 3).cpu().numpy(), np.ones((n, 1), np.uint8), info
    subsets = ops.gen_subsets(n, 5, max_iters, None)
    info.n_generated = len(subsets)
    mask0 = np.zeros((n, 1), np.uint8)
    if len(subsets) == 0:
        return None, mask0, info
    pts_d = _h2d(x4)

    def solve_and_score(lo, hi):
        models, nm = ops.solve_5pt(_h2d(subsets[lo:hi]), pts_d)          # (h,10,9)
        valid = (torch.arange(10, device=nm.device, dtype=torch.int32)[None, :] < nm[:, None]).to(torch.uint8)
        counts = ops.score_ess(models.reshape(-1, 9), pts_d, threshold, valid.reshape(-1))
        return counts.reshape(-1, 10), nm, models

    def replay(counts, nm):
        return tuple(int(v) for v in ops.ransac_replay(counts, nm, n, 5, prob, max_iters).cpu())

    best_iter, best_k, max_good, niters, model = sharded_ransac(len(subsets), 10, 9, solve_and_score, replay, group)
    info.best_iter, info.best_model, info.max_good, info.niters = best_iter, best_k, max_good, niters
    if best_iter < 0 or max_good <= 0:
        return None, mask0, info
    mask = ops.mask_ess(model, pts_d, threshold)
    return model.reshape(3, 3).cpu().numpy(), mask.cpu().numpy().reshape(n, 1), info


# ------------------------------------------------------------------------------- PnP
def solve_pnp_ransac_p3p(object_points, image_points, K, iterations_count: int, reprojection_error: float,
                         confidence: float, group=None):
    """cv.solvePnPRansac(obj, img, K, distCoeffs=0, flags=SOLVEPNP_P3P, ...).

    -> (ok, rvec (3,1) f64, tvec (3,1) f64, inliers (L,1) i32 | None, RansacInfo)
    """
    obj = np.ascontiguousarray(np.asarray(object_points).reshape(-1, 3), dtype=np.float32)
    img = np.ascontiguousarray(np.asarray(image_points).reshape(-1, 2), dtype=np.float32)
    K = np.asarray(K, dtype=np.float64)
    n = obj.shape[0]
    info = RansacInfo()
    if n < 4 or img.shape[0] != n:
        raise ValueError("solvePnPRansac needs >= 4 correspondences of equal count "
                         f"(got {n} object and {img.shape[0]} image points)")
    if not (0 < confidence < 1):
        raise ValueError("solvePnPRansac: confidence must be in (0,1)")
    threshold = float(np.float32(reprojection_error))          # the API takes a float
    zeros = np.zeros((3, 1), np.float64)
    obj_d, img_d = _h2d(obj), _h2d(img)
    if n == 4:
        models, rvecs, nm = ops.solve_p3p(_h2d(np.arange(4, dtype=np.int32)[None]), obj_d, img_d, K)
        if int(nm.cpu()[0]) <= 0:
            return False, zeros, zeros.copy(), None, info
        rv = rvecs[0].cpu().numpy().reshape(3, 1)
        tv = models[0, 9:].cpu().numpy().reshape(3, 1)
        return True, rv, tv, np.arange(4, dtype=np.int32).reshape(4, 1), info

    subsets = ops.gen_subsets(n, 4, iterations_count, None)
    info.n_generated = len(subsets)
    if len(subsets) == 0:
        return False, zeros, zeros.copy(), None, info

    def solve_and_score(lo, hi):
        models, rvecs, nm = ops.solve_p3p(_h2d(subsets[lo:hi]), obj_d, img_d, K)   # (h,12), (h,3), (h,)
        counts = ops.score_pnp(models, obj_d, img_d, K, threshold, nm.to(torch.uint8))
        return counts.reshape(-1, 1), nm, torch.cat([models, rvecs], dim=1).reshape(-1, 1, 15)

    def replay(counts, nm):
        return tuple(int(v) for v in ops.ransac_replay(counts, nm, n, 4, confidence, iterations_count).cpu())

    best_iter, best_k, max_good, niters, packed = sharded_ransac(len(subsets), 1, 15, solve_and_score, replay, group)
    info.best_iter, info.best_model, info.max_good, info.niters = best_iter, best_k, max_good, niters
    if best_iter < 0 or max_good <= 0:
        return False, zeros, zeros.copy(), None, info
    model = packed[:12].contiguous()
    mask = ops.mask_pnp(model, obj_d, img_d, K, threshold).cpu().numpy().astype(bool)
    p3p_rvec = packed[12:].cpu().numpy().reshape(3, 1)
    p3p_tvec = model[9:].cpu().numpy().reshape(3, 1)

    # EPnP refit on the winner's inliers (serial, O(inliers), host): the pose OpenCV returns
    ok, R, t = epnp_pose(obj[mask].astype(np.float64), img[mask].astype(np.float64), K)
    if not ok:
        return False, p3p_rvec, p3p_tvec, None, info
    inliers = np.nonzero(mask)[0].astype(np.int32).reshape(-1, 1)
    return True, rodrigues_matrix_to_vector(R).reshape(3, 1), t.reshape(3, 1), inliers, info
