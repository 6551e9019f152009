"""cv2-shaped entry points for the three OpenCV calls on the reference's hot path, so a caller can
swap the module object instead of the wrapper functions:

    cv.BFMatcher().knnMatch(q, t, k=2)                      reference src/features.py:100-101
    cv2.findFundamentalMat(p1, p2, cv2.RANSAC, thr, conf, n) reference src/utils/geometry.py:39
    cv2.findEssentialMat(p1, p2, K, cv2.RANSAC, prob, thr, n) reference src/utils/geometry.py:44
    cv.solvePnPRansac(..., flags=cv.SOLVEPNP_P3P)            reference src/pnp.py:16-26
    cv.Rodrigues(rvec)                                       reference src/pnp.py:32

Only the argument combinations the reference uses are implemented; anything else raises
NotImplementedError rather than silently doing something different.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import ransac
from .features import knn2_arrays
from .rodrigues import Rodrigues  # noqa: F401  (re-exported)

NORM_L2 = 4
FM_RANSAC = 8
RANSAC = 8
SOLVEPNP_ITERATIVE = 0
SOLVEPNP_EPNP = 1
SOLVEPNP_P3P = 2


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx: int = -1, trainIdx: int = -1, imgIdx: int = -1, distance: float = 3.4028234663852886e38):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = queryIdx, trainIdx, imgIdx, distance

    def __repr__(self):
        return f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, imgIdx={self.imgIdx}, distance={self.distance})"


class BFMatcher:
    def __init__(self, normType: int = NORM_L2, crossCheck: bool = False):
        if normType != NORM_L2 or crossCheck:
            raise NotImplementedError("only BFMatcher(NORM_L2, crossCheck=False), the reference's default, is provided")

    def knnMatch(self, queryDescriptors, trainDescriptors, k: int = 2) -> List[List[DMatch]]:
        if k not in (1, 2):
            raise NotImplementedError("knnMatch is provided for k <= 2 (the reference uses k=2)")
        idx, dist, _ = knn2_arrays(queryDescriptors, trainDescriptors)
        out = []
        idx_l, dist_l = idx.tolist(), dist.astype(np.float64).tolist()
        for qi, (ids, ds) in enumerate(zip(idx_l, dist_l)):
            out.append([DMatch(qi, ids[c], 0, ds[c]) for c in range(k) if ids[c] >= 0])
        return out


def findFundamentalMat(points1, points2, method: int = FM_RANSAC, ransacReprojThreshold: float = 3.0,
                       confidence: float = 0.99, maxIters: int = 1000):
    if method != FM_RANSAC:
        raise NotImplementedError("findFundamentalMat: only cv2.RANSAC is provided")
    F, mask, _ = ransac.find_fundamental_ransac(points1, points2, ransacReprojThreshold, confidence, maxIters)
    return F, mask


def findEssentialMat(points1, points2, cameraMatrix, method: int = RANSAC, prob: float = 0.999,
                     threshold: float = 1.0, maxIters: int = 1000):
    if method != RANSAC:
        raise NotImplementedError("findEssentialMat: only cv2.RANSAC is provided")
    E, mask, _ = ransac.find_essential_ransac(points1, points2, cameraMatrix, prob, threshold, maxIters)
    return E, mask


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, rvec=None, tvec=None,
                   useExtrinsicGuess: bool = False, iterationsCount: int = 100, reprojectionError: float = 8.0,
                   confidence: float = 0.99, inliers=None, flags: int = SOLVEPNP_ITERATIVE):
    if flags != SOLVEPNP_P3P:
        raise NotImplementedError("solvePnPRansac: only flags=SOLVEPNP_P3P (the reference's choice) is provided")
    if useExtrinsicGuess:
        raise NotImplementedError("solvePnPRansac: useExtrinsicGuess is not provided")
    if distCoeffs is not None and np.any(np.asarray(distCoeffs, dtype=np.float64) != 0):
        raise NotImplementedError("solvePnPRansac: only zero distortion (the reference's distCoeffs) is provided")
    ok, rv, tv, inl, _ = ransac.solve_pnp_ransac_p3p(objectPoints, imagePoints, cameraMatrix, iterationsCount,
                                                     reprojectionError, confidence)
    return ok, rv, tv, inl


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2) -> np.ndarray:
    """cv.triangulatePoints (reference src/utils/disambiguate_relative_pose.py:42, src/tracking.py:153):
    (3,4) projection matrices, (2,N) points -> (4,N) homogeneous points, computed on the device."""
    import torch
    from . import ops
    x1 = np.ascontiguousarray(np.asarray(projPoints1, dtype=np.float64).reshape(2, -1).T)
    x2 = np.ascontiguousarray(np.asarray(projPoints2, dtype=np.float64).reshape(2, -1).T)
    if x1.shape != x2.shape:
        raise ValueError("triangulatePoints: point sets differ in length")
    if x1.shape[0] == 0:
        return np.zeros((4, 0))
    dev = torch.device("cuda", torch.cuda.current_device())
    X = ops.triangulate(projMatr1, projMatr2, torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev))
    return np.ascontiguousarray(X.cpu().numpy().T)
