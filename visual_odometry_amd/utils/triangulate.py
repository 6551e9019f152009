"""cv.triangulatePoints for the pose tail (reference src/utils/disambiguate_relative_pose.py:42,
src/twoDtwoD.py:113): per point the 4x4 DLT system
    [x1 P1[2]-P1[0]; y1 P1[2]-P1[1]; x2 P2[2]-P2[0]; y2 P2[2]-P2[1]] X = 0
solved by the right singular vector of the smallest singular value.
Batched over all points in one numpy SVD call; the homogeneous scale/sign is arbitrary, as in
OpenCV, and callers divide by the 4th row."""
import numpy as np


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2) -> np.ndarray:
    P1 = np.asarray(projMatr1, dtype=np.float64).reshape(3, 4)
    P2 = np.asarray(projMatr2, dtype=np.float64).reshape(3, 4)
    x1 = np.asarray(projPoints1, dtype=np.float64).reshape(2, -1)
    x2 = np.asarray(projPoints2, dtype=np.float64).reshape(2, -1)
    if x1.shape[1] != x2.shape[1]:
        raise ValueError("triangulatePoints: point sets differ in length")
    n = x1.shape[1]
    A = np.empty((n, 4, 4))
    A[:, 0] = x1[0][:, None] * P1[2] - P1[0]
    A[:, 1] = x1[1][:, None] * P1[2] - P1[1]
    A[:, 2] = x2[0][:, None] * P2[2] - P2[0]
    A[:, 3] = x2[1][:, None] * P2[2] - P2[1]
    if n == 0:
        return np.zeros((4, 0))
    _, _, Vt = np.linalg.svd(A)
    return np.ascontiguousarray(Vt[:, 3, :].T)
