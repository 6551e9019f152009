"""cv.Rodrigues in numpy (host, O(1) per call): reference src/pnp.py:32 converts the rotation
vector returned by solvePnPRansac into a matrix.  Formulas as in OpenCV's calibration.cpp
(SURVEY.md Appendix A.6)."""
from __future__ import annotations

import numpy as np

_DBL_EPS = float(np.finfo(np.float64).eps)


def rodrigues_vector_to_matrix(rvec) -> np.ndarray:
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < _DBL_EPS:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    k = r * (1.0 / theta)
    rrt = np.outer(k, k)
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return (c * np.eye(3) + (1.0 - c) * rrt) + s * kx


def rodrigues_matrix_to_vector(R) -> np.ndarray:
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    if not np.all((R > -100) & (R < 100)):
        return np.zeros(3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(float(v @ v) * 0.25)
    c = min(1.0, max(-1.0, (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        x = np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
        y = np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        z = np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and ((R[1, 2] > 0) != (y * z > 0)):
            z = -z
        v = np.array([x, y, z])
        return v * (theta / np.linalg.norm(v))
    return v * (theta / (2 * s))


def Rodrigues(src):
    """cv.Rodrigues-shaped: (3,), (3,1) or (1,3) vector -> (R (3,3), None); (3,3) -> (rvec (3,1), None)."""
    a = np.asarray(src, dtype=np.float64)
    if a.shape == (3, 3):
        return rodrigues_matrix_to_vector(a).reshape(3, 1), None
    return rodrigues_vector_to_matrix(a), None
