"""Drop-in for reference src/utils/disambiguate_relative_pose.py:5-72: choose among the four
(R, +-t) candidates by cheirality, then mask points behind the camera or too far away."""
import numpy as np

from ..params import get_params
from .triangulate import triangulatePoints


def disambiguateRelativePose(Rots, u3, points0_h, points1_h, K1, K2):
    """Rots (3,3,2), u3 (3,), points*_h (3,N) homogeneous pixels -> (T_cami_camj (4,4),
    P_cami (L,4) dehomogenised points in camera-i coordinates, mask (N,) bool).

    Candidate order and tie rule as in the reference: rotations outer, sign of t inner (+ first),
    a later candidate replaces the current one only when STRICTLY more points lie in front of both
    cameras."""
    max_depth = get_params().INIT_PARAMS.MAX_DEPTH_DISTANCE
    M1 = K1 @ np.eye(3, 4)
    best_score, best = -np.inf, None
    for r in range(2):
        R = Rots[:, :, r]
        for sign in (1.0, -1.0):
            t = u3 * sign
            Rt = np.c_[R, t]
            X1 = triangulatePoints(M1, K2 @ Rt, points0_h[:2, :], points1_h[:2, :])
            X1 = X1 / X1[3, :]
            X2 = Rt @ X1
            score = np.sum(X1[2, :] > 0) + np.sum(X2[2, :] > 0)
            if score > best_score:
                best_score, best = score, (R, t, X1)
    R, t, P_cami = best
    T_cami_camj = np.eye(4)
    T_cami_camj[:3, :3] = R
    T_cami_camj[:3, 3] = t.ravel()
    min_depth = min(0, T_cami_camj[2, 2])
    mask = np.logical_and(P_cami[2, :] > min_depth, np.abs(np.linalg.norm(P_cami, axis=0)) < max_depth)
    print(f"2D2D/DISAMBIGUATE: {np.sum(~mask)} rejected points during 3D reconstruction")
    return T_cami_camj, P_cami[:, mask].T, mask
