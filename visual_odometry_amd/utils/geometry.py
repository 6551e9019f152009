"""Drop-in for the three functions of reference src/utils/geometry.py:37-49 -- same names,
argument order, defaults and return tuples -- with the OpenCV calls replaced by the MI355X RANSAC
engines of ``visual_odometry_amd.ransac``."""
import numpy as np

from .. import ransac
from ..params import get_params


def calc_fundamental_mat(points1, points2):
    """cv2.findFundamentalMat(p1, p2, cv2.RANSAC, thr, conf, iters) with the active dataset's
    INIT_PARAMS.RANSAC_PARAMS_F -> (F (3,3) f64 | None, mask (N,1) u8 | None)."""
    cfg = get_params().INIT_PARAMS.RANSAC_PARAMS_F
    F, mask, _ = ransac.find_fundamental_ransac(points1, points2, cfg.THRESHOLD, cfg.CONFIDENCE,
                                                cfg.NUM_ITERATIONS)
    return F, mask


def calc_essential_mat(points1, points2, K, reprojection_error=1, confidence=0.9999, num_iterations=50000):
    """cv2.findEssentialMat(p1, p2, K, cv2.RANSAC, confidence, reprojection_error, num_iterations)."""
    E, mask_e, _ = ransac.find_essential_ransac(points1, points2, K, confidence, reprojection_error,
                                                num_iterations)
    return E, mask_e


def calc_essential_mat_from_fundamental_mat(F, K):
    K = np.asarray(K)
    return K.T @ F @ K
