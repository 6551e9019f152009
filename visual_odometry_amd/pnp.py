"""Drop-in for ``pnp`` of reference src/pnp.py:13-41: P3P RANSAC + EPnP refit on the MI355X, then
the same Rodrigues -> 4x4 -> inverse tail."""
import numpy as np

from . import ransac
from .params import get_params
from .rodrigues import rodrigues_vector_to_matrix
from .utils.utils import hom_inv


def pnp(points_2d: np.ndarray, points_3d: np.ndarray, K: np.ndarray):
    """-> (cam_to_world (4,4) f64, inlier indices (L,1) int32 | None).

    Like the reference, a failed RANSAC is reported on stdout and NOT raised: the caller gets the
    transform built from whatever rvec/tvec came back and ``None`` for the inliers."""
    cfg = get_params().CONT_PARAMS.PNP_RANSAC_PARAMS
    ok, rvec, tvec, inliers, _ = ransac.solve_pnp_ransac_p3p(
        points_3d, points_2d, K, cfg.NUM_ITERATIONS, cfg.THRESHOLD, cfg.CONFIDENCE)
    if not ok:
        print("OpenCV PNP failed")
    world_to_cam = np.eye(4)
    world_to_cam[:3, :3] = rodrigues_vector_to_matrix(rvec)
    world_to_cam[:3, 3] = np.asarray(tvec, dtype=np.float64).ravel()
    return hom_inv(world_to_cam), inliers
