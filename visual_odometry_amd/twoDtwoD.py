"""The geometry half of the reference's 2D-2D initialisation (src/twoDtwoD.py:57-79 and :81-94):
matched pixel positions in, relative pose / landmarks out.  Image reading, SIFT detection and KLT
(src/twoDtwoD.py:28-46) are image-domain stages outside this package; descriptor matching is
``features.match_features``."""
import numpy as np

from .utils import geometry as geom
from .utils.decompose_essential_matrix import decomposeEssentialMatrix
from .utils.disambiguate_relative_pose import disambiguateRelativePose
from .utils.utils import hom_inv


def _homogeneous_inliers(pos, mask_f):
    p = np.hstack([pos, np.ones((pos.shape[0], 1))]).T
    return p[:, mask_f.ravel() == 1]


def twoDtwoD_from_matches(pos_i, pos_j, K, cam_to_world_i=None):
    """src/twoDtwoD.py:57-79 -> (M_camj_to_world (4,4), P_world (L,3), p_i (L,2), p_j (L,2)).

    pos_i / pos_j: (M,2) matched pixel positions (``Features.get_positions()``)."""
    F, mask_f = geom.calc_fundamental_mat(pos_i, pos_j)
    E = geom.calc_essential_mat_from_fundamental_mat(F, K)
    p_i, p_j = _homogeneous_inliers(pos_i, mask_f), _homogeneous_inliers(pos_j, mask_f)
    R, u3 = decomposeEssentialMatrix(E)
    T_cami_camj, P_cami, mask = disambiguateRelativePose(R, u3, p_i, p_j, K, K)
    M_cami_to_world = np.eye(4) if cam_to_world_i is None else np.asarray(cam_to_world_i)
    M_camj_to_world = M_cami_to_world @ hom_inv(T_cami_camj)
    P_world = M_cami_to_world @ P_cami.T
    return M_camj_to_world, P_world[:3, :].T, p_i[:2, mask].squeeze().T, p_j[:2, mask].squeeze().T


def calculate_relative_pose(points_i, points_j, K):
    """src/twoDtwoD.py:81-94 -> (R_cami_to_camj (3,3), T_cami_to_camj (3,))."""
    F, mask_f = geom.calc_fundamental_mat(points_i, points_j)
    E = geom.calc_essential_mat_from_fundamental_mat(F, K)
    R, u3 = decomposeEssentialMatrix(E)
    T, _, _ = disambiguateRelativePose(R, u3, _homogeneous_inliers(points_i, mask_f),
                                       _homogeneous_inliers(points_j, mask_f), K, K)
    return T[:3, :3], T[:3, 3]


def twoDtwoD_from_pipeline(res, tail, b: int, cam_to_world_i=None):
    """The return tuple of src/twoDtwoD.py:57-79 for pair `b` of a device-resident batch
    (``pipeline.match_and_ransac`` + ``pipeline.pose_tail``): only the small results are downloaded.
    -> (M_camj_to_world (4,4), P_world (L,3), p_i (L,2), p_j (L,2))."""
    n = int(res.n_match[b])
    keep = tail["keep"][b, :n].cpu().numpy().astype(bool)
    T = tail["T"][b].cpu().numpy()
    P_cami = tail["X"][b, :n].cpu().numpy()[keep]
    pts = res.pts4[b, :n].cpu().numpy().astype(np.float64)[keep]
    M_cami_to_world = np.eye(4) if cam_to_world_i is None else np.asarray(cam_to_world_i)
    M_camj_to_world = M_cami_to_world @ hom_inv(T)
    P_world = M_cami_to_world @ P_cami.T
    return M_camj_to_world, P_world[:3, :].T, pts[:, :2], pts[:, 2:]
