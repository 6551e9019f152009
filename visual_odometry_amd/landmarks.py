"""New-landmark triangulation, the geometric core of TrackManager.get_new_landmarks (reference
src/tracking.py:150-172): the start / end keypoints of a track and the two camera poses in, world
points in front of camera j and closer than MAX_DEPTH_DISTANCE out.  The track bookkeeping around
it (baseline test, track pruning: src/tracking.py:125-148,173-238) is out of scope."""
import numpy as np

from . import cv2_shim


def triangulate_track(K, cam_to_world_i, cam_to_world_j, keypoints_start, keypoints_end, max_depth_distance):
    """-> (P_world (L,4) homogeneous with last column 1, kp_j (L,3) homogeneous end keypoints,
    mask_distance (n,) bool), with the reference's operation order: cv.triangulatePoints on
    K [R|t]_i, K [R|t]_j, dehomogenise, transform into camera j, keep z > 0 and |P_camj| < max depth
    (norm over all four homogeneous components, as the reference writes it)."""
    K = np.asarray(K, dtype=np.float64)
    Ti, Tj = np.linalg.inv(cam_to_world_i), np.linalg.inv(cam_to_world_j)
    n = len(keypoints_start)
    kp_i = np.hstack([keypoints_start, np.ones((n, 1))])
    kp_j = np.hstack([keypoints_end, np.ones((n, 1))])
    P_world = cv2_shim.triangulatePoints(K @ Ti[0:3, :], K @ Tj[0:3, :], kp_i[:, :2].T, kp_j[:, :2].T).T
    P_world = P_world / np.reshape(P_world[:, 3], (-1, 1))
    P_camj = (Tj @ P_world.T).T
    print(f"Found {P_world.shape} new landmarks")
    if P_world.shape[0] == 0:
        return P_world, kp_j, np.zeros((0,), bool)
    mask_distance = np.logical_and(P_camj[:, 2] > 0, np.abs(np.linalg.norm(P_camj, axis=1)) < max_depth_distance)
    print(f"filtering out {np.sum(np.logical_not(mask_distance))} landmarks that"
          " are too far away or behind the camera")
    return P_world[mask_distance, :], kp_j[mask_distance, :], mask_distance
