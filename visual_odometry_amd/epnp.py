"""EPnP refit on the inliers of the winning RANSAC model: what cv.solvePnPRansac(flags=SOLVEPNP_P3P)
returns as its pose (reference src/pnp.py:16-26).  Serial, O(inliers), once
per call: a host routine of ``libvo_hip.so`` (``csrc/vo_epnp.hip``), not a GPU kernel."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _lib


def epnp_pose(object_points, image_points, K) -> Tuple[bool, np.ndarray, np.ndarray]:
    """(n,3) world points, (n,2) pixels, K (3,3) -> (ok, R (3,3), t (3,)) mapping world -> camera."""
    obj = np.ascontiguousarray(np.asarray(object_points, dtype=np.float64).reshape(-1, 3))
    img = np.ascontiguousarray(np.asarray(image_points, dtype=np.float64).reshape(-1, 2))
    K = np.asarray(K, dtype=np.float64)
    n = obj.shape[0]
    if n < 4 or img.shape[0] != n:
        return False, np.eye(3), np.zeros(3)
    K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dtype=np.float64)
    R, t = np.zeros(9), np.zeros(3)
    rc = _lib.load().vo_host_epnp(obj.ctypes.data, img.ctypes.data, n, K4.ctypes.data, R.ctypes.data, t.ctypes.data)
    _lib.check(rc, "vo_host_epnp")
    return rc == 1, R.reshape(3, 3), t
