"""cv.triangulatePoints for the pose tail (reference src/utils/disambiguate_relative_pose.py:42,
src/twoDtwoD.py:110): per point the 4x4 DLT system
    [x1 P1[2]-P1[0]; y1 P1[2]-P1[1]; x2 P2[2]-P2[0]; y2 P2[2]-P2[1]] X = 0
solved by the right singular vector of the smallest singular value, on the
device (``vo_triangulate``: one thread per point, one-sided Jacobi SVD in registers).  The homogeneous
scale/sign is arbitrary, as in OpenCV, and callers divide by the 4th row.  There is no host fallback."""
from ..cv2_shim import triangulatePoints  # noqa: F401  (same entry point as the cv2-shaped shim)
