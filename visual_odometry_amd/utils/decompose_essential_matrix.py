"""Drop-in for reference src/utils/decompose_essential_matrix.py:4-37 (O(1), host)."""
import numpy as np

_W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])


def decomposeEssentialMatrix(E):
    """E (3,3) -> (R (3,3,2): the two rotations U W V^T and U W^T V^T made proper, u3 (3,): the
    unit left singular vector of the smallest singular value, i.e. the translation direction)."""
    U, _, Vt = np.linalg.svd(E)
    R = np.stack([U @ _W @ Vt, U @ _W.T @ Vt], axis=2)
    for i in range(2):
        if np.linalg.det(R[:, :, i]) < 0:
            R[:, :, i] = -R[:, :, i]
    u3 = U[:, 2]
    n = np.linalg.norm(u3)
    if n != 0:
        u3 = u3 / n
    return R, u3
