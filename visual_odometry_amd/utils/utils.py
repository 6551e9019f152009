"""``hom_inv`` of reference src/utils/utils.py:100-106 (the only helper of that module on the
hot path: ``pnp`` returns its result)."""
import numpy as np


def hom_inv(T: np.ndarray) -> np.ndarray:
    """Inverse of a rigid 4x4 transform without a general matrix inverse."""
    T = np.asarray(T)
    assert T.shape == (4, 4)
    Rt = T[:3, :3].T
    out = np.zeros_like(T)
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ T[:3, 3]
    out[3, 3] = 1
    return out
