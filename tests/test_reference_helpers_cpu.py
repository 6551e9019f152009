"""The product's own numpy drop-ins for the reference's pure-numpy helpers, checked DIRECTLY against outputs of the
reference's code (tests/golden/ref_helpers.npz: made by tests/golden/make_golden.py importing
/root/reference/src/utils/decompose_essential_matrix.py:4-37 and src/utils/utils.py:100-106 in the build container).
No oracle, no GPU: this is the part of the path whose parity IS pinned by the reference itself."""
import os

import numpy as np

from visual_odometry_amd.utils.decompose_essential_matrix import decomposeEssentialMatrix
from visual_odometry_amd.utils.utils import hom_inv

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_helpers.npz"))


def test_decompose_essential_matrix_equals_reference_outputs():
    assert len(G["E"]) >= 4
    for E, R, u3 in zip(G["E"], G["R"], G["u3"]):
        r, u = decomposeEssentialMatrix(E.copy())
        assert r.shape == (3, 3, 2) and u.shape == (3,)
        assert np.allclose(r, R, atol=1e-12) and np.allclose(u, u3, atol=1e-12)
        for i in range(2):                                  # proper rotations, as the reference returns them
            assert abs(np.linalg.det(r[:, :, i]) - 1.0) < 1e-12


def test_hom_inv_equals_reference_outputs():
    for T, Tinv in zip(G["T"], G["Tinv"]):
        got = hom_inv(T)
        assert got.dtype == Tinv.dtype and np.allclose(got, Tinv, atol=1e-15)
        assert np.allclose(got @ T, np.eye(4), atol=1e-12)
