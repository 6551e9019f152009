"""No GPU needed: the C-ABI library loads, exports what include/vo_hip.h declares, its host-only
entry points agree with the oracle, and device entry points refuse to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from visual_odometry_amd import _lib, ops, synth
from visual_odometry_amd.epnp import epnp_pose
from visual_odometry_amd.rodrigues import Rodrigues, rodrigues_matrix_to_vector, rodrigues_vector_to_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    lib = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vo_hip.h but not exported"
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)
    assert _lib.load().vo_version() >= 100


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.VoError):
        ops.desc_to_i8(torch.zeros((4, 128)))
    from visual_odometry_amd import features
    with pytest.raises(_lib.VoError):
        features.knn2_arrays(np.zeros((4, 128), np.float32), np.zeros((4, 128), np.float32))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "visual_odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "vo_oracle.h" not in src and "libvo_oracle" not in src, f


@pytest.mark.parametrize("n,m,iters", [(15, 7, 64), (100, 4, 64), (8192, 5, 64), (2000, 7, 300), (4, 4, 10), (3, 4, 10)])
def test_host_subsets_equal_oracle(oracle_mod, n, m, iters):
    sc = synth.two_view_scene(max(n, 8), seed=n)
    p1, p2 = sc["p1"][:n], sc["p2"][:n]
    if m == 7:
        a = ops.gen_subsets(n, m, iters, np.concatenate([p1, p2], 1))
        b = oracle_mod.gen_subsets(n, m, iters, p1, p2)
    else:
        a = ops.gen_subsets(n, m, iters)
        b = oracle_mod.gen_subsets(n, m, iters)
    assert np.array_equal(a, b)
    if n >= m:
        assert len(a) == iters and all(len(set(r)) == m for r in a.tolist())
    else:
        assert len(a) == 0


def test_host_subsets_collinear_rejection(oracle_mod):
    """all points on one line in image 1: checkSubset rejects every sample, generation stops."""
    n = 20
    p1 = np.stack([np.arange(n, dtype=np.float32), 2 * np.arange(n, dtype=np.float32)], 1)
    p2 = np.random.default_rng(0).uniform(0, 100, (n, 2)).astype(np.float32)
    a = ops.gen_subsets(n, 7, 5, np.concatenate([p1, p2], 1))
    b = oracle_mod.gen_subsets(n, 7, 5, p1, p2)
    assert len(a) == 0 and len(b) == 0


def test_update_num_iters_equal_oracle(oracle_mod):
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for _ in range(2000):
        p = float(rng.choice([0.9, 0.99, 0.999, 0.9999, 0.5]))
        ep = float(rng.uniform(0, 1))
        m = int(rng.choice([4, 5, 7]))
        mi = int(rng.choice([1, 50, 100, 10000, 65536]))
        assert lib.vo_host_update_num_iters(p, ep, m, mi) == oracle_mod.update_num_iters(p, ep, m, mi)
    for ep in (0.0, 1.0):
        assert lib.vo_host_update_num_iters(0.99, ep, 7, 100) == oracle_mod.update_num_iters(0.99, ep, 7, 100)


@pytest.mark.parametrize("n,seed,noise", [(500, 0, 0.3), (40, 1, 1.0), (8, 2, 0.0), (6, 3, 0.1), (3000, 4, 0.5)])
def test_host_epnp_matches_oracle(oracle_mod, n, seed, noise):
    ps = synth.pnp_scene(n, seed=seed, outlier_frac=0.0, noise_px=noise)
    obj, img = ps["obj"].astype(np.float64), ps["img"].astype(np.float64)
    ok, R, t = epnp_pose(obj, img, ps["K"])
    ook, oR, ot = oracle_mod.epnp(obj, img, ps["K"])
    assert ok and ook
    assert np.allclose(R, oR, rtol=1e-6, atol=1e-9) and np.allclose(t, ot, rtol=1e-6, atol=1e-9)
    assert abs(np.linalg.det(R) - 1) < 1e-9


def test_rodrigues_matches_oracle(oracle_mod):
    rng = np.random.default_rng(2)
    for _ in range(200):
        r = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 1.0, 3.0])
        R = rodrigues_vector_to_matrix(r)
        assert np.allclose(R, oracle_mod.rodrigues_v2m(r), rtol=0, atol=1e-15)
        assert np.allclose(rodrigues_matrix_to_vector(R), oracle_mod.rodrigues_m2v(R), rtol=1e-9, atol=1e-12)
    R, _ = Rodrigues(np.array([[0.1], [0.2], [0.3]]))
    rv, _ = Rodrigues(R)
    assert R.shape == (3, 3) and rv.shape == (3, 1) and np.allclose(rv.ravel(), [0.1, 0.2, 0.3])
    # rotation by pi
    Rpi = rodrigues_vector_to_matrix(np.array([np.pi, 0, 0]))
    assert np.allclose(np.abs(rodrigues_matrix_to_vector(Rpi)), [np.pi, 0, 0], atol=1e-7)
