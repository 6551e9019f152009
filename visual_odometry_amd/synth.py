"""Synthetic inputs of the shapes the reference feeds its hot path: there are no
datasets and no SIFT detector in this environment, so benches and tests use these generators.

* ``sift_like_pair``  -- descriptor sets with OpenCV-SIFT-like statistics (integer 0..255 values, L2
  norm ~512) where a fraction of train rows are noisy copies of query rows;
* ``two_view_scene`` / ``pnp_scene`` -- point correspondences for a camera with one of the
  reference's hard-coded intrinsics (src/utils/dataloader.py:93-109), with pixel noise and outliers.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

K_KITTI = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], dtype=np.float64)
K_MALAGA = np.array([[621.18428, 0, 404.0076], [0, 621.18428, 309.05989], [0, 0, 1]], dtype=np.float64)
K_PARKING = np.array([[331.37, 0, 320], [0, 369.568, 240], [0, 0, 1]], dtype=np.float64)
IMAGE_SIZE = {"kitti": (1241, 376), "malaga": (1024, 768), "parking": (640, 480)}


def sift_like(n: int, rng: np.random.Generator) -> np.ndarray:
    """(n,128) f32 rows: Gamma(0.6) magnitudes, L2-normalise, clip 0.2, renormalise, x512, round, saturate."""
    d = rng.gamma(0.6, 1.0, size=(n, 128))
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    d = np.minimum(d, 0.2)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    return np.clip(np.rint(d * 512.0), 0, 255).astype(np.float32)


def sift_like_pair(nq: int, nt: int, seed: int = 1234, match_frac: float = 0.6,
                   noise_sigma: float = 8.0) -> Tuple[np.ndarray, np.ndarray]:
    """Query set and a train set in which ~match_frac of rows are noisy, shuffled copies of queries."""
    rng = np.random.default_rng(seed)
    q = sift_like(nq, rng)
    t = sift_like(nt, rng)
    m = min(nq, nt)
    nmatch = int(match_frac * m)
    src = rng.permutation(nq)[:nmatch]
    dst = rng.permutation(nt)[:nmatch]
    noisy = q[src] + rng.normal(0.0, noise_sigma, size=(nmatch, 128))
    t[dst] = np.clip(np.rint(noisy), 0, 255).astype(np.float32)
    return q, t


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around integer arithmetic only)."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hashed_descriptor_pair(nq: int, nt: int, seed: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """A SIFT-shaped descriptor pair ((n,128) f32, integers 0..255, skewed towards small values, every row of
    L2 norm < 1024) made from a counter-based integer hash ONLY -- no floating-point function and no numpy
    Generator stream, so the same bytes come out on every numpy version and platform (golden fixtures whose
    inputs are regenerated instead of stored).  Half of the train rows are query rows with +-8 integer noise at
    hashed positions, the rest are fresh rows."""
    def words(n, stream):
        i = np.arange(n * 128, dtype=np.uint64) + np.uint64((seed * 4 + stream) << 40)
        return _mix64(i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))

    def rows(n, stream):
        w = words(n, stream)
        a, b, c = (w >> np.uint64(8)) & np.uint64(255), (w >> np.uint64(24)) & np.uint64(255), (w >> np.uint64(40)) & np.uint64(255)
        return ((a * b * c) >> np.uint64(17)).astype(np.int64).reshape(n, 128)        # 0..126, ~cubic skew

    q, t = rows(nq, 0), rows(nt, 1)
    w = words(nt, 2).reshape(nt, 128)
    src = (w[:, 0] >> np.uint64(20)) % np.uint64(max(nq, 1))
    copy = ((w[:, 1] >> np.uint64(33)) & np.uint64(1)).astype(bool) if nq else np.zeros(nt, bool)
    noise = ((w >> np.uint64(50)) % np.uint64(17)).astype(np.int64) - 8
    noisy = np.clip(q[src.astype(np.int64)] + noise, 0, 255) if nq else t
    t = np.where(copy[:, None], noisy, t)
    return q.astype(np.float32), t.astype(np.float32)


def _rodrigues(axis_angle: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(axis_angle)
    if th < 1e-15:
        return np.eye(3)
    k = axis_angle / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def two_view_scene(n: int, K: np.ndarray = K_KITTI, size=(1241, 376), seed: int = 4321,
                   noise_px: float = 0.3, outlier_frac: float = 0.3, max_rot_deg: float = 5.0):
    """-> dict(R, t, X (n,3) f64 in cam1, p1 (n,2) f32, p2 (n,2) f32, inlier (n,) bool)."""
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    R = _rodrigues(ax * np.deg2rad(max_rot_deg) * rng.uniform(0.3, 1.0))
    t = rng.normal(size=3)
    t /= np.linalg.norm(t)
    w, h = size
    z = rng.uniform(4.0, 40.0, n)
    u = rng.uniform(0, w, n)
    v = rng.uniform(0, h, n)
    X = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    Xc = X @ R.T + t
    p2 = Xc[:, :2] / Xc[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    p1 = np.stack([u, v], axis=1)
    if noise_px > 0:
        p1 = p1 + rng.normal(0, noise_px, p1.shape)
        p2 = p2 + rng.normal(0, noise_px, p2.shape)
    inlier = np.ones(n, dtype=bool)
    no = int(outlier_frac * n)
    if no:
        sel = rng.permutation(n)[:no]
        p2[sel] = np.stack([rng.uniform(0, w, no), rng.uniform(0, h, no)], axis=1)
        inlier[sel] = False
    return dict(R=R, t=t, X=X, p1=p1.astype(np.float32), p2=p2.astype(np.float32), inlier=inlier, K=K)


def pnp_scene(n: int, K: np.ndarray = K_MALAGA, size=(1024, 768), seed: int = 4321,
              noise_px: float = 0.3, outlier_frac: float = 0.3, max_rot_deg: float = 5.0):
    """-> dict(R, t (world->camera), obj (n,3) f32 world points, img (n,2) f32 pixels, inlier)."""
    s = two_view_scene(n, K, size, seed, noise_px, outlier_frac, max_rot_deg)
    return dict(R=s["R"], t=s["t"], obj=s["X"].astype(np.float32), img=s["p2"], inlier=s["inlier"], K=K)


def frame_pair(nq: int, nt: int, seed: int = 0, K: np.ndarray = K_MALAGA, size=(1024, 768), match_frac: float = 0.6,
               desc_noise: float = 8.0, noise_px: float = 0.3, landmark_outlier_frac: float = 0.0):
    """Two synthetic frames as SIFT would describe them, with geometry behind the true matches:
    -> dict(q, t (n,128) f32 descriptors; kp_q (nq,2), kp_t (nt,2) f32 positions; xyz_q (nq,3) f32
    landmarks of the query keypoints in the query camera frame; R, t the relative pose; K).
    ~match_frac of the train features are noisy copies of query features placed at the projection of
    the query's landmark; the rest is clutter at random positions.  landmark_outlier_frac of the query
    landmarks are replaced by unrelated points (stale or mis-triangulated landmarks: PnP outliers that the
    descriptor match cannot see)."""
    rng = np.random.default_rng(seed)
    sc = two_view_scene(nq, K=K, size=size, seed=seed, noise_px=noise_px, outlier_frac=0.0)
    q = sift_like(nq, rng)
    t = sift_like(nt, rng)
    nmatch = int(match_frac * min(nq, nt))
    src, dst = rng.permutation(nq)[:nmatch], rng.permutation(nt)[:nmatch]
    t[dst] = np.clip(np.rint(q[src] + rng.normal(0.0, desc_noise, (nmatch, 128))), 0, 255).astype(np.float32)
    kp_t = np.stack([rng.uniform(0, size[0], nt), rng.uniform(0, size[1], nt)], axis=1).astype(np.float32)
    kp_t[dst] = sc["p2"][src]
    xyz = sc["X"].astype(np.float32)
    if landmark_outlier_frac > 0:
        bad = rng.permutation(nq)[:int(landmark_outlier_frac * nq)]
        xyz[bad] = xyz[rng.permutation(nq)[:len(bad)]] + rng.normal(0, 1.0, (len(bad), 3)).astype(np.float32)
    return dict(q=q, t=t, kp_q=sc["p1"].astype(np.float32), kp_t=kp_t, xyz_q=xyz, R=sc["R"], t_vec=sc["t"], K=K)
