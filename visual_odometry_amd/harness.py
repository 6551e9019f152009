"""Synthetic stand-ins for BASELINE configs 2 and 3, driving the drop-in entry points with the call
pattern of the reference's pipeline (there are no datasets and no SIFT detector here):

* ``run_init``       -- src/twoDtwoD.py:45-79: match two frames' descriptors, F-RANSAC, pose + landmarks
                        (config 2: KITTI intrinsics and parameters);
* ``run_continuous`` -- the per-frame localisation of src/tracking.py:382-414 with descriptor matching
                        in place of KLT, as in src/sift_test.py:104-119: match the keyframe's
                        landmark-bearing features against each new frame, then ``pnp`` on the matched
                        (2D, 3D) pairs (config 3: Malaga intrinsics and parameters).

A "frame" is what SIFT would have produced for it: keypoint positions and 128-D integer descriptors.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from . import params, synth
from .features import Features, Keypoint, match_features
from .pnp import pnp
from .twoDtwoD import twoDtwoD_from_matches


@dataclass
class Frame:
    features: Features            # keypoints (object array of Keypoint) + descriptors (n,128) f32
    landmark_id: np.ndarray       # (n,) index of the world point behind each feature, -1 for clutter
    world_to_cam: np.ndarray      # (4,4) ground truth


@dataclass
class Sequence:
    K: np.ndarray
    landmarks: np.ndarray         # (L,3) world points
    frames: List[Frame]


def _kps(xy: np.ndarray) -> np.ndarray:
    out = np.empty(len(xy), dtype=object)
    for i, p in enumerate(xy):
        out[i] = Keypoint(np.asarray(p, dtype=np.float32), 1.0, 0.0, 1.0, 0, -1)
    return out


def synthetic_sequence(n_frames: int, n_landmarks: int, K: np.ndarray, size, seed: int = 0,
                       noise_px: float = 0.3, desc_noise: float = 6.0, clutter: float = 0.3,
                       step: float = 0.25) -> Sequence:
    """A camera moving forward with a slow yaw through a cloud of landmarks."""
    rng = np.random.default_rng(seed)
    w, h = size
    z = rng.uniform(6.0, 45.0, n_landmarks)
    u, v = rng.uniform(0, w, n_landmarks), rng.uniform(0, h, n_landmarks)
    X = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    desc = synth.sift_like(n_landmarks, rng)
    frames = []
    for f in range(n_frames):
        R = synth._rodrigues(np.array([0.0, 0.004 * f, 0.001 * f]))
        t = -R @ np.array([0.02 * f, 0.0, step * f])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        Xc = X @ R.T + t
        uv = Xc[:, :2] / Xc[:, 2:3] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
        vis = (Xc[:, 2] > 1.0) & (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
        ids = np.nonzero(vis)[0]
        xy = uv[ids] + rng.normal(0, noise_px, (len(ids), 2))
        d = np.clip(np.rint(desc[ids] + rng.normal(0, desc_noise, (len(ids), 128))), 0, 255).astype(np.float32)
        nc = int(clutter * len(ids))
        xy = np.concatenate([xy, np.stack([rng.uniform(0, w, nc), rng.uniform(0, h, nc)], 1)])
        d = np.concatenate([d, synth.sift_like(nc, rng)])
        lid = np.concatenate([ids, -np.ones(nc, dtype=np.int64)])
        perm = rng.permutation(len(lid))
        frames.append(Frame(Features(_kps(xy[perm]), d[perm]), lid[perm], T))
    return Sequence(K, X, frames)


def run_init(seq: Sequence, i: int = 0, j: int = 1, dataset: int = 0):
    """2D-2D initialisation between frames i and j -> (M_camj_to_world (4,4), P_world (L,3), p_i, p_j,
    matched Features of i and j)."""
    params.which_dataset(dataset)
    mf_i, mf_j = match_features(seq.frames[i].features, seq.frames[j].features)
    out = twoDtwoD_from_matches(mf_i.get_positions(), mf_j.get_positions(), seq.K)
    return out + (mf_i, mf_j)


def run_continuous(seq: Sequence, keyframe: int = 0, dataset: int = 1):
    """Localise every later frame against the keyframe's landmark-bearing features.
    -> list of (cam_to_world (4,4), inlier index array | None, matched 2D (M,2), matched 3D (M,3))."""
    params.which_dataset(dataset)
    kf = seq.frames[keyframe]
    have = kf.landmark_id >= 0                      # features with a (ground-truth) landmark
    kf_feats = Features(kf.features.keypoints[have], kf.features.descriptors[have])
    kf_xyz = seq.landmarks[kf.landmark_id[have]]
    # landmark of a keyframe feature, looked up through its position (match_features returns gathered
    # Features, not indices, exactly like the reference)
    key = {tuple(np.round(k.pt, 4)): n for n, k in enumerate(kf_feats.keypoints)}
    out = []
    for fr in seq.frames[keyframe + 1:]:
        mf_k, mf_f = match_features(kf_feats, fr.features)
        rows = np.array([key[tuple(np.round(k.pt, 4))] for k in mf_k.keypoints], dtype=np.int64)
        pts2d = mf_f.get_positions().astype(np.float32)
        pts3d = kf_xyz[rows]
        T, inl = pnp(pts2d, pts3d, seq.K)
        out.append((T, inl, pts2d, pts3d))
    return out
