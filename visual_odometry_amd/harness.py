"""Synthetic stand-ins for BASELINE configs 2 and 3, driving the drop-in entry points with the call
pattern of the reference's pipeline (there are no datasets and no SIFT detector here):

* ``run_init``       -- src/twoDtwoD.py:45-79: match two frames' descriptors, F-RANSAC, pose + landmarks
                        (config 2: KITTI intrinsics and parameters);
* ``run_continuous`` -- the per-frame localisation of src/tracking.py:382-414 with descriptor matching
                        in place of KLT, as in src/sift_test.py:104-119: match the keyframe's
                        landmark-bearing features against each new frame, then ``pnp`` on the matched
                        (2D, 3D) pairs (config 3: Malaga intrinsics and parameters).

* ``run_tracking``   -- the frame-to-frame loop of src/tracking.py:382-414 over a long sequence (``Corridor``,
                        frames made on demand): the tracked keypoints and their landmarks are carried from
                        frame t to t+1 (descriptor matching stands in for ``klt``), ``pnp``, the inlier and
                        distance masks of lines 410-425, and a key-frame hand-over when the track has thinned
                        out (stand-in for TrackManager.get_new_landmarks, which is not on the hot path).

A "frame" is what SIFT would have produced for it: keypoint positions and 128-D integer descriptors.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional

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


# ---------------------------------------------------------------------------- long sequences
class Corridor:
    """A camera driving `n_frames * step` metres down a corridor of landmarks; frame f is generated on demand
    (deterministically from (seed, f)), so a 1000-frame sequence costs no memory."""

    def __init__(self, n_frames: int, K: np.ndarray, size, seed: int = 0, per_frame: int = 1500, step: float = 0.25,
                 noise_px: float = 0.3, desc_noise: float = 6.0, clutter: float = 0.3, z_far: float = 45.0):
        self.n_frames, self.K, self.size, self.seed, self.step = n_frames, np.asarray(K, dtype=np.float64), size, seed, step
        self.noise_px, self.desc_noise, self.clutter, self.z_far = noise_px, desc_noise, clutter, z_far
        rng = np.random.default_rng([seed, 1 << 20])
        w, h = size
        hx, hy = 0.5 * w / K[0, 0] * z_far * 1.15, 0.5 * h / K[1, 1] * z_far * 1.15       # half-extent of the slab
        length = n_frames * step + z_far + 5.0
        frustum = (w / K[0, 0]) * (h / K[1, 1]) * z_far ** 3 / 3.0
        n_land = int(per_frame * (2 * hx) * (2 * hy) * length / frustum)
        self.landmarks = np.stack([rng.uniform(-hx, hx, n_land), rng.uniform(-hy, hy, n_land),
                                   rng.uniform(0.0, length, n_land)], axis=1)
        self.desc = synth.sift_like(n_land, rng)

    def world_to_cam(self, f: int) -> np.ndarray:
        R = synth._rodrigues(np.array([0.0, 0.02 * np.sin(0.01 * f), 0.01 * np.sin(0.013 * f)]))
        c = np.array([0.5 * np.sin(0.02 * f), 0.0, self.step * f])                       # camera centre in the world
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = -R @ c
        return T

    def frame(self, f: int) -> Frame:
        rng = np.random.default_rng([self.seed, f])
        K, (w, h) = self.K, self.size
        T = self.world_to_cam(f)
        Xc = self.landmarks @ T[:3, :3].T + T[:3, 3]
        front = np.nonzero((Xc[:, 2] > 1.0) & (Xc[:, 2] < self.z_far))[0]
        uv = Xc[front, :2] / Xc[front, 2:3] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
        vis = (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
        ids, uv = front[vis], uv[vis]
        xy = uv + rng.normal(0, self.noise_px, uv.shape)
        d = np.clip(np.rint(self.desc[ids] + rng.normal(0, self.desc_noise, (len(ids), 128))), 0, 255).astype(np.float32)
        nc = int(self.clutter * len(ids))
        xy = np.concatenate([xy, np.stack([rng.uniform(0, w, nc), rng.uniform(0, h, nc)], 1)])
        d = np.concatenate([d, synth.sift_like(nc, rng)])
        lid = np.concatenate([ids, -np.ones(nc, dtype=np.int64)])
        perm = rng.permutation(len(lid))
        return Frame(Features(_kps(xy[perm]), d[perm]), lid[perm], T)


def _with_row_ids(feats: Features) -> Features:
    """The same features with Keypoint.class_id = row number, so that rows can be followed through
    match_features (which returns gathered Features, not indices, like the reference)."""
    kps = np.empty(len(feats.keypoints), dtype=object)
    for i, k in enumerate(feats.keypoints):
        kps[i] = k._replace(class_id=i)
    return Features(kps, feats.descriptors)


def run_tracking(seq: Corridor, frames: Optional[Iterable[int]] = None, dataset: int = 1, min_tracked: int = 400,
                 landmark_noise: float = 0.01, record: Iterable[int] = (), new_landmarks: str = "truth",
                 track_length: int = 15) -> List[Dict]:
    """Frame-to-frame localisation with key-frame hand-over (src/tracking.py:382-425's call pattern).

    The state is the tracked keypoints (with descriptors) and their world landmarks.  Per frame: match the state
    against the new frame (for ``klt`` + ``mask_klt``), carry the landmarks over, ``pnp``, keep the RANSAC
    inliers, drop landmarks behind the camera or farther than 100 (lines 416-425).  When fewer than
    `min_tracked` remain, the frame becomes a key frame and the state is replaced:
      * new_landmarks="truth": by the frame's landmark-bearing features and their world points (ground truth +
        `landmark_noise`) -- isolates the match + PnP path from map quality;
      * new_landmarks="triangulate": by what TrackManager.get_new_landmarks would add (src/tracking.py:150-172): all
        features of the frame `track_length` frames back (a track started there, MIN_TRACK_LENGTH's role: enough
        baseline) are matched against this frame and triangulated between the two ESTIMATED camera poses
        (``landmarks.triangulate_track`` -> ``vo_triangulate``), depth-filtered; until that many frames exist the
        state is re-seeded from ground truth, which also fixes the scale (as the 2D-2D initialisation plus a known
        baseline would).  The trajectory then drifts like a real VO's.
    -> one dict per frame: T (cam_to_world), n_state, n_matched, n_inliers, n_kept, handover; frames listed in
    `record` also keep the inputs (state / frame descriptors, pts2d, pts3d, inlier indices) for an external check."""
    from .landmarks import triangulate_track
    if new_landmarks not in ("truth", "triangulate"):
        raise ValueError("new_landmarks must be 'truth' or 'triangulate'")
    params.which_dataset(dataset)
    max_depth = params.get_params().CONT_PARAMS.MAX_DEPTH_DISTANCE
    frames = list(range(seq.n_frames) if frames is None else frames)
    record = set(record)
    rng = np.random.default_rng([seq.seed, 1 << 21])

    def key_frame(fr: Frame):
        have = fr.landmark_id >= 0
        xyz = seq.landmarks[fr.landmark_id[have]] + rng.normal(0, landmark_noise, (int(have.sum()), 3))
        return _with_row_ids(Features(fr.features.keypoints[have], fr.features.descriptors[have])), xyz

    first = seq.frame(frames[0])
    state, landmarks = key_frame(first)
    past = [(first.features, np.linalg.inv(first.world_to_cam))]              # recent frames: all features + estimated pose
    out: List[Dict] = []
    for f in frames[1:]:
        fr = seq.frame(f)
        mf_prev, mf_cur = match_features(state, fr.features)
        rows = np.array([k.class_id for k in mf_prev.keypoints], dtype=np.int64)
        pts2d = mf_cur.get_positions().astype(np.float32).reshape(-1, 2)
        pts3d = landmarks[rows]
        T, inl = pnp(pts2d, pts3d, seq.K)
        rec = dict(frame=f, T=T, n_state=len(state.keypoints), n_matched=len(rows), n_inliers=0, n_kept=0, handover=False)
        if f in record:
            rec.update(state_desc=state.descriptors.copy(), frame_desc=fr.features.descriptors.copy(), rows=rows,
                       pts2d=pts2d, pts3d=pts3d.copy(), inliers=None if inl is None else np.asarray(inl).copy())
        keep = np.zeros(len(rows), dtype=bool)
        if inl is not None:
            keep[np.squeeze(inl, axis=1) if np.ndim(inl) == 2 else inl] = True
            rec["n_inliers"] = int(keep.sum())
            P_cam = (np.linalg.inv(T) @ np.c_[pts3d, np.ones(len(pts3d))].T).T
            keep &= (P_cam[:, 2] > 0) & (np.linalg.norm(P_cam, axis=1) < 100)       # tracking.py:421-423 (4-vector norm, as there)
        rec["n_kept"] = int(keep.sum())
        if rec["n_kept"] < min_tracked:
            rec["handover"] = True
            if new_landmarks == "truth" or len(past) < track_length:
                state, landmarks = key_frame(fr)
            else:
                old_feats, old_pose = past[-track_length]
                m_old, m_cur = match_features(old_feats, fr.features)
                P_world, kp_j, mask = triangulate_track(seq.K, old_pose, T, m_old.get_positions().reshape(-1, 2),
                                                        m_cur.get_positions().reshape(-1, 2), max_depth)
                rec["n_new_landmarks"] = int(mask.sum())
                state = _with_row_ids(Features(m_cur.keypoints[mask], m_cur.descriptors[mask]))
                landmarks = P_world[:, :3]
        else:
            state = _with_row_ids(Features(mf_cur.keypoints[keep], mf_cur.descriptors[keep]))
            landmarks = pts3d[keep]
        past.append((fr.features, T))
        if len(past) > track_length:
            past.pop(0)
        out.append(rec)
    return out
