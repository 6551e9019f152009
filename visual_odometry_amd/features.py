"""Descriptor matching drop-in for reference src/features.py:15-30 (``Keypoint``, ``Features``)
and :95-124 (``match_features``): brute-force 2-nearest-neighbour search in L2 over 128-D SIFT
descriptors plus Lowe's ratio test, on the MI355X.

Device path per call: f32 descriptors -> int8 (value-128) + squared norms
(``vo_desc_f32_to_i8``), exact integer 2-NN on int8 MFMA (``vo_knn2_l2_i8``), ratio test
(``vo_ratio_test``); the host then gathers the surviving rows, as the reference's list
comprehensions do.  Semantics follow OpenCV 4.6.0 ``BFMatcher().knnMatch(q, t, k=2)``
(SURVEY.md Appendix A.1): ties go to the lower train index, the ratio test is strict.
"""
from __future__ import annotations

from enum import Enum
from typing import NamedTuple, Tuple

import numpy as np
import torch

from . import _lib, ops
from .params import get_params


class FeatureDetector(Enum):
    KLT = 0
    SIFT = 1


class Keypoint(NamedTuple):
    """Field-for-field cv2.KeyPoint (reference src/features.py:15-21)."""
    pt: np.ndarray
    size: float
    angle: float
    response: float
    octave: int
    class_id: int


class Features(NamedTuple):
    keypoints: np.ndarray
    descriptors: np.ndarray

    def get_positions(self) -> np.ndarray:
        return np.array([k.pt for k in self.keypoints])


def _as_descriptors(des, name: str) -> np.ndarray:
    if des is None:
        raise ValueError(f"{name}: descriptors are None (no features detected)")
    a = np.asarray(des)
    if a.ndim != 2 or a.shape[1] != 128:
        raise ValueError(f"{name}: expected (N,128) SIFT descriptors, got shape {a.shape}")
    return np.ascontiguousarray(a, dtype=np.float32)


def knn2_arrays(des_q, des_t) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (idx (nq,2) i32, dist (nq,2) f32, d2 (nq,2) i32): BFMatcher(NORM_L2).knnMatch(q,t,k=2)
    as arrays.  Missing neighbours (nt < 2) have idx -1, dist FLT_MAX."""
    q = _as_descriptors(des_q, "query")
    t = _as_descriptors(des_t, "train")
    nq, nt = q.shape[0], t.shape[0]
    _lib.require_device(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    dev = torch.device("cuda", torch.cuda.current_device())
    if nq == 0 or nt == 0:
        idx = np.full((nq, 2), -1, np.int32)
        return idx, np.full((nq, 2), np.finfo(np.float32).max, np.float32), np.full((nq, 2), 2**31 - 1, np.int32)
    if nt > 262144:
        raise ValueError("BFMatcher supports at most 262144 train descriptors")
    both = torch.from_numpy(np.concatenate([q, t], axis=0)).to(dev)
    i8, norm, flags = ops.desc_to_i8(both)
    idx_d, d2_d = ops.knn2(i8[:nq], norm[:nq], i8[nq:], norm[nq:])
    out = torch.cat([idx_d, d2_d], dim=1).cpu().numpy()
    fl = int(flags.item())
    if fl & _lib.VO_DESC_BAD_VALUE:
        raise _lib.VoError("descriptors must hold integers in [0,255] (OpenCV SIFT output); the int8 "
                           "MFMA matcher is exact only for such inputs")
    if fl & _lib.VO_DESC_BAD_NORM:
        raise _lib.VoError("a descriptor has L2 norm > 1024: integer distance keys could order two "
                           "neighbours differently from OpenCV's f32 sqrt (SURVEY.md A.1)")
    idx, d2 = out[:, :2].copy(), out[:, 2:].copy()
    dist = np.sqrt(d2.astype(np.float32))          # exact integers < 2^24 -> correctly rounded sqrtf
    dist[idx < 0] = np.finfo(np.float32).max
    return idx, dist, d2


def match_features_indices(des_i, des_j, threshold: float) -> Tuple[np.ndarray, np.ndarray]:
    """Row indices (query, train) of the matches that survive ``m.distance < T * n.distance``,
    in ascending query order."""
    q = _as_descriptors(des_i, "query")
    t = _as_descriptors(des_j, "train")
    if 0 < q.shape[0] and t.shape[0] < 2:
        # the reference unpacks every knnMatch entry into (m, n)
        raise ValueError(f"not enough values to unpack (expected 2, got {t.shape[0]})")
    idx, dist, _ = knn2_arrays(q, t)
    d = dist.astype(np.float64)
    keep = d[:, 0] < float(threshold) * d[:, 1]
    qi = np.nonzero(keep)[0].astype(np.int32)
    return qi, idx[qi, 0].astype(np.int32)


def _take(seq, index: np.ndarray):
    """``np.array([seq[i] for i in index])`` of the reference, for descriptor rows (numeric
    ndarray -> fancy index) and for keypoint sequences (tuple/list/object array of cv2.KeyPoint or
    ``Keypoint`` -> 1-D object array, one element per match)."""
    if isinstance(seq, np.ndarray) and seq.dtype != object:
        return seq[index]
    out = np.empty(len(index), dtype=object)
    for k, i in enumerate(index):
        out[k] = seq[int(i)]
    return out


def match_features(feature_set_i: Features, feature_set_j: Features) -> Tuple[Features, Features]:
    threshold = get_params().SIFT_PARAMS.MATCHING_THRESHOLD
    qi, ti = match_features_indices(feature_set_i.descriptors, feature_set_j.descriptors, threshold)
    print(f"#SIFT Matches filtered: {len(qi)}")
    des_i = np.asarray(feature_set_i.descriptors)
    des_j = np.asarray(feature_set_j.descriptors)
    return (Features(_take(feature_set_i.keypoints, qi), _take(des_i, qi)),
            Features(_take(feature_set_j.keypoints, ti), _take(des_j, ti)))
