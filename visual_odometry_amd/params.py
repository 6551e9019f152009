"""Hot-path parameters under the reference's attribute names, without its ``import cv2``.

The reference keeps one nested class per dataset and switches between them through a module
global (src/params.py:7-132).  Only the groups the drop-in entry points read are carried here
(ratio-test threshold, F-RANSAC, PnP-RANSAC, depth limits, baseline frames); SIFT/KLT detector
settings belong to image-domain stages outside this package.  Attribute paths are unchanged, e.g.
``get_params().CONT_PARAMS.PNP_RANSAC_PARAMS.NUM_ITERATIONS``.
"""
from types import SimpleNamespace


class FeatureDetector:
    KLT = 0
    SIFT = 1


# dataset id (src/utils/dataloader.py:20-24) -> values (src/params.py:8-120)
#            ratio  F: thr  conf   iters   PnP: thr conf  iters  baseline frames
_TABLE = {
    0: ("KITTI", 0.8, (0.1, 0.9999, 100), (1, 0.9999, 100), [0, 10]),
    1: ("MALAGA", 0.6, (0.1, 0.9999, 50), (1, 0.9999, 10000), [0, 5]),
    2: ("PARKING", 0.8, (0.1, 0.9999, 10000), (5, 0.9999, 10000), [0, 6]),
}


def _ransac(thr, conf, iters):
    return SimpleNamespace(THRESHOLD=thr, CONFIDENCE=conf, NUM_ITERATIONS=iters)


def _build(name, ratio, f, p, frames):
    return SimpleNamespace(
        NAME=name,
        SIFT_PARAMS=SimpleNamespace(MATCHING_THRESHOLD=ratio),
        INIT_PARAMS=SimpleNamespace(RANSAC_PARAMS_F=_ransac(*f), BASELINE_FRAME_INDICES=frames,
                                    MATCHER=FeatureDetector.SIFT, MAX_DEPTH_DISTANCE=50),
        CONT_PARAMS=SimpleNamespace(PNP_RANSAC_PARAMS=_ransac(*p), MAX_DEPTH_DISTANCE=50),
    )


KITTIParams, MALAGAParams, PARKINGParams = (_build(*_TABLE[k]) for k in (0, 1, 2))
DATASET = 0


def get_params():
    """0 -> KITTI, 1 -> MALAGA, anything else -> PARKING (as the reference does)."""
    return {0: KITTIParams, 1: MALAGAParams}.get(DATASET, PARKINGParams)


def which_dataset(dataset: int):
    global DATASET
    DATASET = dataset
