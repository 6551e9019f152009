#!/usr/bin/env python3
"""Frames per second of harness.run_tracking (reference-shaped single-pair entry points: host arrays in and out,
one match + one PnP per frame) over a synthetic 1000-frame corridor.  usage: bench_tracking.py [frames] [features]"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import _lib, harness, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 900
_lib.require_device(0)
torch.cuda.set_device(0)
seq = harness.Corridor(n, synth.K_MALAGA, synth.IMAGE_SIZE["malaga"], seed=11, per_frame=per)
t0 = time.perf_counter()
for f in range(n):
    seq.frame(f)
t_gen = time.perf_counter() - t0
with contextlib.redirect_stdout(io.StringIO()):            # match_features prints one line per call, like the reference
    harness.run_tracking(seq, frames=range(5))
    t0 = time.perf_counter()
    res = harness.run_tracking(seq, min_tracked=300)
    t_all = time.perf_counter() - t0
errs = [np.abs(r["T"][:3, 3] - np.linalg.inv(seq.world_to_cam(r["frame"]))[:3, 3]).max() for r in res]
print(json.dumps({"frames": n, "features_per_frame": per, "seconds": t_all, "frame_synthesis_seconds": t_gen,
                  "frames_per_s_excluding_synthesis": (n - 1) / max(t_all - t_gen, 1e-9), "handovers": sum(r["handover"] for r in res),
                  "median_state": int(np.median([r["n_state"] for r in res])), "median_inliers": int(np.median([r["n_inliers"] for r in res])),
                  "max_translation_error": float(max(errs))}))
