#!/usr/bin/env python3
"""Measurement only: worst deviation of the MFMA tier's d~ from the f64 value over every evaluation of one bench step, in
units of the tier's error budget (eta = 48 units of 2^-24 S~), from a library built with -DVO_SCORE_MF_CHECK
(scripts/mf_variants.sh check:"-DVO_SCORE_MF_CHECK"; VO_HIP_LIB=ablate/libvo_mf_check.so).  -> profiles/r3_mfma_tier_error.txt"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from visual_odometry_amd import _lib  # noqa: E402

cfg = bench.CONFIGS["cfg4"]
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_lib.require_device(0)
torch.cuda.set_device(0)
ps = bench.PairSet(cfg, pairs, torch.device("cuda", 0), 0)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
L.vo_score_stats(buf, 1)
bench.step(ps, cfg, None); torch.cuda.synchronize()
L.vo_score_stats(buf, 0)
worst, n, over4 = int(buf[0]) / 1000.0, int(buf[1]), int(buf[2])
print(f"{pairs} cfg4 frame pairs, {n} evaluations of the MFMA tier checked against f64:")
print(f"worst |d~ - kappa d| = {worst:.2f} units of 2^-24 S~ (budget eta = 48 units); {over4} evaluations ({over4 / max(n, 1):.2e}) above 4 units")
