#!/usr/bin/env python3
"""Debug only: tier statistics of the F count kernel on one bench step, from a library built with -DVO_SCORE_STATS
(VO_HIP_LIB=ablate/libvo_stats.so).  Prints candidates / trips / undecided per 64-correspondence block."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from visual_odometry_amd import _lib  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_lib.require_device(0)
torch.cuda.set_device(0)
ps = bench.PairSet(cfg, pairs, torch.device("cuda", 0), 0)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
bench.step(ps, cfg, None); torch.cuda.synchronize()
L.vo_score_stats(buf, 1)
bench.step(ps, cfg, None); torch.cuda.synchronize()
L.vo_score_stats(buf, 0)
cand, trips, und, blocks, trips2, exact, flushes = [int(buf[i]) for i in range(7)]
print(f"wave-blocks {blocks}; candidates/eval {cand / (blocks * 64 * 64):.4f}; tier-1b flushes (128 pairs) per block {trips / blocks:.2f}; "
      f"undecided/eval {und / (blocks * 64 * 64):.5f}; owner-lane tier-2 trips per block {trips2 / blocks:.2f}; "
      f"pooled tier-2 flushes per block {flushes / blocks:.3f}; exact evaluations per undecided {exact / max(und, 1):.4f}")
