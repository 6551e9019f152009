#!/usr/bin/env python3
"""Debug only: wave-time per phase of the MFMA F count kernel over one bench step, from a library built with
-DVO_SCORE_MF_PROF (scripts/mf_variants.sh prof:"-DVO_SCORE_MF_PROF"; VO_HIP_LIB=ablate/libvo_mf_prof.so)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from visual_odometry_amd import _lib  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_lib.require_device(0)
torch.cuda.set_device(0)
ps = bench.PairSet(cfg, pairs, torch.device("cuda", 0), 0)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
bench.step(ps, cfg, None); torch.cuda.synchronize()
L.vo_score_stats(buf, 1)
bench.step(ps, cfg, None); torch.cuda.synchronize()
L.vo_score_stats(buf, 0)
v = [int(buf[i]) for i in range(8)]
names = ["stage+barriers", "tier 1a (MFMA+epilogue)", "queue push", "flush1 (tier 1b)", "drain2 (tier 2)", "end-of-chunk flush"]
tot = v[6]
print(f"waves {v[7]}, ticks per wave {tot / max(v[7], 1):.0f}")
for n, x in zip(names, v[:6]):
    print(f"  {n:28s} {x / tot:6.1%}")
print(f"  {'other (set-up, tail)':28s} {1 - sum(v[:6]) / tot:6.1%}")
