#!/usr/bin/env python3
"""One-off validation: vo_ransac_batch(kind = F) with pruned blocks (first_block > 0) against the single exact block over random
batch shapes, scene statistics, thresholds, confidences and first blocks -- result record, F and mask must be identical.
usage: pruned_fuzz.py [first seed] [number of seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_odometry_amd import ops, synth  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 9))
    H = int(rng.choice([60, 300, 1000, 2500, 4096]))
    N = int(rng.choice([40, 300, 1500, 5000]))
    thr = float(10 ** rng.uniform(-1.3, 1.0))
    conf = float(rng.choice([0.9, 0.99, 0.9999]))
    fb = int(rng.choice([1, 7, 16, 64, 100, 256, 1000]))
    pts = np.zeros((B, N, 4), np.float32); nd = np.zeros(B, np.int32); subs = np.zeros((B, H, 7), np.int32)
    for b in range(B):
        n = int(rng.integers(15, N + 1))
        sc = synth.two_view_scene(n, seed=seed * 16 + b, outlier_frac=float(rng.uniform(0, 0.8)), noise_px=float(10 ** rng.uniform(-2, 0.5)))
        p = np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)
        if rng.uniform() < 0.15:                             # repeated correspondences: many equal counts
            p = np.tile(p[:max(8, n // 5)], (6, 1))[:n]
        pts[b, :n] = p; nd[b] = n
        subs[b] = ops.gen_subsets(n, 7, H, p)
    outs = []
    for f in (0, fb):
        res, best, _, mask = ops.ransac_batch(ops.KIND_FUND, d(subs), d(pts), thr, conf, H, n_dev=d(nd), first_block=f)
        outs.append((res.cpu().numpy(), best.cpu().numpy(), mask.cpu().numpy()))
    (r0, f0, m0), (r1, f1, m1) = outs
    ok = np.array_equal(r0, r1)
    for b in range(B):
        if r0[b, 0] >= 0:
            ok = ok and np.array_equal(f0[b], f1[b]) and np.array_equal(m0[b, :nd[b]], m1[b, :nd[b]])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, dict(B=B, H=H, N=N, thr=thr, conf=conf, fb=fb), r0.tolist(), r1.tolist(), flush=True)
    if (seed - s0) % 25 == 24:
        print("seeds done", seed - s0 + 1, "mismatching", bad, flush=True)
print("configurations", ns, "mismatching", bad)
sys.exit(1 if bad else 0)
