#!/usr/bin/env python3
"""One-off validation: the fuzz of tests/test_ransac_gpu.py::test_score_fund_fuzz over many more seeds (random scenes, image
scales 1e-2..1e2, model scalings 1e-30..1e30, thresholds 0.01..30 px); every count must equal the oracle's.
usage: fscore_fuzz.py [first seed] [number of seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from visual_odometry_amd import ops, synth  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = evals = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(33, 6000))
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=float(rng.uniform(0, 0.6)), noise_px=float(rng.uniform(0, 2)),
                              K=synth.K_MALAGA if seed % 2 else synth.K_KITTI, size=(1024, 768) if seed % 2 else (1241, 376))
    scale = float(10 ** rng.uniform(-2, 2))
    p1 = (sc["p1"] * scale).astype(np.float32); p2 = (sc["p2"] * scale).astype(np.float32)
    subs = oracle.gen_subsets(n, 7, 120, p1, p2)
    models = np.array([F.reshape(9) for s_ in subs for F in oracle.run7point(p1[s_], p2[s_])])
    models = models * (10 ** rng.uniform(-30, 30, (len(models), 1)))
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    for thr in 10 ** rng.uniform(-2, 1.5, 3) * scale:
        cnt = ops.score_fund(d(models), d(pts4), float(thr)).cpu().numpy()
        ref = np.array([oracle.find_inliers(oracle.fund_error(m, p1, p2), float(thr))[0] for m in models])
        evals += len(models) * n
        if not np.array_equal(cnt, ref):
            bad += 1
            print("MISMATCH seed", seed, "thr", thr, "scale", scale, "models", np.nonzero(cnt != ref)[0][:5], flush=True)
    if (seed - s0) % 20 == 19:
        print("seeds done", seed - s0 + 1, "evaluations", evals, "mismatching launches", bad, flush=True)
print("evaluations", evals, "mismatching launches", bad)
sys.exit(1 if bad else 0)
