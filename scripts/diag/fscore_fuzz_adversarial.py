#!/usr/bin/env python3
"""One-off validation of the F count kernel's matrix-core tier on inputs the scene fuzz does not produce: coordinates of
mixed dynamic range inside one problem (a few points near the 2^20 limit set the power-of-two scaling, the rest are tiny
against it: f16 lo halves go subnormal or vanish), models with unbalanced entries (per-entry scales 1e-3..1e3), sparse and
rank-one models, points exactly on their epipolar line, thresholds 1e-3..1e2 of the image scale.  Every count must equal
the oracle's.   usage: fscore_fuzz_adversarial.py [first seed] [number of seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from visual_odometry_amd import ops, synth  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = evals = 0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(40, 3000))
    W = float(10 ** rng.uniform(-1, 3))
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=float(rng.uniform(0, 0.5)), noise_px=float(rng.uniform(0, 1)),
                              K=synth.K_MALAGA, size=(1024, 768))
    p1 = sc["p1"] * (W / 1024.0); p2 = sc["p2"] * (W / 1024.0)
    subs = oracle.gen_subsets(n, 7, 40, p1.astype(np.float32), p2.astype(np.float32))
    geo = [F.reshape(9) for s_ in subs for F in oracle.run7point(p1[s_].astype(np.float32), p2[s_].astype(np.float32))]
    # a few far points: the problem-wide bound becomes up to 2^19, everything else is tiny against it
    kind = seed % 4
    if kind != 0:
        far = rng.choice(n, size=max(1, n // 50), replace=False)
        big = 2.0 ** rng.uniform(4, 19 - np.log2(max(W, 1.0)), size=(len(far), 1))
        which = rng.integers(0, 3)
        if which in (0, 2): p1[far] = p1[far] * big
        if which in (1, 2): p2[far] = p2[far] * big
    if kind == 2:                                  # some points at / next to the origin
        z = rng.choice(n, size=5, replace=False)
        p1[z[:3]] = 0.0; p2[z[2:]] = rng.uniform(-1e-6, 1e-6, (3, 2)) * W
    p1 = p1.astype(np.float32); p2 = p2.astype(np.float32)
    models = list(geo)
    for _ in range(40):
        M = rng.standard_normal((3, 3)) * 10 ** rng.uniform(-3, 3, (3, 3))
        models.append(M.reshape(9))
    for _ in range(10):                            # sparse: random entries zeroed
        M = rng.standard_normal((3, 3)) * (rng.uniform(0, 1, (3, 3)) < 0.5)
        models.append(M.reshape(9))
    for _ in range(10):                            # rank one
        models.append(np.outer(rng.standard_normal(3), rng.standard_normal(3)).reshape(9))
    for g in geo[:10]:                             # a true model with its last column / row perturbed by an ulp-scale amount
        M = g.copy(); M[rng.integers(0, 9)] *= 1 + rng.uniform(-1e-7, 1e-7); models.append(M)
    models = np.array(models, dtype=np.float64)
    # points exactly on their epipolar line under the first model: move y2 onto the line (f32 rounding leaves them near it)
    if len(geo):
        F = geo[0].reshape(3, 3)
        for i in rng.choice(n, size=min(20, n), replace=False):
            l = F @ np.array([p1[i, 0], p1[i, 1], 1.0])
            if abs(l[1]) > 1e-12 * (abs(l[0]) + abs(l[2]) + 1e-300):
                y = -(l[0] * float(p2[i, 0]) + l[2]) / l[1]
                if abs(y) < 2.0 ** 19: p2[i, 1] = np.float32(y)
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    for thr in 10 ** rng.uniform(-3, 2, 3) * W:
        cnt = ops.score_fund(d(models), d(pts4), float(thr)).cpu().numpy()
        ref = np.array([oracle.find_inliers(oracle.fund_error(m, p1, p2), float(thr))[0] for m in models])
        evals += len(models) * n
        if not np.array_equal(cnt, ref):
            bad += 1
            w = np.nonzero(cnt != ref)[0]
            print("MISMATCH seed", seed, "kind", kind, "thr", thr, "W", W, "models", w[:5], cnt[w[:5]], ref[w[:5]], flush=True)
    if (seed - s0) % 20 == 19:
        print("seeds done", seed - s0 + 1, "evaluations", evals, "mismatching launches", bad, flush=True)
print("evaluations", evals, "mismatching launches", bad)
sys.exit(1 if bad else 0)
