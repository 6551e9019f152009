#!/usr/bin/env python3
"""Debug aid: F scoring counts of one bench-sized launch against the oracle, at a coordinate scale; prints where they differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from visual_odometry_amd import ops, synth  # noqa: E402
from visual_odometry_amd.features import match_features_indices  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
dev = torch.device("cuda", 0)
fp = synth.frame_pair(8192, 8192, seed=1234, K=synth.K_MALAGA, size=synth.IMAGE_SIZE["malaga"])
qi, ti = match_features_indices(fp["q"], fp["t"], 0.8)
pts4 = (np.concatenate([fp["kp_q"][qi], fp["kp_t"][ti]], 1) * np.float32(scale)).astype(np.float32)
n, H, thr = len(pts4), 4096, 0.1 * scale
subs = ops.gen_subsets(n, 7, H, pts4)
p = torch.from_numpy(pts4).to(dev)
models, nm = ops.solve_7pt(torch.from_numpy(subs).to(dev), p)
valid = (torch.arange(3, device=dev)[None, :] < nm[:, None]).to(torch.uint8).reshape(-1)
counts = ops.score_fund(models.reshape(-1, 9), p, thr, valid=valid).cpu().numpy().reshape(H, 3)
models, nm = models.cpu().numpy(), nm.cpu().numpy()
bad = 0
badF = []
for h in range(H):
    for k in range(nm[h]):
        err = oracle.fund_error(models[h, k].reshape(3, 3), pts4[:, :2], pts4[:, 2:])
        e = oracle.find_inliers(err, thr)[0]
        if e != counts[h, k]:
            bad += 1
            badF.append(models[h, k])
            if bad <= 12:
                F = models[h, k]
                print(f"model {h},{k}: gpu {counts[h, k]} oracle {e}  max|F| {np.abs(F).max():.3g} F={np.array2string(F, precision=3)}")
np.save(os.path.join(ROOT, "gpurun_out", f"fscore_bad_{scale:g}.npy"), np.array(badF))
print("mismatching models:", bad, "of", int(nm.sum()))
