"""Diagnostic (GPU box): how tight is vo_score_fund_bound on a bench-shaped problem (4096 samples x <= 3 models, thr 0.1 px),
and how many hypotheses would stay contenders against the best count of the first 256 iterations."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, synth
dev = torch.device("cuda:0")
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for seed in (1, 2):
    n = 4915
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=0.25, noise_px=0.5)
    pts4 = np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)
    subs = ops.gen_subsets(n, 7, 4096, pts4)
    models, nm = ops.solve_7pt(d(subs), d(pts4))
    nm = nm.cpu().numpy()
    valid = (np.arange(3)[None, :] < nm[:, None]).reshape(-1)
    m = models.reshape(-1, 9)[torch.from_numpy(valid).to(dev)]
    for thr in (0.1, 1.0):
        cnt = ops.score_fund(m, d(pts4), thr).cpu().numpy()
        ub = ops.score_fund_bound(m, d(pts4), thr).cpu().numpy()
        it = np.repeat(np.arange(4096), 3)[valid]
        best256 = cnt[it < 256].max()
        print(f"seed {seed} thr {thr}: models {len(cnt)}, best count {cnt.max()} (first 256 iterations: {best256}), mean count {cnt.mean():.1f}, "
              f"mean bound {ub.mean():.1f}, bound == n for {(ub == n).sum()} models, contenders vs best256: {(ub[it >= 256] > best256).sum()}, "
              f"percentiles of bound - count: {np.percentile(ub - cnt, [50, 90, 99, 100])}", flush=True)

# how many contenders a perfectly informed sequential pruning would leave (bound against the running best of the TRUE counts at
# each model's own iteration) against the static threshold of the first 256 iterations -- the room any threshold refinement has
for seed in (1, 2, 3):
    n = 4915
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=0.25, noise_px=0.5)
    pts4 = np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)
    subs = ops.gen_subsets(n, 7, 4096, pts4)
    models, nm = ops.solve_7pt(d(subs), d(pts4))
    nm = nm.cpu().numpy()
    valid = (np.arange(3)[None, :] < nm[:, None]).reshape(-1)
    m = models.reshape(-1, 9)[torch.from_numpy(valid).to(dev)]
    cnt = ops.score_fund(m, d(pts4), 0.1).cpu().numpy()
    ub = ops.score_fund_bound(m, d(pts4), 0.1).cpu().numpy()
    it = np.repeat(np.arange(4096), 3)[valid]
    later = it >= 256
    best256 = cnt[~later].max()
    run = np.maximum.accumulate(np.concatenate([[best256], cnt[later]]))[:-1]
    print(f"seed {seed}: static threshold {best256}: {(ub[later] > best256).sum()} contenders; running best of true counts: {(ub[later] > run).sum()}; "
          f"final best {cnt.max()}", flush=True)
