"""Diagnostic (GPU box): 7-point solver time on the bench's own samples vs on a clean synthetic scene."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, synth
import bench
dev = torch.device("cuda:0")
ps = bench.PairSet(1, dev, 0, 1)
r = bench.step(ps); torch.cuda.synchronize()
n = int(r.n_match[0])
pts4 = r.pts4[0].contiguous()
smp = r.f_samples[0].contiguous()
def t(fn, rep=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3
big = smp.repeat(64, 1).contiguous()
print("bench data: 4096 samples", round(t(lambda: ops.solve_7pt(smp, pts4)), 1), "us; 262144 samples", round(t(lambda: ops.solve_7pt(big, pts4)), 1), "us")
models, nm = ops.solve_7pt(smp, pts4)
print("models per sample histogram", np.bincount(nm.cpu().numpy(), minlength=4))
p = pts4[:n].cpu().numpy()
print("points: n", n, "x range", p[:, 0].min(), p[:, 0].max(), "unique rows", len(np.unique(p, axis=0)))
