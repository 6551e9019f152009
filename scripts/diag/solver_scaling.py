"""Diagnostic (GPU box): solver kernel time vs number of samples (waves)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, synth
dev = torch.device("cuda:0")
sc = synth.two_view_scene(4000, seed=3)
pts4 = torch.from_numpy(np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)).to(dev)
rng = np.random.default_rng(0)
for H in (64, 4096, 16384, 65536, 131072, 262144):
    smp = torch.from_numpy(np.stack([rng.choice(4000, 7, replace=False) for _ in range(4096)]).astype(np.int32)).to(dev)
    smp = smp.repeat((H + 4095) // 4096, 1)[:H].contiguous()
    ops.solve_7pt(smp, pts4); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.solve_7pt(smp, pts4)
    e1.record(); torch.cuda.synchronize()
    print("7pt H", H, "waves", H // 64, "us", round(e0.elapsed_time(e1) / 3 * 1e3, 1), flush=True)
