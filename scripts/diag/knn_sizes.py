"""Diagnostic (GPU box): kNN sweep rate at the sizes of the configs (per launch, incl. merge)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
for B, n in ((1, 4096), (1, 8192), (16, 8192), (64, 8192), (1, 32768), (4, 32768)):
    q = torch.randint(-128, 128, (B, n, 128), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
    t = torch.randint(-128, 128, (B, n, 128), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
    qn = (q.to(torch.int32) ** 2).sum(-1).to(torch.int32); tn = (t.to(torch.int32) ** 2).sum(-1).to(torch.int32)
    ws = torch.empty((ops._lib.load().vo_knn2_workspace_bytes(n, n, B) + 16,), dtype=torch.uint8, device=dev)
    ops.knn2(q, qn, t, tn, ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.knn2(q, qn, t, tn, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tops = 2.0 * B * n * n * 128 / (ms * 1e-3) / 1e12
    print(f"B={B} n={n}: {ms*1e3:.1f} us per launch, {tops:.0f} TOP/s = {tops/5033.16*100:.1f}% of int8 peak", flush=True)
