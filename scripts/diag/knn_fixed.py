"""Diagnostic (GPU box): kNN sweep time against the number of train rows at a fixed query count -- separates the
per-workgroup fixed cost (query load, first stage, drain, merge, stores) from the per-row cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops
dev = torch.device("cuda:0")
B, nq = 256, 8192
g = torch.Generator(device=dev); g.manual_seed(0)
q = torch.randint(-128, 128, (B, nq, 128), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
qn = (q.to(torch.int32) ** 2).sum(-1).to(torch.int32)
for nt in (192, 384, 768, 1536, 3072, 8192):
    t = q[:, :nt].contiguous(); tn = qn[:, :nt].contiguous()
    ws = torch.empty((ops._lib.load().vo_knn2_workspace_bytes(nq, nt, B) + 16,), dtype=torch.uint8, device=dev)
    for _ in range(3): ops.knn2(q, qn, t, tn, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.knn2(q, qn, t, tn, ws)
    e1.record(); torch.cuda.synchronize()
    print(f"nt={nt}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
