"""Diagnostic (GPU box): time of the f32 -> int8 descriptor conversion alone (vo_desc_f32_to_i8 through the C ABI), 256 x 8192
rows as in one cfg4 step, whatever library VO_HIP_LIB names."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops
dev = torch.device("cuda:0")
n = 256 * 8192
x = [torch.randint(0, 64, (n, 128), device=dev, dtype=torch.int32).to(torch.float32) for _ in range(2)]
out = (torch.empty((n, 128), dtype=torch.int8, device=dev), torch.empty((n,), dtype=torch.int32, device=dev))
flags = torch.zeros((1,), dtype=torch.int32, device=dev)
for i in range(4): ops.desc_to_i8(x[i & 1], flags, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40): ops.desc_to_i8(x[i & 1], flags, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
print(f"{os.path.basename(os.environ.get('VO_HIP_LIB', 'in-tree'))}: {us:.1f} us per {n} rows, {n * 644 / us / 1e6:.2f} TB/s", flush=True)
