"""Diagnostic (GPU box): does the kNN sweep's time depend on the VALUES of the int8 rows (power: the kernel is clock-limited)?
The same descriptor recipe encoded with different offsets u - o (distances are offset-invariant; values above 127 + o are
clipped here, so this is a timing experiment, not a product path), plus all-zero and uniform random rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, sweep, synth
dev = torch.device("cuda:0")
n, B = 8192, 256
qs, ts = [], []
for lo in range(0, B, 32):
    g = sweep.frame_pairs_device(list(range(lo, min(B, lo + 32))), n, n, synth.K_MALAGA, synth.IMAGE_SIZE["malaga"], dev)
    qs.append(g["q"].to(torch.int16)); ts.append(g["t"].to(torch.int16))
q, t = torch.cat(qs), torch.cat(ts)
del qs, ts
print("value histogram: mean %.1f, frac zero %.3f, frac > 127 %.5f, max %d" % (q.float().mean().item(), (q == 0).float().mean().item(),
      (q > 127).float().mean().item(), int(q.max())), flush=True)
ws = torch.empty((ops._lib.load().vo_knn2_workspace_bytes(n, n, B) + 16,), dtype=torch.uint8, device=dev)

def enc(x, o):
    c = (x - o).clamp(-128, 127)
    return c.to(torch.int8).contiguous(), (c.to(torch.int32) ** 2).sum(-1).to(torch.int32).contiguous()

def run(name, qi8, qn, ti8, tn):
    for _ in range(3): ops.knn2(qi8, qn, ti8, tn, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.knn2(qi8, qn, ti8, tn, ws)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 100:.1f} us", flush=True)

for o in (128, 64, 32, 16, 0):
    run(f"offset {o}", *enc(q, o), *enc(t, o))
z = torch.zeros_like(q)
run("all zero", *enc(z, 0), *enc(z, 0))
r = torch.randint(0, 256, q.shape, device=dev, dtype=torch.int16)
r2 = torch.randint(0, 256, q.shape, device=dev, dtype=torch.int16)
run("uniform random", *enc(r, 128), *enc(r2, 128))
run("offset 128 again", *enc(q, 128), *enc(t, 128))
