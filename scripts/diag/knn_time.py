"""Diagnostic (GPU box): time of the kNN sweep alone (vo_knn2_l2_i8 through the C ABI) on the bench's descriptor recipe,
whatever library VO_HIP_LIB names -- also for timing-only builds whose results are wrong.  argv: cfg4 | cfg5 ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, sweep, synth
dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["cfg4"]):
    n, B = {"cfg4": (8192, 256), "cfg5": (32768, 32)}[cfg]
    qs, ts = [], []
    for lo in range(0, B, 32):
        g = sweep.frame_pairs_device(list(range(lo, min(B, lo + 32))), n, n, synth.K_MALAGA, synth.IMAGE_SIZE["malaga"], dev)
        qs.append(g["q"]); ts.append(g["t"])
    q, t = torch.cat(qs), torch.cat(ts)
    flags = torch.zeros((2,), dtype=torch.int32, device=dev)
    qi8, qn, _ = ops.desc_to_i8(q.reshape(B * n, 128), flags[0:1]); ti8, tn, _ = ops.desc_to_i8(t.reshape(B * n, 128), flags[1:2])
    del q, t, qs, ts
    a = (qi8.reshape(B, n, 128), qn.reshape(B, n), ti8.reshape(B, n, 128), tn.reshape(B, n))
    ws = torch.empty((ops._lib.load().vo_knn2_workspace_bytes(n, n, B) + 16,), dtype=torch.uint8, device=dev)
    for _ in range(3): ops.knn2(*a, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.knn2(*a, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tops = 2.0 * B * n * n * 128 / (ms * 1e-3) / 1e12
    print(f"{os.path.basename(os.environ.get('VO_HIP_LIB', 'in-tree'))} {cfg}: {ms * 1e3:.1f} us, {tops / 5033.16:.3f} of int8 peak", flush=True)
