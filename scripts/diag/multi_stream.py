"""Diagnostic (GPU box): one 256-pair batch per step on one stream, against the same pairs as 2 or 4
concurrent sub-batches on streams of their own (the small latency-bound kernels of one sub-batch then run
under the GPU-filling kernels of another).  usage: multi_stream.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from visual_odometry_amd import pipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ps = bench.PairSet(256, dev, 0)


def run(parts):
    B = 256 // parts
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(parts - 1)]
    wss = [torch.empty((pipeline.workspace_bytes(bench.NQ, bench.NT, B, bench.H, bench.H),), dtype=torch.uint8, device=dev)
           for _ in range(parts)]

    def step():
        out = []
        for i, st in enumerate(streams):
            sl = slice(i * B, (i + 1) * B)
            with torch.cuda.stream(st):
                out.append(pipeline.match_and_ransac(ps.q[sl], ps.t[sl], ps.kp_q[sl], ps.kp_t[sl], bench.RATIO, bench.F_THR,
                                                     bench.F_CONF, bench.H, xyz_q=ps.xyz_q[sl], K=bench.K, p_threshold=bench.P_THR,
                                                     p_confidence=bench.P_CONF, p_iters=bench.H, first_block_pnp=bench.FIRST_BLOCK,
                                                     workspace=wss[i], wait_pnp=False))
        return out
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{parts} stream(s) x {B} pairs: {dt * 1e3:.3f} ms per 256 pairs = {256 / dt:.0f} pairs/s", flush=True)
    return r


for rep in range(2):
    for parts in (1, 2, 4):
        run(parts)
