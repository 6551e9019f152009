#!/usr/bin/env python3
"""Standalone timing of the three scoring kernels at BASELINE cfg5's shape (65 536 models x 32 768
correspondences, ~50 % outliers), HIP events around vo_score_* on the launch stream.  Prints one JSON object
(-> profiles/r3_score_standalone.json).  The models are P3P / 7-point / 5-point solutions of random minimal
samples of the scene, replicated to H rows (what RANSAC feeds the kernels).  usage: bench_score.py [H] [N] [reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import _lib, ops, synth  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
_lib.require_device(0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(REPS):
        e0.record(); out = fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts)), out


out = {"H": H, "N": N, "reps": REPS, "device": torch.cuda.get_device_name(0),
       "how": "scripts/bench_score.py on one MI355X (gpurun): HIP events around vo_score_* at H models x N correspondences, ~50 % outliers; "
              "valu_issue_frac = evaluations x VALU instructions per evaluation of the filter loop every evaluation goes through "
              "(profiles/r3_score_*_isa.json, scripts/isa_count.py) / time / 78.64 T lane-instructions/s (256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz)"}
VALU_PEAK = 78.6432e12


def isa(kind):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", f"r3_score_{kind}_isa.json")))["valu_per_evaluation"]
    except (OSError, ValueError, KeyError):
        return None


def roof(kind, t):
    v = isa(kind)
    return {"valu_per_evaluation": v, "valu_issue_frac": (H * N * v / t / VALU_PEAK) if v else None}

K = synth.K_MALAGA
ps = synth.pnp_scene(N, K=K, size=synth.IMAGE_SIZE["malaga"], seed=9, outlier_frac=0.5)
obj, img = d(ps["obj"]), d(ps["img"])
subs = d(ops.gen_subsets(N, 4, 4096))
models, _, nm = ops.solve_p3p(subs, obj, img, K)
models = models[nm > 0]
models = models[torch.arange(H, device=dev) % models.shape[0]].contiguous()
t, cnt = timed(lambda: ops.score_pnp(models, obj, img, K, 1.0))
out["pnp"] = {"seconds": t, "evaluations": H * N, "evals_per_s": H * N / t, **roof("pnp", t), "max_inliers": int(cnt.max()),
              "mean_inliers": float(cnt.float().mean())}

sc = synth.two_view_scene(N, K=K, size=synth.IMAGE_SIZE["malaga"], seed=9, outlier_frac=0.5)
pts4 = d(np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32))
subs = d(ops.gen_subsets(N, 7, 4096, np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)))
m7, nm7 = ops.solve_7pt(subs, pts4)
m7 = m7.reshape(-1, 9)[(torch.arange(3, device=dev)[None, :] < nm7[:, None]).reshape(-1)]
m7 = m7[torch.arange(H, device=dev) % m7.shape[0]].contiguous()
t, cnt = timed(lambda: ops.score_fund(m7, pts4, 0.1))
out["fund"] = {"seconds": t, "evaluations": H * N, "evals_per_s": H * N / t, **roof("fund", t), "max_inliers": int(cnt.max())}

fxy = np.array([K[0, 0], K[1, 1], K[0, 0], K[1, 1]]); cxy = np.array([K[0, 2], K[1, 2], K[0, 2], K[1, 2]])
x4 = d((np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float64) - cxy) / fxy)
subs = d(ops.gen_subsets(N, 5, 1024))
m5, nm5 = ops.solve_5pt(subs, x4)
m5 = m5.reshape(-1, 9)[(torch.arange(10, device=dev)[None, :] < nm5[:, None]).reshape(-1)]
m5 = m5[torch.arange(H, device=dev) % m5.shape[0]].contiguous()
t, cnt = timed(lambda: ops.score_ess(m5, x4, 1.0 / ((K[0, 0] + K[1, 1]) / 2)))
out["ess"] = {"seconds": t, "evaluations": H * N, "evals_per_s": H * N / t, **roof("ess", t), "max_inliers": int(cnt.max())}
out["library"] = os.environ.get("VO_HIP_LIB", "visual_odometry_amd/libvo_hip.so")
print(json.dumps(out, indent=1))
